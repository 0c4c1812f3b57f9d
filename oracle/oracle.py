"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, bench.py (cpu_baseline leg), __graft_entry__.smoke().
The product package (vdlm2dec_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_DIR = os.path.join(_HERE, "_ref")

FMT = {"cu8": 0, "cs16": 1, "cf32": 2, "f32": 3, "cu8_quirk": 4}
NP_DTYPE = {"cu8": np.uint8, "cs16": np.int16, "cf32": np.float32, "f32": np.float32, "cu8_quirk": np.uint8}
PER_SAMPLE = {"cu8": 2, "cs16": 2, "cf32": 2, "f32": 1, "cu8_quirk": 2}


class VoBlock(C.Structure):
    _fields_ = [("nbrow", C.c_int32), ("nlbyte", C.c_int32), ("df", C.c_float), ("ppm", C.c_float),
                ("trig_dec", C.c_int64), ("end_dec", C.c_int64), ("data", (C.c_uint8 * 255) * 8)]


class VoTrigger(C.Structure):
    _fields_ = [("dec_index", C.c_int64), ("p2err", C.c_float), ("perr", C.c_float), ("err", C.c_float),
                ("pfr", C.c_float), ("of", C.c_float), ("clk", C.c_int32), ("accepted", C.c_int32),
                ("len_bits", C.c_int32), ("nhead", C.c_int32), ("head", C.c_float * 25)]


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "vdl2_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.vo_create.restype = C.c_void_p
        L.vo_create.argtypes = [C.c_uint, C.c_int, C.c_int]
        L.vo_destroy.argtypes = [C.c_void_p]
        L.vo_enable_taps.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.vo_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.vo_num_blocks.restype = C.c_size_t
        L.vo_num_blocks.argtypes = [C.c_void_p]
        L.vo_blocks.restype = C.POINTER(VoBlock)
        L.vo_blocks.argtypes = [C.c_void_p]
        L.vo_num_triggers.restype = C.c_size_t
        L.vo_num_triggers.argtypes = [C.c_void_p]
        L.vo_triggers.restype = C.POINTER(VoTrigger)
        L.vo_triggers.argtypes = [C.c_void_p]
        L.vo_num_dec.restype = C.c_size_t
        L.vo_num_dec.argtypes = [C.c_void_p]
        L.vo_dec_tap.restype = C.POINTER(C.c_float)
        L.vo_dec_tap.argtypes = [C.c_void_p]
        L.vo_num_phase_tap.restype = C.c_size_t
        L.vo_num_phase_tap.argtypes = [C.c_void_p]
        L.vo_phase_tap.restype = C.POINTER(C.c_float)
        L.vo_phase_tap.argtypes = [C.c_void_p]
        L.vo_lo_table.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.vo_block_frames.restype = C.c_int
        L.vo_block_frames.argtypes = [C.POINTER(VoBlock), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.vo_rs_decode.restype = C.c_int
        L.vo_rs_decode.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
        L.vo_reversebits.restype = C.c_uint
        L.vo_reversebits.argtypes = [C.c_uint, C.c_int]
        L.vo_pn_bits.argtypes = [C.c_void_p, C.c_size_t]
        L.vo_header_decode.restype = C.c_uint
        L.vo_header_decode.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        L.vo_atan2f.restype = C.c_float
        L.vo_atan2f.argtypes = [C.c_float, C.c_float]
        _lib = L
    return _lib


class Block:
    """Plain-Python view of one burst record (msgblk_t fields, vdlm2.h:39-47)."""
    __slots__ = ("chn", "nbrow", "nlbyte", "df", "ppm", "trig_dec", "end_dec", "data")

    def __init__(self, chn, nbrow, nlbyte, df, ppm, trig_dec, end_dec, data):
        self.chn, self.nbrow, self.nlbyte, self.df, self.ppm = chn, nbrow, nlbyte, df, ppm
        self.trig_dec, self.end_dec, self.data = trig_dec, end_dec, data  # data: bytes (8*255)

    def key(self):
        return (self.chn, self.nbrow, self.nlbyte, self.data)

    def __repr__(self):
        return f"Block(chn={self.chn}, nbrow={self.nbrow}, nlbyte={self.nlbyte}, trig={self.trig_dec}, end={self.end_dec})"


def frames_of_block(nbrow: int, nlbyte: int, data: bytes) -> List[bytes]:
    """Host block path (RS + HDLC unstuff + FCS) applied to one burst record."""
    L = lib()
    vb = VoBlock()
    vb.nbrow, vb.nlbyte = nbrow, nlbyte
    C.memmove(vb.data, data, 8 * 255)
    buf = (C.c_uint8 * 8192)()
    used = C.c_size_t(0)
    n = L.vo_block_frames(C.byref(vb), buf, len(buf), C.byref(used))
    raw = bytes(buf[:used.value])
    out, p = [], 0
    for _ in range(n):
        ln = raw[p] | (raw[p + 1] << 8)
        out.append(raw[p + 2:p + 2 + ln])
        p += 2 + ln
    return out


class OracleChannel:
    def __init__(self, rate: int, fo: int, fr: int, chn: int = 0, tap_dec: bool = False, tap_phase: bool = False):
        self.L = lib()
        self.h = self.L.vo_create(rate, fo, fr)
        self.chn = chn
        if tap_dec or tap_phase:
            self.L.vo_enable_taps(self.h, int(tap_dec), int(tap_phase))

    def feed(self, raw: np.ndarray, fmt: str):
        raw = np.ascontiguousarray(raw, dtype=NP_DTYPE[fmt])
        n = raw.size // PER_SAMPLE[fmt]
        self.L.vo_feed(self.h, raw.ctypes.data_as(C.c_void_p), n, FMT[fmt])

    def blocks(self) -> List[Block]:
        n = self.L.vo_num_blocks(self.h)
        p = self.L.vo_blocks(self.h)
        return [Block(self.chn, p[i].nbrow, p[i].nlbyte, p[i].df, p[i].ppm, p[i].trig_dec, p[i].end_dec,
                      bytes(p[i].data)) for i in range(n)]

    def triggers(self):
        n = self.L.vo_num_triggers(self.h)
        p = self.L.vo_triggers(self.h)
        return [dict(dec_index=p[i].dec_index, p2err=p[i].p2err, perr=p[i].perr, err=p[i].err, pfr=p[i].pfr,
                     of=p[i].of, clk=p[i].clk, accepted=p[i].accepted, len_bits=p[i].len_bits,
                     head=np.array(p[i].head[:p[i].nhead], np.float32)) for i in range(n)]

    def heads(self) -> np.ndarray:
        """every soft bit given to the header Viterbi (viterbi_add, d8psk.c:83), in call order"""
        t = self.triggers()
        return np.concatenate([x["head"] for x in t]) if t else np.zeros(0, np.float32)

    def dec(self) -> np.ndarray:
        n = self.L.vo_num_dec(self.h)
        p = self.L.vo_dec_tap(self.h)
        return np.ctypeslib.as_array(p, shape=(2 * n,)).copy().view(np.complex64) if n else np.zeros(0, np.complex64)

    def phases(self) -> np.ndarray:
        n = self.L.vo_num_phase_tap(self.h)
        p = self.L.vo_phase_tap(self.h)
        return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.float32)

    def lo_table(self) -> np.ndarray:
        buf = np.zeros(2 * 4096, np.float32)
        ln = C.c_int(0)
        self.L.vo_lo_table(self.h, buf.ctypes.data_as(C.c_void_p), C.byref(ln))
        return buf[:2 * ln.value].view(np.complex64).copy()

    def close(self):
        if self.h:
            self.L.vo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_oracle(raw: np.ndarray, fmt: str, rate: int, fos: Sequence[int], fc: int = 136975000,
               chunk: Optional[int] = None) -> List[Block]:
    """All channels of one wideband stream through the CPU restatement."""
    out: List[Block] = []
    for c, fo in enumerate(fos):
        ch = OracleChannel(rate, fo, fc + fo, chn=c)
        if chunk:
            per = PER_SAMPLE[fmt]
            for s in range(0, raw.size // per, chunk):
                ch.feed(raw[s * per:(s + chunk) * per], fmt)
        else:
            ch.feed(raw, fmt)
        out += ch.blocks()
        ch.close()
    return out


# --------------------------------------------------------------------------- the real reference
def have_ref() -> bool:
    return os.path.exists(os.path.join(REF_DIR, "ref_rtl"))


def build_ref() -> bool:
    """Build oracle/_ref from /root/reference when it exists (this container only)."""
    if not os.path.isdir("/root/reference"):
        return False
    subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return True


def run_ref(path: str, fmt: str, rate: int, fo: int, fr: int, out_path: str, quirk: int = 0,
            tap_path: str = "", ofast: bool = False):
    """One channel through the REAL reference (oracle/_ref/ref_rtl or ref_air; ofast: the build with the
    reference's own -Ofast -march=native, CMakeLists.txt:4)."""
    exe = os.path.join(REF_DIR, ("ref_air" if fmt == "f32" else "ref_rtl") + ("_ofast" if ofast else ""))
    subprocess.check_call([exe, path, fmt, str(rate), str(fo), str(fr), out_path, str(quirk), tap_path])
    blocks, frames = [], []
    with open(out_path) as f:
        for line in f:
            p = line.split()
            if p[0] == "B":
                tv = p[5][1:].split(".")     # "t<sec>.<usec>": the harness's gettimeofday is the sample clock
                blocks.append(dict(nbrow=int(p[1]), nlbyte=int(p[2]), ppm=float(p[3]), df_bits=int(p[4], 16),
                                   tv=int(tv[0]) * 1_000_000 + int(tv[1]), data=bytes.fromhex(p[-1])))
            elif p[0] == "F":
                frames.append(dict(nbrow=int(p[1]), nlbyte=int(p[2]), frame=bytes.fromhex(p[4])))
    taps = None
    if tap_path:
        taps = np.fromfile(tap_path, dtype=np.dtype([("t", "<u4"), ("a", "<f4"), ("b", "<f4"), ("c", "<f4")]))
    return blocks, frames, taps
