"""Host-side mirror of the reference's receive interface, over the C ABI.

The reference starts one ``rcv_thread(thread_param_t*)`` per channel (main.c:228-231)
and each hands ``msgblk_t`` bursts to ``decodeVdlm2`` (d8psk.c:201).  ``Receiver`` keeps
those names and meanings: it is configured with the same ``(chn, Fr, Fo)`` triples and
``SDRINRATE``; ``push()`` plays the producer's Bar1/Bar2 hand-off of one sample block
(rtl.c:283-294) for *all* channels; ``poll()`` yields ``Burst`` objects carrying exactly the
``msgblk_t`` fields.  All DSP happens in libvdl2gpu.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import lib as _lib

STEPRATE = 25000           # vdlm2.h:33
RTLINBUFSZ = 16 * 4096     # vdlm2.h:35 (bytes per RTL hand-off = 32768 complex samples)


@dataclass(frozen=True)
class ThreadParam:
    """thread_param_t, vdlm2.h:49-52."""
    chn: int
    Fr: int
    Fo: int


@dataclass
class Burst:
    """The msgblk_t fields the DSP fills (vdlm2.h:39-47) plus stream-time stamps."""
    stream: int
    chn: int
    Fr: int
    nbrow: int
    nlbyte: int
    df: float
    ppm: float
    trig_dec: int
    end_dec: int
    trig_sample: int
    end_sample: int
    data: bytes            # 8 rows x 255 bytes, row-major

    def key(self):
        return (self.stream, self.chn, self.nbrow, self.nlbyte, self.data)


def plan_channels(fc: int, offsets: Sequence[int]) -> List[ThreadParam]:
    """thread_param_t list the way rtl.c:245-247 fills it: Fo = Fr - Fc."""
    return [ThreadParam(chn=i, Fr=fc + fo, Fo=fo) for i, fo in enumerate(offsets)]


class Receiver:
    def __init__(self, sdrinrate: int, channels: Sequence[ThreadParam] | Sequence[Sequence[ThreadParam]],
                 fmt: str = "cu8", max_push: int = 1 << 22, device: int = 0, sdrclk: int = 0,
                 max_bursts: int = 0, keep_dec: bool = False, serial: bool = False, full_scan: bool = False,
                 frames: bool = False, rtl_quirk: bool = False, flags: int = 0, testhooks: bool = False):
        # testhooks: load libvdl2gpu_test.so, the build that honours F_TEST_NOREGION / VDL2GPU_PRIM_DROP / VDL2GPU_SPLIT_SAMPLES
        self.L = _lib.load(testhooks=testhooks or bool(flags & _lib.F_TEST_NOREGION))
        if channels and isinstance(channels[0], ThreadParam):
            channels = [list(channels)]
        self.nstreams = len(channels)
        self.nbch = len(channels[0])
        if any(len(c) != self.nbch for c in channels):
            raise ValueError("every stream needs the same number of channels")
        self.fmt = fmt
        self.rate = sdrinrate
        self.sample_bytes = _lib.SAMPLE_BYTES[fmt]
        self._chan = (_lib.ChanT * (self.nstreams * self.nbch))()
        for s, plan in enumerate(channels):
            for c, tp in enumerate(plan):
                self._chan[s * self.nbch + c] = _lib.ChanT(tp.chn, tp.Fr, tp.Fo)
        cfg = _lib.ConfigT()
        cfg.struct_size = C.sizeof(_lib.ConfigT)
        cfg.sdrinrate = sdrinrate
        cfg.sdrclk = sdrclk
        cfg.fmt = _lib.FMT[fmt]
        cfg.nbch = self.nbch
        cfg.nstreams = self.nstreams
        cfg.chan = self._chan
        cfg.max_push = max_push
        cfg.device = device
        cfg.max_bursts = max_bursts
        cfg.flags = (_lib.F_KEEP_DEC if keep_dec else 0) | (_lib.F_SERIAL if serial else 0) | (_lib.F_FULLSCAN if full_scan else 0) | (_lib.F_FRAMES if frames else 0) | (_lib.F_RTL_QUIRK if rtl_quirk else 0) | flags
        self.max_push = max_push
        self.h = C.c_void_p()
        rc = self.L.vdl2gpu_create(C.byref(cfg), C.byref(self.h))
        if rc != 0:
            self.h = None
            raise _lib.Vdl2GpuError(f"vdl2gpu_create failed: {self.L.vdl2gpu_strerror(rc).decode()}")

    # -------------------------------------------------------------------- data path
    def _check(self, rc: int):
        if rc < 0:
            msg = self.L.vdl2gpu_last_error(self.h).decode()
            raise _lib.Vdl2GpuError(f"{self.L.vdl2gpu_strerror(rc).decode()}: {msg}")
        return rc

    def push(self, raw: np.ndarray):
        """One hand-off of host samples. ``raw``: 1-D (single stream) or [nstreams, n] raw array."""
        a = np.ascontiguousarray(raw)
        if a.ndim == 1:
            a = a[None, :]
        if a.shape[0] != self.nstreams:
            raise ValueError("first dimension must be nstreams")
        nbytes = a.shape[1] * a.itemsize
        n = nbytes // self.sample_bytes
        self._check(self.L.vdl2gpu_push(self.h, a.ctypes.data_as(C.c_void_p), n, nbytes, _lib.MEM_HOST))
        return n

    # ------------------------------------------------------------------ ingest ring (include/vdl2gpu.h)
    def ring_init(self, slot_samples: int, nslots: int = 4):
        self._check(self.L.vdl2gpu_ring_init(self.h, slot_samples, nslots))
        self._ring_slot_samples = slot_samples

    def ring_acquire(self) -> np.ndarray:
        """The next slot of page-locked host memory as a [nstreams, slot_bytes] uint8 view: fill it in
        place (first ``nsamples * sample_bytes`` bytes of every row), then ``ring_commit(nsamples)``."""
        stride = C.c_size_t(0)
        p = self.L.vdl2gpu_ring_acquire(self.h, C.byref(stride))
        if not p:
            raise RuntimeError("vdl2gpu_ring_acquire: " + self.L.vdl2gpu_last_error(self.h).decode())
        buf = (C.c_uint8 * (stride.value * self.nstreams)).from_address(p)
        return np.frombuffer(buf, dtype=np.uint8).reshape(self.nstreams, stride.value)

    def ring_commit(self, nsamples: int):
        self._check(self.L.vdl2gpu_ring_commit(self.h, nsamples))

    def push_device(self, ptr: int, nsamples: int, stream_stride_bytes: int = 0):
        """Samples already resident in HBM (e.g. ``tensor.data_ptr()``)."""
        self._check(self.L.vdl2gpu_push(self.h, C.c_void_p(ptr), nsamples, stream_stride_bytes, _lib.MEM_DEVICE))

    def sync(self):
        self._check(self.L.vdl2gpu_sync(self.h))

    def poll(self, max_bursts: int = 4096) -> List[Burst]:
        out: List[Burst] = []
        buf = (_lib.BurstT * max_bursts)()
        while True:
            n = self._check(self.L.vdl2gpu_poll(self.h, buf, max_bursts))
            for i in range(n):
                b = buf[i]
                out.append(Burst(b.stream, b.chn, b.Fr, b.nbrow, b.nlbyte, b.df, b.ppm, b.trig_dec, b.end_dec,
                                 b.trig_sample, b.end_sample, bytes(b.data)))
            if n < max_bursts:
                return out

    def poll_ready(self, max_bursts: int = 4096) -> List[Burst]:
        """Bursts of pushes the GPU has already finished; never waits."""
        out: List[Burst] = []
        buf = (_lib.BurstT * max_bursts)()
        while True:
            n = self.poll_ready_raw(buf, max_bursts)
            for i in range(n):
                b = buf[i]
                out.append(Burst(b.stream, b.chn, b.Fr, b.nbrow, b.nlbyte, b.df, b.ppm, b.trig_dec, b.end_dec,
                                 b.trig_sample, b.end_sample, bytes(b.data)))
            if n < max_bursts:
                return out

    def poll_ready_raw(self, buf, max_bursts: int) -> int:
        """vdl2gpu_poll_ready: bursts of pushes that have already finished; never waits."""
        return self._check(self.L.vdl2gpu_poll_ready(self.h, buf, max_bursts))

    def poll_raw(self, buf, max_bursts: int) -> int:
        """vdl2gpu_poll into a caller-owned (lib.BurstT * max_bursts) array; no Python objects."""
        return self._check(self.L.vdl2gpu_poll(self.h, buf, max_bursts))

    def run(self, raw: np.ndarray, block: Optional[int] = None) -> List[Burst]:
        """Feed a whole recording in ``block``-sample hand-offs and return every burst."""
        a = np.ascontiguousarray(raw)
        if a.ndim == 1:
            a = a[None, :]
        per = self.sample_bytes // a.itemsize
        n = a.shape[1] // per
        block = block or self.max_push
        out: List[Burst] = []
        for s in range(0, n, block):
            e = min(n, s + block)
            self.push(a[:, s * per:e * per])
            out += self.poll()
        return out

    # ------------------------------------------------------------------ block path
    def decode_blocks(self, blocks: Sequence, max_frames: int = 0) -> List[Tuple[int, bytes]]:
        """blk_thread for a batch (vdlm2.c:84-161): (index of the block, hdata) of every frame the
        reference would pass to out().  ``blocks``: Burst objects or (nbrow, nlbyte, data) tuples."""
        n = len(blocks)
        if n == 0:
            return []
        arr = (_lib.BurstT * n)()
        for i, b in enumerate(blocks):
            nbrow, nlbyte, data = (b.nbrow, b.nlbyte, b.data) if hasattr(b, "nbrow") else b
            arr[i].nbrow, arr[i].nlbyte = nbrow, nlbyte
            C.memmove(C.addressof(arr[i].data), bytes(data), min(len(data), 8 * 255))
            if hasattr(b, "chn"):
                arr[i].stream, arr[i].chn, arr[i].Fr = b.stream, b.chn, b.Fr
        cap = max_frames or 4 * n
        out = (_lib.FrameT * cap)()
        dropped = C.c_int(0)
        nf = self._check(self.L.vdl2gpu_decode_blocks(self.h, arr, n, out, cap, C.byref(dropped)))
        if dropped.value:
            raise _lib.Vdl2GpuError(f"{dropped.value} frames dropped: raise max_frames")
        return [(out[i].block, bytes(out[i].data[:out[i].len])) for i in range(nf)]

    def poll_frames(self, max_frames: int = 4096) -> List[Tuple[int, int, bytes]]:
        """(stream, chn, hdata) of every frame of everything pushed so far (needs frames=True)."""
        out = []
        buf = (_lib.FrameT * max_frames)()
        while True:
            n = self._check(self.L.vdl2gpu_poll_frames(self.h, buf, max_frames))
            out += [(buf[i].stream, buf[i].chn, bytes(buf[i].data[:buf[i].len])) for i in range(n)]
            if n < max_frames:
                return out

    # ------------------------------------------------------------------ bookkeeping
    def stats(self) -> dict:
        st = _lib.StatsT()
        self._check(self.L.vdl2gpu_get_stats(self.h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in _lib.StatsT._fields_}

    def timing(self, reset: bool = False) -> dict:
        t = _lib.TimingT()
        self._check(self.L.vdl2gpu_get_timing(self.h, C.byref(t), int(reset)))
        return {n: getattr(t, n) for n, _ in _lib.TimingT._fields_}

    # ------------------------------------------------------------------ diagnostics
    def debug_dec(self, stream: int, ch: int, max_complex: int = 1 << 24) -> np.ndarray:
        buf = np.empty(2 * max_complex, np.float32)
        n = self.L.vdl2gpu_debug_dec(self.h, stream, ch, buf.ctypes.data_as(C.c_void_p), max_complex)
        self._check(int(n))
        return buf[:2 * n].view(np.complex64).copy()

    def debug_atan2f(self, y: np.ndarray, x: np.ndarray) -> np.ndarray:
        y = np.ascontiguousarray(y, np.float32)
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(y)
        self._check(self.L.vdl2gpu_debug_atan2f(self.h, y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p),
                                                out.ctypes.data_as(C.c_void_p), y.size))
        return out

    def debug_cands(self, stream: int, ch: int, max_cands: int = 6144) -> np.ndarray:
        buf = np.zeros((max_cands, 6), np.int32)
        n = self._check(self.L.vdl2gpu_debug_cands(self.h, stream, ch, buf.ctypes.data_as(C.c_void_p), max_cands))
        return buf[:n].copy()

    def host_profile(self, reset: bool = False) -> dict:
        """seconds of the calling thread inside vdl2gpu_push since the last reset: `enqueue` (launches and event operations) and
        `wait_for_ring` (blocked until the GPU has finished the push whose output ring, tables and planes this one reuses)"""
        buf = (C.c_double * 8)()
        self._check(self.L.vdl2gpu_get_host_profile(self.h, buf, int(reset)))
        v = list(buf)
        return {"wait_input": v[0], "enqueue": v[1] + v[3] + v[5], "wait_for_ring": v[2], "spill": v[4], "event_wait": v[6], "pushes": int(v[7])}

    def debug_clheads(self, stream: int, ch: int, max_cands: int = 6144) -> np.ndarray:
        """(n_s_rel, packed) per candidate of the last push, in debug_cands()'s order: where and how the idle search resumes."""
        buf = np.zeros((max_cands, 2), np.int32)
        n = self._check(self.L.vdl2gpu_debug_clheads(self.h, stream, ch, buf.ctypes.data_as(C.c_void_p), max_cands))
        return buf[:n].copy()

    def debug_heads(self, max_entries: int = 1 << 16) -> np.ndarray:
        """Header soft bits of every trigger of the last push (needs flags=lib.F_DEBUG_HEADS): structured array
        (nstar, sc, clk0, p2err, perr, err, pfr, soft[25])."""
        dt = np.dtype([("nstar", "<i8"), ("sc", "<i4"), ("clk0", "<i4"), ("p2err", "<f4"), ("perr", "<f4"),
                       ("err", "<f4"), ("pfr", "<f4"), ("soft", "<f4", (25,)), ("pad", "<u4")])
        buf = np.zeros(max_entries, dt)
        n = self._check(self.L.vdl2gpu_debug_heads(self.h, buf.ctypes.data_as(C.c_void_p), max_entries))
        return buf[:n].copy()

    def debug_counters(self, n: int = 16, reset: bool = True):
        buf = (C.c_ulonglong * 64)()
        self._check(self.L.vdl2gpu_debug_counters(self.h, buf, n, int(reset)))
        return list(buf[:n])

    def close(self):
        if getattr(self, "h", None):
            self.L.vdl2gpu_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def choose_fc(freqs: Sequence[int], sdrinrate: int = 2_000_000, tuner: str = "rtl"):
    """The reference's frequency plan for a list of channel frequencies in Hz.

    tuner "rtl": chooseFc() of rtl.c:123-160, Fo = Fr - Fc (rtl.c:245-247); returns (Fc, [ThreadParam...]).
    tuner "air": chooseFc() of air.c:47-70, Fo = Fr - (Fc + SDRINRATE/4) (air.c:182-184); returns
    (Fc, [ThreadParam...], (r10, r11)) with the two R820T2 filter registers the reference writes at 5 MS/s.
    Fc == 0 means the reference refuses the list ("Frequencies too far apart")."""
    L = _lib.load()
    n = len(freqs)
    fr = (C.c_uint * n)(*[int(f) for f in freqs])
    fo = (C.c_int * n)()
    fc = C.c_uint(0)
    if tuner == "rtl":
        rc = L.vdl2gpu_choose_fc_rtl(fr, n, sdrinrate, C.byref(fc), fo)
        extra = None
    elif tuner == "air":
        r10, r11 = C.c_int(0), C.c_int(0)
        rc = L.vdl2gpu_choose_fc_air(fr, n, sdrinrate, C.byref(fc), fo, C.byref(r10), C.byref(r11))
        extra = (r10.value, r11.value)
    else:
        raise ValueError("tuner must be 'rtl' or 'air'")
    if rc < 0:
        raise _lib.Vdl2GpuError("vdl2gpu_choose_fc failed: " + L.vdl2gpu_strerror(rc).decode())
    plan = [ThreadParam(chn=i, Fr=int(freqs[i]), Fo=int(fo[i])) for i in range(n)]
    return (fc.value, plan) if extra is None else (fc.value, plan, extra)


def lo_table(sdrinrate: int, fo: int) -> np.ndarray:
    """Host LO table exactly as the library uploads it (d8psk.c:353-357)."""
    L = _lib.load()
    n = sdrinrate // STEPRATE
    buf = np.empty(2 * n, np.float32)
    rc = L.vdl2gpu_lo_table(sdrinrate, fo, buf.ctypes.data_as(C.c_void_p), n)
    if rc < 0:
        raise _lib.Vdl2GpuError("vdl2gpu_lo_table failed")
    return buf.view(np.complex64).copy()


def plan(total_in: int, n: int, sdrclk: int, lo_len: int) -> Tuple[int, int, int, int]:
    """(c0, no0, nf0, nout): decimation schedule of one push (d8psk.c:374-381 closed form)."""
    L = _lib.load()
    c0, no0, nf0, nout = C.c_int(), C.c_int(), C.c_int(), C.c_int64()
    rc = L.vdl2gpu_plan(total_in, n, sdrclk, lo_len, C.byref(c0), C.byref(no0), C.byref(nf0), C.byref(nout))
    if rc < 0:
        raise _lib.Vdl2GpuError("vdl2gpu_plan failed")
    return c0.value, no0.value, nf0.value, nout.value
