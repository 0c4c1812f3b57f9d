"""Synthetic VDL Mode 2 transmitter (test / benchmark signal source).

The reference ships no recorded IQ and no transmitter, so every
fixture and benchmark stream is produced here: AVLC frame -> FCS-16 -> HDLC bit
stuffing -> burst header (25 bits, (25,20) code whose parity-check columns are the
data table at viterbi.c:29-35) -> RS(255,249) parity per 249-byte row (field and
generator roots as consumed by rs.c:71-109) -> column-major interleave exactly as
the receiver de-interleaves it (d8psk.c:117-206) -> scrambler seeded 0x4D4B
(d8psk.c:54-65, 299) -> Gray-coded tribits -> differential 8-PSK symbols behind the
17-symbol sync word that d8psk.h:20-26 (SW) encodes -> raised-cosine pulse
(alpha 0.6) -> frequency-translate to the channel offset -> sum, AWGN, quantise
to cu8 (rtl.c:287-289 inverse) / cs16 / cf32 / real f32.

This is a tool on the *input* side of the hot path; it
is plain numpy on purpose and is never part of the timed region.
"""
from __future__ import annotations

import dataclasses
import math
from typing import List, Optional, Sequence

import numpy as np

SYMRATE = 10500.0
ROW_DATA = 249
ROW_BITS = ROW_DATA * 8  # 1992, d8psk.c:94
SCRAMBLER_SEED = 0x4D4B  # d8psk.c:299

# parity-check columns of the (25,20) header code (data, viterbi.c:29-35)
HEADER_H = (
    0b00110, 0b00111, 0b01001, 0b01010, 0b01011,
    0b01100, 0b01110, 0b01111, 0b10001, 0b10011,
    0b10101, 0b10110, 0b11000, 0b11001, 0b11010,
    0b11011, 0b11100, 0b11101, 0b11110, 0b11111,
    0b10000, 0b01000, 0b00100, 0b00010, 0b00001,
)

# unique-word phase increments (x pi/4) implied by SW[]
UW_INCR = (0, 3, 2, 4, 0, 1, 6, 4, 1, 7, 2, 5, 6, 5, 7, 3)

# tribit (b0 b1 b2, b0 first on air) -> phase increment (x pi/4), Gray
_GRAY = {0b000: 0, 0b001: 1, 0b011: 2, 0b010: 3, 0b110: 4, 0b111: 5, 0b101: 6, 0b100: 7}
GRAY_LUT = np.array([_GRAY[i] for i in range(8)], dtype=np.int64)


# --------------------------------------------------------------------------- GF(256)
def _gf_tables():
    exp = np.zeros(512, dtype=np.int64)
    log = np.zeros(256, dtype=np.int64)
    x = 1
    for i in range(255):
        exp[i] = x
        log[x] = i
        x <<= 1
        if x & 0x100:
            x ^= 0x187
    exp[255:510] = exp[0:255]
    return exp, log


GF_EXP, GF_LOG = _gf_tables()


def _gf_mul(a: int, b: int) -> int:
    if a == 0 or b == 0:
        return 0
    return int(GF_EXP[GF_LOG[a] + GF_LOG[b]])


def _rs_genpoly() -> List[int]:
    """g(x) = prod_{i=0..5} (x - alpha^(120+i)), coefficients highest power first."""
    g = [1]
    for i in range(6):
        root = int(GF_EXP[(120 + i) % 255])
        ng = g + [0]
        for j in range(len(g)):
            ng[j + 1] ^= _gf_mul(g[j], root)
        g = ng
    return g  # length 7, g[0] == 1


RS_GEN = _rs_genpoly()


def rs_parity(row: Sequence[int]) -> List[int]:
    """6 parity bytes for a 249-byte row: remainder of row(x)*x^6 mod g(x).

    Codeword polynomial is sum data[j] x^(254-j) (rs.c:94-109 evaluates it that way),
    parity goes to columns 249..254 in this order.
    """
    assert len(row) == ROW_DATA
    rem = [0] * 6
    for d in row:
        fb = d ^ rem[0]
        rem = rem[1:] + [0]
        if fb:
            for j in range(6):
                rem[j] ^= _gf_mul(fb, RS_GEN[j + 1])
    return rem


# --------------------------------------------------------------------------- HDLC / FCS
def fcs16(data: bytes) -> int:
    """CRC-16/X.25 (reflected 0x8408, init/xorout 0xffff); receiver residue 0xf0b8."""
    crc = 0xFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ 0x8408 if crc & 1 else crc >> 1
    return crc ^ 0xFFFF


def icao_addr_bytes(addr27: int, low_bits: int = 0) -> bytes:
    """Inverse of the address unpacking at out.c:426-435 (6/7/7/7 split, bit reversed)."""
    def rev(v, n):
        o = 0
        for _ in range(n):
            o = (o << 1) | (v & 1)
            v >>= 1
        return o
    b0 = (rev((addr27 >> 21) & 0x3F, 6) << 2) | (low_bits & 3)
    b1 = rev((addr27 >> 14) & 0x7F, 7) << 1
    b2 = rev((addr27 >> 7) & 0x7F, 7) << 1
    b3 = (rev(addr27 & 0x7F, 7) << 1) | 1
    return bytes([b0, b1, b2, b3])


def avlc_frame(info: bytes, src: int = (1 << 24) | 0x4CA2B1, dst: int = (2 << 24) | 0x10C55A,
               ctrl: int = 0x13) -> bytes:
    """dst[4] src[4] ctrl info -- what sits between the flags before the FCS."""
    return icao_addr_bytes(dst) + icao_addr_bytes(src) + bytes([ctrl]) + bytes(info)


def hdlc_payload(frame: bytes) -> bytes:
    """0x7e | bit-stuffed(frame + FCS lo,hi) | 0x7e, zero padded to a byte boundary."""
    crc = fcs16(frame)
    body = frame + bytes([crc & 0xFF, crc >> 8])
    bits: List[int] = []
    flag = [0, 1, 1, 1, 1, 1, 1, 0]
    bits += flag
    ones = 0
    for byte in body:
        for n in range(8):
            b = (byte >> n) & 1
            bits.append(b)
            if b:
                ones += 1
                if ones == 5:
                    bits.append(0)
                    ones = 0
            else:
                ones = 0
    bits += flag
    while len(bits) % 8:
        bits.append(0)
    arr = np.array(bits, dtype=np.uint8).reshape(-1, 8)
    return bytes((arr << np.arange(8, dtype=np.uint8)).sum(axis=1).astype(np.uint8).tolist())


# --------------------------------------------------------------------------- burst framing
def header_bits(length_bits: int) -> List[int]:
    hb = [0] * 25
    for i in range(17):
        hb[3 + i] = (length_bits >> i) & 1
    p = 0
    for n in range(20):
        if hb[n]:
            p ^= HEADER_H[n]
    for i in range(5):
        hb[20 + i] = (p >> (4 - i)) & 1
    return hb


def fec_layout(length_bits: int):
    """(nbrow, nlbyte, rows_with_fec, nfec_last) as the receiver derives them
    (d8psk.c:94-95, 153-161)."""
    nbrow = length_bits // ROW_BITS + 1
    nlbyte = (length_bits % ROW_BITS + 7) // 8
    if nlbyte <= 2:
        return nbrow, nlbyte, nbrow - 1, 6
    if nlbyte <= 30:
        return nbrow, nlbyte, nbrow, 2
    if nlbyte <= 67:
        return nbrow, nlbyte, nbrow, 4
    return nbrow, nlbyte, nbrow, 6


def pn_sequence(nbits: int) -> np.ndarray:
    s = SCRAMBLER_SEED
    out = np.empty(nbits, dtype=np.uint8)
    for i in range(nbits):
        b = (s ^ (s >> 14)) & 1
        s = ((s << 1) | b) & 0xFFFFFFFF
        out[i] = b
    return out


def burst_bits(payload: bytes, corrupt: Optional[dict] = None) -> np.ndarray:
    """All on-air bits after the unique word (header + interleaved data + FEC), scrambled.

    ``corrupt`` maps (row, col) -> xor mask applied AFTER RS encoding (to exercise the
    host-side RS decoder in frame-level tests).
    """
    length_bits = 8 * len(payload)
    nbrow, nlbyte, rows_fec, nfec_last = fec_layout(length_bits)
    rows = np.zeros((nbrow, 255), dtype=np.uint8)
    flat = np.frombuffer(payload, dtype=np.uint8)
    for r in range(nbrow):
        chunk = flat[r * ROW_DATA:(r + 1) * ROW_DATA]
        rows[r, :len(chunk)] = chunk
    for r in range(nbrow):
        rows[r, ROW_DATA:] = rs_parity(rows[r, :ROW_DATA].tolist())
    if corrupt:
        for (r, c), m in corrupt.items():
            rows[r, c] ^= m
    seq: List[int] = []
    last = nbrow - 1
    tx_last = nlbyte if nlbyte else 0  # nlbyte == 0: transmitter sent nothing for that row
    for c in range(ROW_DATA):
        for r in range(nbrow):
            if r == last and c >= tx_last:
                continue
            seq.append(int(rows[r, c]))
    for c in range(6):
        for r in range(rows_fec):
            if r == last and rows_fec == nbrow and c >= nfec_last:
                continue
            seq.append(int(rows[r, ROW_DATA + c]))
    body = np.array(seq, dtype=np.uint8)
    bits = ((body[:, None] >> np.arange(8, dtype=np.uint8)) & 1).reshape(-1)
    bits = np.concatenate([np.array(header_bits(length_bits), dtype=np.uint8), bits])
    bits ^= pn_sequence(len(bits))
    return bits


def received_rows(payload: bytes):
    """(nbrow, nlbyte, data[8*255]) of the msgblk_t an error-free receiver builds from burst_bits(payload)
    (d8psk.c:117-206): bytes that are not transmitted -- the tail of the last row and the FEC bytes the
    shortening drops -- stay zero."""
    length_bits = 8 * len(payload)
    nbrow, nlbyte, rows_fec, nfec_last = fec_layout(length_bits)
    rows = np.zeros((8, 255), dtype=np.uint8)
    flat = np.frombuffer(payload, dtype=np.uint8)
    for r in range(nbrow):
        chunk = flat[r * ROW_DATA:(r + 1) * ROW_DATA]
        rows[r, :len(chunk)] = chunk
        par = rs_parity(rows[r, :ROW_DATA].tolist())
        if r < nbrow - 1:
            rows[r, ROW_DATA:] = par
        elif rows_fec == nbrow:
            rows[r, ROW_DATA:ROW_DATA + nfec_last] = par[:nfec_last]
    return nbrow, nlbyte, rows.tobytes()


def burst_increments(bits: np.ndarray, n_ramp: int = 4) -> np.ndarray:
    """Phase increments (x pi/4) for ramp + reference + unique word + data symbols."""
    pad = (-len(bits)) % 3
    b = np.concatenate([bits, np.zeros(pad, dtype=np.uint8)]).reshape(-1, 3).astype(np.int64)
    tri = (b[:, 0] << 2) | (b[:, 1] << 1) | b[:, 2]
    data_inc = GRAY_LUT[tri]
    return np.concatenate([np.zeros(n_ramp + 1, dtype=np.int64), np.array(UW_INCR, dtype=np.int64), data_inc])


# --------------------------------------------------------------------------- waveform
def rc_pulse(t_over_T: np.ndarray, alpha: float = 0.6) -> np.ndarray:
    x = np.asarray(t_over_T, dtype=np.float64)
    den = 1.0 - (2.0 * alpha * x) ** 2
    sing = np.abs(den) < 1e-9
    den_safe = np.where(sing, 1.0, den)
    h = np.sinc(x) * np.cos(np.pi * alpha * x) / den_safe
    h = np.where(sing, (np.pi / 4.0) * np.sinc(1.0 / (2.0 * alpha)), h)
    return np.where(np.abs(x) > 4.0, 0.0, h)


def modulate_into(acc: np.ndarray, incr: np.ndarray, rate: float, t0: float, amp: float,
                  freq_hz: float, real_lo: bool = False) -> None:
    """Add one burst (complex baseband at ``freq_hz``) into ``acc`` (complex128) in place.

    ``t0`` is the time (s) of symbol 0's pulse centre.  Symbol k sits at t0 + k/10500.
    """
    T = 1.0 / SYMRATE
    phases = np.cumsum(incr) * (np.pi / 4.0)
    a = np.exp(1j * phases)
    nsym = len(a)
    n_lo = max(0, int(math.floor((t0 - 4 * T) * rate)))
    n_hi = min(len(acc), int(math.ceil((t0 + (nsym - 1 + 4) * T) * rate)) + 1)
    if n_hi <= n_lo:
        return
    n = np.arange(n_lo, n_hi, dtype=np.float64)
    u = (n / rate - t0) / T  # symbol-time coordinate
    kc = np.floor(u).astype(np.int64)
    s = np.zeros(len(n), dtype=np.complex128)
    for j in range(-4, 6):
        k = kc + j
        ok = (k >= 0) & (k < nsym)
        kk = np.clip(k, 0, nsym - 1)
        s += np.where(ok, a[kk] * rc_pulse(u - k), 0.0)
    s *= amp * np.exp(2j * np.pi * freq_hz * (n / rate))
    acc[n_lo:n_hi] += s


@dataclasses.dataclass
class Burst:
    chan: int            # index into StreamSpec.fo
    t0: float            # s, pulse centre of the first ramp symbol
    info: bytes          # AVLC info field
    amp: float = 30.0    # LSB at cu8 scale
    cfo: float = 0.0     # Hz carrier offset
    corrupt: Optional[dict] = None
    raw_payload: Optional[bytes] = None  # bypass AVLC/HDLC: bytes handed straight to burst_bits

    def payload(self) -> bytes:
        if self.raw_payload is not None:
            return self.raw_payload
        return hdlc_payload(avlc_frame(self.info, src=(1 << 24) | (0x400000 + self.chan * 0x111 + len(self.info))))

    def n_symbols(self) -> int:
        return len(burst_increments(burst_bits(self.payload())))

    def duration(self) -> float:
        return (self.n_symbols() + 8) / SYMRATE


@dataclasses.dataclass
class StreamSpec:
    rate: int                     # SDRINRATE
    fo: Sequence[int]             # channel offsets from the tuner centre, Hz
    nsamples: int
    bursts: List[Burst]
    noise: float = 1.7            # sigma per component, LSB at cu8 scale
    seed: int = 0


def synth_complex(spec: StreamSpec) -> np.ndarray:
    rng = np.random.default_rng(spec.seed)
    acc = np.zeros(spec.nsamples, dtype=np.complex128)
    for b in spec.bursts:
        incr = burst_increments(burst_bits(b.payload(), b.corrupt))
        modulate_into(acc, incr, float(spec.rate), b.t0, b.amp, spec.fo[b.chan] + b.cfo)
    if spec.noise > 0:
        acc += spec.noise * (rng.standard_normal(spec.nsamples) + 1j * rng.standard_normal(spec.nsamples))
    return acc


def quantise(x: np.ndarray, fmt: str) -> np.ndarray:
    """complex stream (cu8 LSB scale) -> raw interleaved samples in ``fmt``."""
    if fmt == "cu8":
        out = np.empty(2 * len(x), dtype=np.uint8)
        out[0::2] = np.clip(np.rint(x.real + 127.37), 0, 255).astype(np.uint8)
        out[1::2] = np.clip(np.rint(x.imag + 127.37), 0, 255).astype(np.uint8)
        return out
    if fmt == "cs16":
        out = np.empty(2 * len(x), dtype=np.int16)
        out[0::2] = np.clip(np.rint(x.real * 256.0), -32768, 32767).astype(np.int16)
        out[1::2] = np.clip(np.rint(x.imag * 256.0), -32768, 32767).astype(np.int16)
        return out
    if fmt == "cf32":
        out = np.empty(2 * len(x), dtype=np.float32)
        out[0::2] = x.real.astype(np.float32)
        out[1::2] = x.imag.astype(np.float32)
        return out
    if fmt == "f32":  # real sampling (air.c:190): only the real part exists
        return x.real.astype(np.float32)
    raise ValueError(fmt)


def synth_stream(spec: StreamSpec, fmt: str) -> np.ndarray:
    return quantise(synth_complex(spec), fmt)


# --------------------------------------------------------------------------- canned scenarios
DEFAULT_FO_8CH = (-450000, -350000, -250000, -150000, -50000, 100000, 200000, 300000)


def random_scenario(rate: int, fo: Sequence[int], nsamples: int, seed: int,
                    bursts_per_s: float = 6.0, info_max: int = 300, noise: float = 1.7,
                    amp_range=(8.0, 60.0), cfo_max: float = 400.0,
                    info_choices: Optional[Sequence[int]] = None) -> StreamSpec:
    """Poisson-ish non-overlapping bursts per channel."""
    rng = np.random.default_rng(seed ^ 0x5EED)
    dur = nsamples / rate
    bursts: List[Burst] = []
    for c in range(len(fo)):
        t = 0.002 + rng.exponential(1.0 / bursts_per_s) * 0.3
        while True:
            if info_choices is not None:
                n_info = int(rng.choice(info_choices))
            else:
                n_info = int(rng.integers(1, info_max + 1))
            info = bytes(rng.integers(0, 256, n_info, dtype=np.uint8).tolist())
            b = Burst(chan=c, t0=t, info=info, amp=float(rng.uniform(*amp_range)),
                      cfo=float(rng.uniform(-cfo_max, cfo_max)))
            d = b.duration()
            if t + d + 0.001 > dur:
                break
            bursts.append(b)
            t += d + 0.0015 + rng.exponential(1.0 / bursts_per_s)
    return StreamSpec(rate=rate, fo=tuple(fo), nsamples=nsamples, bursts=bursts, noise=noise, seed=seed)


# --------------------------------------------------------------------------- command line
def _cli(argv=None) -> int:
    """python -m vdlm2dec_amd.synth -- write a synthetic VDL2 recording and its ground truth.

    The reference has no transmitter and no file input: this is the tool that makes every
    fixture, benchmark stream and BER curve of this repository reproducible from a seed."""
    import argparse
    import json
    ap = argparse.ArgumentParser(prog="python -m vdlm2dec_amd.synth", description=_cli.__doc__)
    ap.add_argument("out", help="raw IQ file to write (interleaved I,Q in --fmt; real samples for f32)")
    ap.add_argument("--fmt", default="cu8", choices=["cu8", "cs16", "cf32", "f32"])
    ap.add_argument("--rate", type=int, default=2_000_000, help="SDRINRATE (2000000 rtl; 5000000/6000000 airspy; 10000000)")
    ap.add_argument("--fo", type=int, nargs="+", default=list(DEFAULT_FO_8CH), help="channel offsets from the tuner centre, Hz")
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--bursts-per-s", type=float, default=4.0, help="arrival rate per channel behind each burst")
    ap.add_argument("--info-max", type=int, default=240, help="AVLC info field: 1..N random bytes")
    ap.add_argument("--amp", type=float, nargs=2, default=(8.0, 60.0), metavar=("MIN", "MAX"), help="amplitude, LSB at cu8 scale")
    ap.add_argument("--noise", type=float, default=1.7, help="AWGN sigma per component, LSB at cu8 scale")
    ap.add_argument("--cfo", type=float, default=400.0, help="carrier offsets uniform in +-CFO Hz")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--truth", help="JSON file: every burst sent (channel, start, AVLC info, the frame out() must see, msgblk rows)")
    a = ap.parse_args(argv)
    ns = (int(a.seconds * a.rate) + 32767) // 32768 * 32768      # whole RTL hand-off blocks (vdlm2.h:35)
    spec = random_scenario(a.rate, a.fo, ns, seed=a.seed, bursts_per_s=a.bursts_per_s, info_max=a.info_max,
                           noise=a.noise, amp_range=tuple(a.amp), cfo_max=a.cfo)
    synth_stream(spec, a.fmt).tofile(a.out)
    if a.truth:
        tr = []
        for b in sorted(spec.bursts, key=lambda b: b.t0):
            nbrow, nlbyte, rows = received_rows(b.payload())
            tr.append(dict(chan=b.chan, fo=int(a.fo[b.chan]), t0=b.t0, amp=b.amp, cfo=b.cfo, info=b.info.hex(),
                           payload=b.payload().hex(), nbrow=nbrow, nlbyte=nlbyte, symbols=b.n_symbols()))
        json.dump(dict(rate=a.rate, fmt=a.fmt, fo=list(a.fo), nsamples=ns, noise=a.noise, seed=a.seed, bursts=tr), open(a.truth, "w"))
    print(f"{a.out}: {ns} samples @ {a.rate} S/s {a.fmt}, {len(a.fo)} channels, {len(spec.bursts)} bursts")
    return 0


if __name__ == "__main__":
    raise SystemExit(_cli())
