"""Seeded synthetic recordings shared by the CPU and GPU tests and by tests/golden/make_golden.py."""
from __future__ import annotations

import numpy as np

from vdlm2dec_amd import synth

FC = 136_975_000
FO8 = synth.DEFAULT_FO_8CH


def _pad(n, q=32768):
    return (n + q - 1) // q * q


def regimes(rate=2_000_000, fo=(-50_000, 250_000), seed=3, infos=(1, 2, 3, 28, 31, 60, 66, 70, 200, 247, 250, 497),
            noise=1.7, gap=0.002):
    """One burst per FEC-shortening regime / row count, alternating channels."""
    rng = np.random.default_rng(seed)
    bursts, t = [], 0.003
    for i, n in enumerate(infos):
        b = synth.Burst(chan=i % len(fo), t0=t, info=bytes(rng.integers(0, 256, n, dtype=np.uint8).tolist()),
                        amp=float(rng.uniform(8, 60)), cfo=float(rng.uniform(-400, 400)))
        bursts.append(b)
        t += b.duration() + gap + rng.uniform(0, 1e-3)
    ns = _pad(int((t + 0.005) * rate))
    return synth.StreamSpec(rate=rate, fo=tuple(fo), nsamples=ns, bursts=bursts, noise=noise, seed=seed)


# config 3 (BASELINE.json configs[2]): 8 channels at 10 MS/s (SDRCLK 2500, LO table 400), spread over +-2.3 MHz on the
# 25 kHz grid; and the Airspy shape (air.c:134-138): real samples at 5 MS/s, channels above the mixer centre
FO8_10MS = (-2_250_000, -1_750_000, -1_250_000, -300_000, 475_000, 1_000_000, 1_525_000, 2_300_000)
FO8_AIR_5MS = (150_000, 425_000, 700_000, 1_000_000, 1_300_000, 1_575_000, 1_850_000, 2_200_000)


def eight_channels(rate=2_000_000, seed=8, dur=0.16, info=(3, 17, 40, 64, 90, 130, 5, 75), fo=None):
    """Config-2 shape: 8 channels, bursts overlapping in time on different channels."""
    rng = np.random.default_rng(seed)
    bursts = []
    for c in range(8):
        b = synth.Burst(chan=c, t0=0.002 + 0.004 * c + rng.uniform(0, 1e-3),
                        info=bytes(rng.integers(0, 256, info[c], dtype=np.uint8).tolist()),
                        amp=float(rng.uniform(10, 40)), cfo=float(rng.uniform(-400, 400)))
        bursts.append(b)
        # a second burst on the same channel shortly after the first one ends
        b2 = synth.Burst(chan=c, t0=b.t0 + b.duration() + 0.0012 + rng.uniform(0, 2e-3),
                         info=bytes(rng.integers(0, 256, 6 + c, dtype=np.uint8).tolist()),
                         amp=float(rng.uniform(10, 40)), cfo=float(rng.uniform(-400, 400)))
        if b2.t0 + b2.duration() < dur - 0.002:
            bursts.append(b2)
    return synth.StreamSpec(rate=rate, fo=tuple(fo) if fo else FO8, nsamples=_pad(int(dur * rate)), bursts=bursts, noise=1.6, seed=seed)


def single_short(rate, fo, seed=5, info_len=10, amp=40.0, blocks=4, t0=0.002):
    rng = np.random.default_rng(seed)
    b = synth.Burst(chan=0, t0=t0, info=bytes(rng.integers(0, 256, info_len, dtype=np.uint8).tolist()), amp=amp,
                    cfo=float(rng.uniform(-300, 300)))
    return synth.StreamSpec(rate=rate, fo=(fo,), nsamples=32768 * blocks, bursts=[b], noise=1.5, seed=seed)


def back_to_back(rate=2_000_000, fo=(100_000,), seed=21):
    """Bursts separated by only a few symbols: the next sync is searched with a stale phase ring
     and with perr=p2err=500."""
    rng = np.random.default_rng(seed)
    bursts, t = [], 0.002
    for i, gap_sym in enumerate((2.0, 6.5, 11.25, 18.0, 30.0, 3.3)):
        b = synth.Burst(chan=0, t0=t, info=bytes(rng.integers(0, 256, 4 + 3 * i, dtype=np.uint8).tolist()),
                        amp=float(rng.uniform(20, 50)), cfo=float(rng.uniform(-300, 300)))
        bursts.append(b)
        t += (b.n_symbols() + gap_sym) / synth.SYMRATE
    return synth.StreamSpec(rate=rate, fo=tuple(fo), nsamples=_pad(int((t + 0.01) * rate)), bursts=bursts, noise=1.5,
                            seed=seed)


def odd_headers(rate=2_000_000, fo=(-150_000,), seed=33):
    """Headers the receiver rejects or mis-sizes: len < 96 bits (d8psk.c:97), nbrow > 8 (d8psk.c:103),
    len % 1992 == 0, plus one long 8-row burst."""
    rng = np.random.default_rng(seed)
    bursts, t = [], 0.002
    payloads = [
        bytes(rng.integers(0, 256, 8, dtype=np.uint8).tolist()),       # len 64 bits  -> reject (too short)
        bytes(rng.integers(0, 256, 249, dtype=np.uint8).tolist()),     # len 1992     -> nlbyte == 0 edge
        synth.hdlc_payload(synth.avlc_frame(bytes(rng.integers(0, 256, 30, dtype=np.uint8).tolist()))),
        bytes(rng.integers(0, 256, 1900, dtype=np.uint8).tolist()),    # 8 rows (len 15200)
        synth.hdlc_payload(synth.avlc_frame(bytes(rng.integers(0, 256, 12, dtype=np.uint8).tolist()))),
    ]
    for pl in payloads:
        b = synth.Burst(chan=0, t0=t, info=b"", amp=35.0, cfo=float(rng.uniform(-200, 200)), raw_payload=pl)
        bursts.append(b)
        t += b.duration() + 0.003
    return synth.StreamSpec(rate=rate, fo=tuple(fo), nsamples=_pad(int((t + 0.03) * rate)), bursts=bursts, noise=1.5,
                            seed=seed)


def corrupted(rate=2_000_000, fo=(200_000,), seed=44):
    """Byte errors injected after RS encoding: 1..3 per row must be repaired by the host path,
    4 must not (frame-level parity with the reference's rs.c)."""
    rng = np.random.default_rng(seed)
    bursts, t = [], 0.002
    for nerr in (1, 2, 3, 4, 3):
        info = bytes(rng.integers(0, 256, 120, dtype=np.uint8).tolist())
        cols = rng.choice(100, size=nerr, replace=False)
        b = synth.Burst(chan=0, t0=t, info=info, amp=40.0, cfo=0.0,
                        corrupt={(0, int(c)): int(rng.integers(1, 256)) for c in cols})
        bursts.append(b)
        t += b.duration() + 0.002
    return synth.StreamSpec(rate=rate, fo=tuple(fo), nsamples=_pad(int((t + 0.005) * rate)), bursts=bursts, noise=1.2,
                            seed=seed)
