#!/usr/bin/env python3
"""Frame success / header success vs Es/N0 of the receive path, GPU and oracle side by side.

    python scripts/ber_curve.py [--esn0 14 16 18 20 22 24 27 30 35] [--bursts 160] [--out profiles/r02_ber_curve.json]
    python scripts/ber_curve.py --oracle-only ...        (no GPU: the curve of the CPU restatement alone)

Every point: `--bursts` bursts (random AVLC info 1..120 bytes, +-400 Hz carrier offset, random fractional start) on 4
channels of a 2 MS/s cs16 stream at fixed amplitude, AWGN set for the requested Es/N0 = A^2 * SDRINRATE / (2 sigma^2 *
10500).  Reported per point: headers accepted (msgblk_t records with the sent nbrow/nlbyte), frames that pass RS + FCS
and equal the sent frame, mean byte errors per 100 sliced bytes before RS -- and, the point of the exercise, that the
GPU's records are the oracle's records, byte for byte, at every noise level (decisions near their thresholds included)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vdlm2dec_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker)

FC = 136_975_000
FO = (-450_000, -150_000, 100_000, 300_000)
RATE = 2_000_000
AMP = 20.0


def scenario(nbursts, esn0_db, seed):
    rng = np.random.default_rng(seed)
    sigma = AMP * np.sqrt(RATE / (2.0 * 10500.0 * 10 ** (esn0_db / 10.0)))
    bursts, t = [], [0.002 + rng.uniform(0, 0.001) for _ in FO]
    for i in range(nbursts):
        c = i % len(FO)
        b = synth.Burst(chan=c, t0=t[c], info=bytes(rng.integers(0, 256, int(rng.integers(1, 121)), dtype=np.uint8).tolist()),
                        amp=AMP, cfo=float(rng.uniform(-400, 400)))
        bursts.append(b)
        t[c] += b.duration() + 0.002 + rng.uniform(0, 0.001)
    ns = (int((max(t) + 0.004) * RATE) + 32767) // 32768 * 32768
    return synth.StreamSpec(rate=RATE, fo=FO, nsamples=ns, bursts=bursts, noise=float(sigma), seed=seed), float(sigma)


def score(spec, recs):
    """recs: list of (chn, nbrow, nlbyte, data bytes)."""
    sent = {}
    for b in spec.bursts:
        nbrow, nlbyte, rows = synth.received_rows(b.payload())
        fr = O.frames_of_block(nbrow, nlbyte, rows)
        sent.setdefault(b.chan, []).append((nbrow, nlbyte, rows, fr[0] if fr else None))
    hdr_ok = frames_ok = byte_err = byte_n = 0
    for c, lst in sent.items():
        mine = [r for r in recs if r[0] == c]
        used = set()
        for nbrow, nlbyte, rows, frame in lst:
            for j, r in enumerate(mine):
                if j in used or (r[1], r[2]) != (nbrow, nlbyte):
                    continue
                got = np.frombuffer(r[3], np.uint8)
                want = np.frombuffer(rows, np.uint8)
                live = want != 0
                e = int((got[live] != want[live]).sum())
                if e > 0.25 * max(1, live.sum()):
                    continue            # another burst of the same geometry
                used.add(j)
                hdr_ok += 1
                byte_err += e
                byte_n += int(live.sum())
                fr = O.frames_of_block(r[1], r[2], r[3])
                frames_ok += int(bool(fr) and fr[0] == frame)
                break
    return dict(sent=len(spec.bursts), headers_ok=hdr_ok, frames_ok=frames_ok,
                byte_errors_per_100=(100.0 * byte_err / byte_n) if byte_n else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--esn0", type=float, nargs="+", default=[14, 16, 18, 20, 22, 24, 27, 30, 35])
    ap.add_argument("--bursts", type=int, default=160)
    ap.add_argument("--oracle-only", action="store_true")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    pts = []
    for i, e in enumerate(a.esn0):
        spec, sigma = scenario(a.bursts, e, seed=9000 + i)
        raw = synth.synth_stream(spec, "cs16")
        ob = O.run_oracle(raw, "cs16", RATE, FO, FC)
        orec = sorted((b.chn, b.nbrow, b.nlbyte, b.data) for b in ob)
        pt = dict(esn0_db=e, sigma_lsb=sigma, oracle=score(spec, orec), oracle_records=len(orec))
        if not a.oracle_only:
            from vdlm2dec_amd.demod import Receiver, plan_channels
            with Receiver(RATE, plan_channels(FC, FO), fmt="cs16", max_push=spec.nsamples) as rx:
                gb = rx.run(raw)
            grec = sorted((b.chn, b.nbrow, b.nlbyte, b.data) for b in gb)
            pt["gpu"] = score(spec, grec)
            pt["gpu_records"] = len(grec)
            pt["gpu_equals_oracle"] = grec == orec and \
                sorted((b.chn, b.trig_dec, int(np.float32(b.df).view(np.uint32))) for b in gb) == \
                sorted((b.chn, b.trig_dec, int(np.float32(b.df).view(np.uint32))) for b in ob)
        pts.append(pt)
        s = pt.get("gpu", pt["oracle"])
        print(f"Es/N0 {e:5.1f} dB  sigma {sigma:6.2f}  sent {s['sent']:4d}  headers {s['headers_ok']:4d}  frames {s['frames_ok']:4d}  "
              f"byte errors/100 {s['byte_errors_per_100'] if s['byte_errors_per_100'] is not None else float('nan'):6.3f}"
              + ("" if a.oracle_only else f"  GPU==oracle {pt['gpu_equals_oracle']}"))
    res = dict(rate=RATE, fo=list(FO), amp=AMP, bursts_per_point=a.bursts, points=pts)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)
    if not a.oracle_only and not all(p["gpu_equals_oracle"] for p in pts):
        sys.exit("GPU and oracle differ")


if __name__ == "__main__":
    main()
