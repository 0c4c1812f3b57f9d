"""Generate the golden vectors in this directory from the REAL reference.

Run in the build container (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

For every case it writes
    <case>.npz   the raw IQ recording (data), compressed
    <case>.json  what the reference built from its own sources (oracle/_ref/ref_rtl,
                 ref_air) produced for it, per channel:
                   blocks  every msgblk_t handed to decodeVdlm2 (d8psk.c:201)
                   frames  every CRC-clean frame handed to out() (vdlm2.c:61)
                   phase_sha256 / n_phase  digest of EVERY filteredphase() result
                                           (atan2f return bits, call order)
                   head_sha256 / n_head    digest of every viterbi_add soft bit
The vectors are data only; no reference source travels with them.
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

from oracle import oracle as O  # noqa: E402
from vdlm2dec_amd import synth  # noqa: E402
import scenarios as S  # noqa: E402

CASES = {
    # name: (spec, fmt, quirk)
    "regimes_cu8_2ms": (S.regimes(), "cu8", 0),
    "regimes_cu8_2ms_quirk": (S.regimes(), "cu8", 1),          # rtl.c:291 off-by-one variant, same IQ
    "eight_cs16_2ms": (S.eight_channels(), "cs16", 0),
    "air_f32_5ms": (S.single_short(5_000_000, 375_000, seed=6, info_len=12, blocks=5), "f32", 0),
    "cs16_10ms": (S.single_short(10_000_000, -1_250_000, seed=7, info_len=20, blocks=6), "cs16", 0),
    "short_cf32_2ms": (S.single_short(2_000_000, 100_000, seed=9, info_len=8, blocks=2), "cf32", 0),
    # BASELINE.json configs[2]: 8 channels @ 10 MS/s (SDRCLK 2500)
    "eight_cs16_10ms": (S.eight_channels(rate=10_000_000, seed=10, dur=0.05, info=(3, 9, 5, 12, 7, 4, 6, 8), fo=S.FO8_10MS), "cs16", 0),
}
SHARE_IQ = {"regimes_cu8_2ms_quirk": "regimes_cu8_2ms"}


def main():
    assert O.build_ref(), "needs /root/reference"
    for name, (spec, fmt, quirk) in CASES.items():
        raw = synth.synth_stream(spec, fmt)
        if name not in SHARE_IQ:
            np.savez_compressed(os.path.join(HERE, name + ".npz"), raw=raw)
        meta = dict(name=name, fmt=fmt, rate=spec.rate, fo=list(spec.fo), fc=S.FC, quirk=quirk,
                    nsamples=spec.nsamples, iq=SHARE_IQ.get(name, name), channels=[])
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "iq.raw")
            raw.tofile(p)
            for c, fo in enumerate(spec.fo):
                blocks, frames, taps = O.run_ref(p, fmt, spec.rate, fo, S.FC + fo, os.path.join(td, "out.txt"), quirk,
                                                 os.path.join(td, "taps.bin"))
                ph = taps[taps["t"] == 1]["c"].astype("<f4").tobytes()
                hd = taps[taps["t"] == 3]["a"].astype("<f4").tobytes()
                meta["channels"].append(dict(
                    chn=c, fo=fo,
                    blocks=[dict(nbrow=b["nbrow"], nlbyte=b["nlbyte"], df_bits=b["df_bits"], tv=b["tv"],
                                 ppm_bits=int(np.float32(b["ppm"]).view(np.uint32)), data=b["data"].hex())
                            for b in blocks],
                    frames=[f["frame"].hex() for f in frames],
                    n_phase=len(ph) // 4, phase_sha256=hashlib.sha256(ph).hexdigest(),
                    n_head=len(hd) // 4, head_sha256=hashlib.sha256(hd).hexdigest()))
                print(name, "ch", c, "blocks", len(blocks), "frames", len(frames), "phases", len(ph) // 4)
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(meta, f)


if __name__ == "__main__":
    main()
