"""dev aid: which of the bench's per-stream tiles make the verify pass fail, and what the repair rounds do about it.
One Receiver per seed, the tile repeated `tiles` times per push, a few pushes; prints serial redos per push."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from vdlm2dec_amd import synth
from vdlm2dec_amd.demod import Receiver, plan_channels

tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 4
fos = synth.DEFAULT_FO_8CH
for seed in range(1234, 1234 + 8):
    spec, raw = bench.make_tile(seed, "cs16", 2_000_000, fos)
    big = np.tile(raw, tiles)
    with Receiver(2_000_000, plan_channels(bench.FC, fos), fmt="cs16", max_push=big.size // 2) as rx:
        prev = 0
        out = []
        for p in range(8):
            rx.push(big)
            rx.poll()
            st = rx.stats()
            out.append(st["serial_redos"] - prev)
            prev = st["serial_redos"]
        print("seed", seed, "redos per push", out, "serial_samples", st["serial_samples"])
