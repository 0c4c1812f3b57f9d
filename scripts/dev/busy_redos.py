"""dev aid: how often does the verify pass fail on busy channels?  One handle per recording, pushes of 16 tiles; serial redos per push
(after the first one the library schedules a repair round, which takes care of the next)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from vdlm2dec_amd import synth, lib as _lib
from vdlm2dec_amd.demod import Receiver, plan_channels

bps = float(sys.argv[1]) if len(sys.argv) > 1 else 15.0
fos = synth.DEFAULT_FO_8CH
buf = (_lib.BurstT * (1 << 16))()
for seed in (77, 1077, 2077, 3077, 4077, 5077):
    spec, raw = bench.make_tile(seed, "cs16", 2_000_000, fos, bps)
    big = torch.from_numpy(np.tile(raw, 16)).cuda()
    n = big.numel() // 2
    with Receiver(2_000_000, plan_channels(bench.FC, fos), fmt="cs16", max_push=n, max_bursts=1 << 17) as rx:
        prev, out, ms = 0, [], []
        for p in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rx.push_device(big.data_ptr(), n, 0)
            while rx.poll_raw(buf, 1 << 16) == (1 << 16):
                pass
            ms.append(round((time.perf_counter() - t0) * 1e3, 2))
            st = rx.stats()
            out.append(st["serial_redos"] - prev)
            prev = st["serial_redos"]
        print("seed", seed, "bursts/push", st["bursts"] // 6, "redos per push", out, "ms", ms, flush=True)
