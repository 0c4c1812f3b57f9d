"""dev aid: pushes of 67.2 MS on 8 busy channels (bursts per second and channel as argument): time per push and what
goes through the serial machine, push by push (the library shortens the parts it cuts pushes into when tables overflow)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from vdlm2dec_amd import synth
from vdlm2dec_amd.demod import Receiver, plan_channels

bps = float(sys.argv[1]) if len(sys.argv) > 1 else 15.0
fos = synth.DEFAULT_FO_8CH
spec = synth.random_scenario(2_000_000, fos, bench.TILE, seed=77, bursts_per_s=bps, info_max=240)
raw = synth.synth_stream(spec, "cs16")
big = torch.from_numpy(np.tile(raw, 16)).cuda()
n = big.numel() // 2
with Receiver(2_000_000, plan_channels(bench.FC, fos), fmt="cs16", max_push=n) as rx:
    prev = 0
    for p in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rx.push_device(big.data_ptr(), n, 0)
        got = rx.poll()
        dt = time.perf_counter() - t0
        st = rx.stats()
        print("push %d: %.2f ms, %d bursts, serial samples +%d" % (p, dt * 1e3, len(got), st["serial_samples"] - prev), flush=True)
        prev = st["serial_samples"]
