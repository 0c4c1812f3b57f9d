"""dev aid: the scan kernels' phase counters (VDL2GPU_DEBUG_COUNTERS=1) for synchronous pushes against pipelined ones:
why does the region scan take 96 us with two pushes in flight and 53 us alone?"""
import os, sys
os.environ["VDL2GPU_DEBUG_COUNTERS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from vdlm2dec_amd import synth, lib as _lib
from vdlm2dec_amd.demod import Receiver, plan_channels

fos = synth.DEFAULT_FO_8CH
spec, raw = bench.make_tile(1234, "cs16", 2_000_000, fos)
big = torch.from_numpy(np.tile(raw, 16)).cuda()
n = big.numel() // 2
names = ["stage-in", "fir+unit", "barrier", "steps", "screen1", "barrier", "screen2/3", "flush(in tile)", "-", "barrier", "nwl", "passes",
         "all tiles/wait", "ndl final/park", "final flush", "instants"]
buf = (_lib.BurstT * (1 << 16))()
with Receiver(2_000_000, plan_channels(bench.FC, fos), fmt="cs16", max_push=n, max_bursts=1 << 17) as rx:
    for _ in range(4):
        rx.push_device(big.data_ptr(), n, 0)
        rx.poll_raw(buf, 1 << 16)
    for mode in ("synchronous", "pipelined"):
        rx.debug_counters(64, reset=True)
        for _ in range(8):
            rx.push_device(big.data_ptr(), n, 0)
            if mode == "synchronous":
                rx.poll_raw(buf, 1 << 16)
            else:
                rx.poll_ready_raw(buf, 1 << 16)
        rx.poll_raw(buf, 1 << 16)
        d = rx.debug_counters(64)
        print(mode)
        for base, what in ((32, "probe"), (48, "region")):
            print("  " + what + ": " + ", ".join("%s %d" % (names[i], d[base + i]) for i in range(16) if d[base + i]))
