"""dev aid: duration of a K2b wavefront's first, second, third and later clusters (100 MHz ticks) -- how much of the
kernel is instruction-cache start-up.  VDL2GPU_DEBUG_COUNTERS=1 python scripts/k2b_cold.py"""
import os, sys
os.environ["VDL2GPU_DEBUG_COUNTERS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import scenarios as S
from vdlm2dec_amd import synth
from vdlm2dec_amd.demod import Receiver, plan_channels

n = 16_800_000
spec = synth.random_scenario(2_000_000, S.FO8, n, seed=5, bursts_per_s=4.0 * 8, info_max=200)
raw = synth.synth_stream(spec, "cs16")
with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cs16", max_push=n) as rx:
    for rep in range(4):
        rx.push(raw)
        got = rx.poll()
        c = rx.debug_counters(16, reset=True)
        print("push", rep, "bursts", len(got), "clusters", c[12] + c[13] + c[14] + c[15],
              "mean ticks by ordinal:", [round(c[8 + i] / max(1, c[12 + i]), 1) for i in range(4)], "counts", [int(c[12 + i]) for i in range(4)])
