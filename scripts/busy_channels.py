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
import ctypes as C
from vdlm2dec_amd import lib as _lib
with Receiver(2_000_000, plan_channels(bench.FC, fos), fmt="cs16", max_push=n, max_bursts=1 << 17) as rx:
    prev = 0
    buf = (_lib.BurstT * (1 << 16))()
    for p in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rx.push_device(big.data_ptr(), n, 0)
        got = 0
        while True:                     # no Python objects: the time is the library's (round 2 timed 2 ms of Burst() construction per push here)
            k = rx.poll_raw(buf, 1 << 16)
            got += k
            if k < (1 << 16):
                break
        dt = time.perf_counter() - t0
        st = rx.stats()
        print("push %d: %.2f ms, %d bursts, serial samples +%d" % (p, dt * 1e3, got, st["serial_samples"] - prev), flush=True)
        prev = st["serial_samples"]
        if p == 3 and os.environ.get("VDL2GPU_DEBUG_COUNTERS"):
            rx.debug_counters(32, reset=True)
    # pipelined, like bench.py: push, take what is ready, drain at the end
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tot = 0
    for p in range(8):
        rx.push_device(big.data_ptr(), n, 0)
        tot += rx.poll_ready_raw(buf, 1 << 16)
    while True:
        k = rx.poll_raw(buf, 1 << 16)
        tot += k
        if k < (1 << 16):
            break
    dt = time.perf_counter() - t0
    print("pipelined: %.2f ms per push, %.1f GS/s, %d bursts per push" % (dt / 8 * 1e3, 8 * n / dt / 1e9, tot // 8))
    if os.environ.get("VDL2GPU_DEBUG_COUNTERS"):
        d = rx.debug_counters(64)
        n = max(1, d[20])
        names = ["stage-in", "fir+unit", "barrier", "steps", "screen1", "barrier", "screen2/3", "flush(in tile)", "-", "barrier", "nwl", "passes", "all tiles", "ndl final", "final flush", "instants"]
        for base, what in ((32, "probe"), (48, "region")):
            print(what + ": " + ", ".join("%s %d" % (names[i], d[base + i]) for i in range(16) if d[base + i]))
        print("K2c per call (100 MHz ticks -> us): load %.1f tables %.1f walk+serial %.1f publish %.1f; calls %d; per call: cands %.0f visited %.0f "
              "replays %.2f nonsteady %.2f serial samples %.0f" % (d[16] / n / 100, d[17] / n / 100, d[18] / n / 100, d[19] / n / 100, d[20],
                                                            d[27] / n, d[28] / n, d[25] / n, d[26] / n, d[29] / n))
        print("K2b: clusters %d, ring %.1f us, rest %.1f us per sampled cluster; triggers/cluster %.2f evals/cluster %.1f steady %.2f rejects %.2f" % (
            d[2] * 16, d[0] / max(1, d[2]) / 100, d[1] / max(1, d[2]) / 100, d[3] / max(1, d[2]), d[4] / max(1, d[2]), d[6] / max(1, d[2]), d[7] / max(1, d[2])))
