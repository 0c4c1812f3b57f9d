"""dev aid: where does a long push fall off the parallel tables?  python scripts/long_push.py <rate> <tiles...>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from vdlm2dec_amd.demod import Receiver, plan_channels

rate = int(sys.argv[1])
fos = tuple(int(f * rate / 2_000_000) // 25000 * 25000 for f in bench.synth_default()) if rate != 2_000_000 else bench.synth_default()
spec, raw = bench.make_tile(1234, "cs16", rate, fos)
for tiles in [int(x) for x in sys.argv[2:]]:
    big = np.tile(raw, tiles)
    with Receiver(rate, plan_channels(bench.FC, fos), fmt="cs16", max_push=big.size // 2) as rx:
        for p in range(2):
            rx.push(big)
            got = rx.poll()
        st = rx.stats()
        if os.environ.get("LONG_PUSH_DEBUG"):
            import ctypes as C
            fail = np.zeros(8, np.int32)
            rx.L.vdl2gpu_debug_fail(rx.h, fail.ctypes.data_as(C.c_void_p), 8)
            print("   per channel candidates:", [len(rx.debug_cands(0, ch)) for ch in range(8)], "fail:", [hex(int(x)) for x in fail])
        print("tiles", tiles, "dec samples/push", big.size // 2 * 21 // (rate // 4000), "bursts", len(got), {k: st[k] for k in ("candidates", "serial_redos", "serial_samples")}, flush=True)
