"""dev aid: a recording with an event in one class only (bench.make_tile seed 1077 / 5077), repair rounds 0..4: repairs, serial redos, time"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 2:
    import numpy as np
    import bench
    from vdlm2dec_amd import synth
    from vdlm2dec_amd.demod import Receiver, plan_channels
    seed = int(sys.argv[1])
    spec, raw = bench.make_tile(seed, "cs16", 2_000_000, synth.DEFAULT_FO_8CH, 4.0)
    big = np.tile(raw, int(sys.argv[2]))
    with Receiver(2_000_000, plan_channels(bench.FC, synth.DEFAULT_FO_8CH), fmt="cs16", max_push=big.size // 2) as rx:
        for p in range(3):
            t0 = time.perf_counter()
            rx.push(big)
            n = len(rx.poll())
            dt = time.perf_counter() - t0
            st = rx.stats()
            print("  rounds=%s push %d: %.2f ms, %d bursts, repairs %d, serial redos %d" % (os.environ.get("VDL2GPU_REPAIR_ROUNDS", "default"), p, dt * 1e3, n, st["repairs"], st["serial_redos"]), flush=True)
else:
    for seed in (1077, 5077):
        print("seed", seed)
        for r in ("", "0", "1", "2", "3"):
            env = dict(os.environ)
            if r:
                env["VDL2GPU_REPAIR_ROUNDS"] = r
            subprocess.run([sys.executable, __file__, str(seed), "2"], env=env)
