"""dev aid: randomised parity soak of the PIPELINED device path (three pushes in flight, tails on the payload stream): bench.py's
leg with random load, push length, recordings and step count, every burst against the oracle.  scripts/soak.py <seconds> [seed]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from vdlm2dec_amd import synth

t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 300)
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
ok = bad = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    bps = float(rng.choice([2.0, 4.0, 8.0, 15.0, 30.0, 60.0]))
    ntiles = int(rng.choice([2, 4, 8, 16]))
    steps = int(rng.integers(3, 9))
    warm = int(rng.integers(1, 4))
    nstr = int(rng.choice([1, 1, 1, 2]))
    os.environ.pop("VDL2GPU_REPAIR_ROUNDS", None)
    if rng.integers(0, 4) == 0:
        os.environ["VDL2GPU_REPAIR_ROUNDS"] = str(int(rng.integers(1, 4)))
    r = bench.run_leg("soak", "soak", 0, 2_000_000, synth.DEFAULT_FO_8CH, "cs16", nstr, ntiles, bps, steps=steps, warmup=warm, seed0=seed * 7)
    good = r["parity"]["equal"]
    ok += good
    bad += not good
    print("seed %d: %s  bps %g tiles %d steps %d+%d streams %d rounds %s: %d bursts, %.0f MS/s, repairs %d redos %d%s" % (
        seed, "ok " if good else "BAD", bps, ntiles, warm, steps, nstr, os.environ.get("VDL2GPU_REPAIR_ROUNDS", "-"), r["parity"]["bursts_checked"],
        r["value"] or 0, r["repairs"], r["serial_redos"], "" if good else "  " + json.dumps(r["parity"].get("mismatch"))), flush=True)
    seed += 1
print("soak: %d ok, %d BAD" % (ok, bad))
