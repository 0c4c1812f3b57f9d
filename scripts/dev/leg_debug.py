"""dev aid: one busy leg of bench.py (steps / warmup as the test uses them), printing what differs from the oracle if anything does"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from vdlm2dec_amd import synth
bps = float(sys.argv[1]) if len(sys.argv) > 1 else 15.0
r = bench.run_leg("busy", "test", 0, 2_000_000, synth.DEFAULT_FO_8CH, "cs16", 1, 16, bps, steps=4, warmup=2, seed0=77)
print(json.dumps({k: v for k, v in r.items() if k != "workload"}, indent=1))
