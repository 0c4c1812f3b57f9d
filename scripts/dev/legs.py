"""dev aid: run bench.py's `configs` legs alone: python scripts/dev/legs.py [busy15 busy30 c3 c4 live]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from vdlm2dec_amd import synth
fo2 = synth.DEFAULT_FO_8CH
which = sys.argv[1:] or ["busy15", "busy30"]
W = bench.CONFIGS
for w in which:
    if w == "busy15":
        r = bench.run_leg(w, "x", 0, 2_000_000, fo2, "cs16", 1, 16, 15.0, steps=8, warmup=3, seed0=77)
    elif w == "busy30":
        r = bench.run_leg(w, "x", 0, 2_000_000, fo2, "cs16", 1, 16, 30.0, steps=8, warmup=3, seed0=77)
    elif w == "c3":
        r = bench.run_leg(w, "x", 0, 10_000_000, bench.FO8_10MS, "cs16", 1, 64, 4.0, steps=8, warmup=3, seed0=1234)
    elif w == "c4":
        r = bench.run_leg(w, "x", 0, 2_000_000, fo2, "cs16", 8, 16, 4.0, steps=6, warmup=3, seed0=1234)
    elif w == "live":
        r = bench.live_leg(0)
    else:
        continue
    print(w, json.dumps({a: b for a, b in r.items() if a in ("value", "ms_per_step", "first_push_ms", "max_push_ms", "serial_samples_frac", "latency_ms", "bursts_per_step") or (a == "parity" and not b["equal"])}), flush=True)
