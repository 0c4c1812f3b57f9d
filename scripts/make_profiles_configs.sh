#!/bin/bash
# profiles/r02_config{3,4}_*: bench line and rocprofv3 kernel stats of the two other named workloads
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/profiles
mkdir -p $OUT
here=$(pwd)
for c in 3 4; do
  rm -rf /tmp/prc_$c
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prc_$c -- python $here/bench.py --config $c --no-cpu --no-ring > /tmp/prc_$c.log 2>&1 )
  cp $(find /tmp/prc_$c -name "*kernel_stats.csv" | head -1) $OUT/r02_config${c}_kernel_stats.csv
  python $here/bench.py --config $c --no-cpu --no-ring 2>/dev/null | tail -1 > $OUT/r02_config${c}_bench_line.json
  python - "$OUT" "$c" <<'PY'
import sys, json, csv
out, c = sys.argv[1], sys.argv[2]
d = json.load(open(f"{out}/r02_config{c}_bench_line.json"))
print(f"config {c}:", round(d["value"]), "MS/s", round(d["ms_per_step"], 4), "ms/step; roofline", round(d["roofline"]["frac"], 3), d["roofline"]["kernel"], round(d["roofline"]["avg_launch_ms"], 4), "parity", d["parity"]["equal"], d["parity"]["bursts_checked"], "redos", d["stats"]["serial_redos"])
for r in list(csv.DictReader(open(f"{out}/r02_config{c}_kernel_stats.csv")))[:8]:
    print("   %-40s calls %4s avg %9.1f us %6s%%" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
