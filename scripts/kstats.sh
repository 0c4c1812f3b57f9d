#!/bin/bash
# dev aid: per-kernel average durations of one bench run (rocprofv3 --kernel-trace --stats)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf /tmp/prof_ks
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -- python "$OLDPWD/bench.py" --no-cpu --no-parity --steps 6 --warmup 2 "$@" > /tmp/prof_ks.log 2>&1 )
python - <<'PY'
import glob, csv
f = glob.glob("/tmp/prof_ks/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-42s calls %4s avg %10.1f us  %5s%%" % (r["Name"][:42], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
