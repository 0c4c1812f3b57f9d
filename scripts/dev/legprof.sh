#!/bin/bash
# dev aid: rocprofv3 kernel stats of bench.py legs (scripts/dev/legs.py): scripts/dev/legprof.sh busy30
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/prof_leg
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_leg -- python "$R/scripts/dev/legs.py" "$@" > /tmp/prof_leg.log 2>&1 )
cat /tmp/prof_leg.log | tail -3 | cut -c1-300
python - <<'PY'
import glob, csv
f = glob.glob("/tmp/prof_leg/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-42s calls %4s avg %10.1f us total %8.2f ms %5s%%" % (r["Name"][:42], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
