#!/bin/bash
# dev aid: scripts/busy_channels.py under rocprofv3 --kernel-trace --stats for a burst rate
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=$PWD
for bps in "$@"; do
  rm -rf /tmp/prof_busy
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_busy -- python "$R/scripts/busy_channels.py" $bps > /tmp/prof_busy.log 2>&1 )
  echo "=== $bps bursts/s/channel"; grep "^push" /tmp/prof_busy.log
  python - <<'PY'
import glob, csv
f = glob.glob("/tmp/prof_busy/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-42s calls %4s avg %10.1f us total %8.2f ms %5s%%" % (r["Name"][:42], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
done
