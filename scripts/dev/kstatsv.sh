#!/bin/bash
# dev aid: per-kernel averages (rocprofv3 --kernel-trace --stats) of a bench run for each variants/*.so
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=$PWD
for v in variants/*.so; do
  rm -rf /tmp/prof_ks
  ( cd /tmp && VDL2GPU_LIB=$R/$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -- python "$R/bench.py" --no-cpu --no-parity --no-ring --no-extra --steps 16 --warmup 4 "$@" > /tmp/prof_ks.log 2>&1 )
  echo "== $v"
  python - <<'PY'
import glob, csv
f = glob.glob("/tmp/prof_ks/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-42s calls %4s avg %10.1f us  %5s%%" % (r["Name"][:42], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
