#!/bin/bash
# dev aid (round 5): environment settings alternating, FIVE rounds (the run-to-run noise on one box is +-3 %, outliers 10 %: medians)
cd "$(dirname "$0")/../.."
tag=${1:-r05}; shift
mkdir -p gpurun_out
for rep in 1 2 3 4 5; do
for k in "$@"; do
  env $k python bench.py --no-cpu --no-extra --no-ring --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s' % '$k', round(d['ms_per_step'],4), round(d['steady_state']['ms_per_step'],4))
"
done; done > gpurun_out/${tag}_x5.txt 2>&1
python - gpurun_out/${tag}_x5.txt <<'PY'
import sys, collections, statistics
v = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    p = ln.split()
    if len(p) >= 3:
        v[" ".join(p[:-2])].append((float(p[-2]), float(p[-1])))
for k, xs in v.items():
    print("%-40s step median %.4f (%s)  steady median %.4f" % (k, statistics.median(a for a, _ in xs), " ".join("%.3f" % a for a, _ in xs), statistics.median(b for _, b in xs)))
PY
