#!/bin/bash
# dev aid: SQ counters per kernel for one bench run.  usage: scripts/pmc_kernel.sh "<counters>" [more counter sets...]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$i -- python bench.py --steps 2 --warmup 1 --no-cpu --no-parity $BENCH_ARGS > /tmp/pmc_$i.log 2>&1
  python - "$i" <<'PY'
import sys, glob, csv, collections
i = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f"/tmp/pmc_{i}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:28]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k in sorted(agg):
    if k.startswith(("k2", "k1_", "void k1_", "k3", "k4", "k_")):
        print(k, {c: round(v / max(1, cnt[(k, c)])) for c, v in agg[k].items()})
PY
done
