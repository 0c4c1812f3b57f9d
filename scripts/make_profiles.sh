#!/bin/bash
# Regenerates profiles/r01_* on a GPU box (gpurun):  scripts/make_profiles.sh
#   r01_bench_line.json         the bench line of the default command
#   r01_bench_kernel_stats.csv  rocprofv3 --kernel-trace --stats summary of the same command
#   r01_bench_pmc_hbm.json      FETCH_SIZE / WRITE_SIZE per kernel launch (separate --pmc passes)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/profiles
mkdir -p $OUT
rm -rf /tmp/pr_stats /tmp/pr_f /tmp/pr_w
here=$(pwd)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_stats -- python $here/bench.py --no-cpu > /tmp/pr_stats.log 2>&1 )
cp $(find /tmp/pr_stats -name "*kernel_stats.csv" | head -1) $OUT/r01_bench_kernel_stats.csv
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pr_f -- python $here/bench.py --no-cpu --no-parity --steps 4 --warmup 2 > /tmp/pr_f.log 2>&1 )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pr_w -- python $here/bench.py --no-cpu --no-parity --steps 4 --warmup 2 > /tmp/pr_w.log 2>&1 )
python - "$OUT" <<'PY'
import sys, glob, csv, collections, json
out = sys.argv[1]
res = {}
for tag, d in (("FETCH_SIZE", "/tmp/pr_f"), ("WRITE_SIZE", "/tmp/pr_w")):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != tag:
                continue
            k = r["Kernel_Name"].split("(")[0]
            agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    res[tag + "_KB_per_launch"] = {k: agg[k] / cnt[k] for k in sorted(agg) if k.startswith(("k", "void k"))}
res["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `python bench.py --no-cpu --no-parity "
                "--steps 4 --warmup 2`; KB per kernel launch, averaged over launches (one k1_fast launch per push; the synchronised "
                "pushes and the ingest-ring pushes bench.py adds after its timed region are included)")
json.dump(res, open(out + "/r01_bench_pmc_hbm.json", "w"), indent=1)
PY
python bench.py 2>/dev/null | tail -1 > $OUT/r01_bench_line.json
python - "$OUT" <<'PY'
import sys, json, csv
out = sys.argv[1]
d = json.load(open(out + "/r01_bench_line.json"))
print("bench:", round(d["value"]), "MS/s", round(d["ms_per_step"], 4), "ms/step; roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "alone", d["roofline"]["alone"])
for r in list(csv.DictReader(open(out + "/r01_bench_kernel_stats.csv")))[:14]:
    print("%-44s calls %4s avg %9.1f us %6s%%" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
pm = json.load(open(out + "/r01_bench_pmc_hbm.json"))
for k in pm["FETCH_SIZE_KB_per_launch"]:
    if "k1_fast" in k:
        print(k, "FETCH KB", pm["FETCH_SIZE_KB_per_launch"][k], "WRITE KB", pm["WRITE_SIZE_KB_per_launch"].get(k))
PY
