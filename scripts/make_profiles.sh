#!/bin/bash
# Regenerates profiles/r02_* on a GPU box (gpurun):  scripts/make_profiles.sh
#   r02_bench_line.json         the bench line of the default command
#   r02_bench_kernel_stats.csv  rocprofv3 --kernel-trace --stats summary of the same command
#   r02_bench_pmc_hbm.json      FETCH_SIZE / WRITE_SIZE per kernel launch (separate --pmc passes)
#   r02_bench_pmc_sq.json       SQ counters per kernel launch (three --pmc passes of <= 8 counters)
#   r02_valu_rate.txt           scripts/micro/valu_rate.bin: what packed / plain FP32 and the mixer's instruction mix issue at
#   r02_k1_ablation.txt         k1_fast rebuilt without mixer / barrier / priority rotation (scripts/r02_probe6.sh) and with phase stamps (r02_probe7.sh)
#   r02_clock_rate.txt          scripts/micro/clock_rate.bin: shader clock and packed-FP32 issue interval against wavefronts per SIMD
#   r02_ber_curve.json          scripts/ber_curve.py: frame success vs Es/N0, GPU == oracle at every point
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/profiles
mkdir -p $OUT
rm -rf /tmp/pr_stats /tmp/pr_f /tmp/pr_w /tmp/pr_s1 /tmp/pr_s2 /tmp/pr_s3
here=$(pwd)
B="python $here/bench.py --no-cpu --no-ring"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_stats -- $B > /tmp/pr_stats.log 2>&1 )
cp $(find /tmp/pr_stats -name "*kernel_stats.csv" | head -1) $OUT/r02_bench_kernel_stats.csv
S="--no-parity --steps 4 --warmup 2"
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pr_f -- $B $S > /tmp/pr_f.log 2>&1 )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pr_w -- $B $S > /tmp/pr_w.log 2>&1 )
( cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pr_s1 -- $B $S > /tmp/pr_s1.log 2>&1 )
( cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pr_s2 -- $B $S > /tmp/pr_s2.log 2>&1 )
( cd /tmp && rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pr_s3 -- $B $S > /tmp/pr_s3.log 2>&1 )
python - "$OUT" <<'PY'
import sys, glob, csv, collections, json
out = sys.argv[1]
def collect(dirs):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                if not k.startswith(("k", "void k")):
                    continue
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    return {k: {c: v / cnt[(k, c)] for c, v in sorted(cs.items())} for k, cs in sorted(agg.items())}
hb = collect(["/tmp/pr_f", "/tmp/pr_w"])
res = {"FETCH_SIZE_KB_per_launch": {k: v["FETCH_SIZE"] for k, v in hb.items() if "FETCH_SIZE" in v},
       "WRITE_SIZE_KB_per_launch": {k: v["WRITE_SIZE"] for k, v in hb.items() if "WRITE_SIZE" in v},
       "_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `python bench.py --no-cpu --no-ring --no-parity "
                "--steps 4 --warmup 2`; KB per kernel launch, averaged over launches (the synchronised pushes bench.py adds after its "
                "timed region are included).  Per MI355X_MICROARCH.md FETCH_SIZE reports half the bytes of wide coalesced reads on "
                "gfx950: bench.py's traffic_from_profiles = 2 x FETCH_SIZE + WRITE_SIZE"}
json.dump(res, open(out + "/r02_bench_pmc_hbm.json", "w"), indent=1)
sq = collect(["/tmp/pr_s1", "/tmp/pr_s2", "/tmp/pr_s3"])
json.dump({"per_launch": sq, "_note": "SQ_* in quad-cycles / instructions summed over the chip, GRBM_GUI_ACTIVE summed over the 8 XCDs; "
           "three --pmc passes of the command above"}, open(out + "/r02_bench_pmc_sq.json", "w"), indent=1)
PY
$B 2>/dev/null | tail -1 > $OUT/r02_bench_line.json
timeout 300 scripts/micro/valu_rate.bin > $OUT/r02_valu_rate.txt 2>&1
timeout 120 scripts/micro/clock_rate.bin > $OUT/r02_clock_rate.txt 2>&1
( printf "%s\n" "-DK1F_BASE" "-DK1F_NOMIX" "-DK1F_NOPRIO" | timeout 600 scripts/r02_probe6.sh; printf "%s\n" "-DK1F_BASE" | timeout 300 scripts/r02_probe7.sh ) > $OUT/r02_k1_ablation.txt 2>&1
timeout 900 python scripts/ber_curve.py --out $OUT/r02_ber_curve.json > $OUT/r02_ber_curve.txt 2>&1
python - "$OUT" <<'PY'
import sys, json, csv
out = sys.argv[1]
d = json.load(open(out + "/r02_bench_line.json"))
print("bench:", d["value"], "MS/s", round(d["ms_per_step"], 4), "ms/step; roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "alone", d["roofline"]["alone"]["avg_launch_ms"], "parity", d["parity"]["equal"], d["parity"]["bursts_checked"])
for r in list(csv.DictReader(open(out + "/r02_bench_kernel_stats.csv")))[:16]:
    print("%-44s calls %4s avg %9.1f us %6s%%" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
pm = json.load(open(out + "/r02_bench_pmc_hbm.json"))
for k in pm["FETCH_SIZE_KB_per_launch"]:
    print("%-40s FETCH KB %12.0f  WRITE KB %12.0f" % (k[:40], pm["FETCH_SIZE_KB_per_launch"][k], pm["WRITE_SIZE_KB_per_launch"].get(k, 0)))
PY
cat $OUT/r02_k1_ablation.txt $OUT/r02_ber_curve.txt
