#!/bin/bash
# Regenerates profiles/rNN_* on a GPU box (gpurun):  R=r04 scripts/make_profiles.sh   (copy gpurun_out/profiles/* to profiles/ afterwards)
#   ${R}_bench_line.json         the bench line of the default command
#   ${R}_bench_kernel_stats.csv  rocprofv3 --kernel-trace --stats summary of the same command
#   ${R}_bench_pmc_hbm.json      FETCH_SIZE / WRITE_SIZE per kernel launch (separate --pmc passes)
#   ${R}_bench_pmc_sq.json       SQ counters per kernel launch (three --pmc passes of <= 8 counters)
#   ${R}_bench_line_full.json    the driver's command (--steps 20 --warmup 5) with the `configs` object and the CPU baseline
#   (round 2's micro-benchmarks and k1_fast ablations -- profiles/r02_valu_rate.txt, r02_clock_rate.txt, r02_k1_ablation.txt --
#    describe kernels this round did not change; WITH_BER=1 adds the Es/N0 sweep)
cd "$(dirname "$0")/.."
R=${R:-r05}
export TMPDIR=/tmp
OUT=gpurun_out/profiles
mkdir -p $OUT
rm -rf /tmp/pr_stats /tmp/pr_f /tmp/pr_w /tmp/pr_s1 /tmp/pr_s2 /tmp/pr_s3
here=$(pwd)
B="python $here/bench.py --no-cpu --no-ring --no-extra"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_stats -- $B > /tmp/pr_stats.log 2>&1 )
cp $(find /tmp/pr_stats -name "*kernel_stats.csv" | head -1) $OUT/${R}_bench_kernel_stats.csv
S="--no-parity --steps 4 --warmup 2"
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pr_f -- $B $S > /tmp/pr_f.log 2>&1 )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pr_w -- $B $S > /tmp/pr_w.log 2>&1 )
( cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pr_s1 -- $B $S > /tmp/pr_s1.log 2>&1 )
( cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pr_s2 -- $B $S > /tmp/pr_s2.log 2>&1 )
( cd /tmp && rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pr_s3 -- $B $S > /tmp/pr_s3.log 2>&1 )
python - "$OUT" "$R" <<'PY'
import sys, glob, csv, collections, json
out, R = sys.argv[1], sys.argv[2]
def collect(dirs):
    # per kernel and counter: the mean over its FULL-SIZE launches (within 20 % of the largest): the handle's first push is
    # cut into four parts (vdl2gpu.h, max_push), whose launches would drag a plain mean down
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                if not k.startswith(("k", "void k")):
                    continue
                vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, cs in sorted(vals.items()):
        res[k] = {}
        for c, v in sorted(cs.items()):
            top = [x for x in v if x >= 0.8 * max(v)] if max(v) > 0 else v
            res[k][c] = sum(top) / len(top)
    # per STEP: everything every launch of a kernel counted, over the pushes of the run (the handle's first push runs as four
    # parts: the probe's launches minus three) -- what bench.py's roofline.whole_step adds up (a kernel's second launch of a
    # push, e.g. the repair round's verify pass, is far smaller than its first: launches x the full-size mean overstates it)
    npush = max(1, max((len(v) for v in vals["k2a_probe"].values()), default=4) - 3) if "k2a_probe" in vals else 1
    per_step = {k: {c: sum(v) / npush for c, v in cs.items()} for k, cs in vals.items()}
    return res, per_step
hb, hb_step = collect(["/tmp/pr_f", "/tmp/pr_w"])
res = {"FETCH_SIZE_KB_per_launch": {k: v["FETCH_SIZE"] for k, v in hb.items() if "FETCH_SIZE" in v},
       "WRITE_SIZE_KB_per_launch": {k: v["WRITE_SIZE"] for k, v in hb.items() if "WRITE_SIZE" in v},
       "_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `python bench.py --no-cpu --no-ring --no-parity "
                "--steps 4 --warmup 2`; KB per kernel launch, averaged over the full-size launches (the first push of the handle runs as four parts; the "
                "synchronised pushes bench.py adds after its timed region are included).  Per MI355X_MICROARCH.md FETCH_SIZE reports half the bytes of wide coalesced reads on "
                "gfx950: bench.py's traffic_from_profiles = 2 x FETCH_SIZE + WRITE_SIZE"}
# launches of every kernel per step (push), from the traced run: what bench.py's roofline.whole_step multiplies the per-launch figures by
try:
    ks = {r["Name"].split("(")[0]: int(r["Calls"]) for r in csv.DictReader(open(out + "/" + R + "_bench_kernel_stats.csv"))}
    pushes = ks.get("k2a_probe", 0) or 1
    res["launches_per_step"] = {k: round(v / pushes, 2) for k, v in ks.items() if k.startswith(("k", "void k"))}
except (OSError, KeyError, ValueError):
    pass
res["per_step_bytes"] = {k: (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0 for k, v in hb_step.items()}
# which build these counters belong to: bench.py quotes them only when its own sources hash to the same key (R_HEAD: the commit, handed in by
# whoever starts the script -- the GPU box has no .git)
import os
sys.path.insert(0, os.getcwd())
import bench
res["source_key"] = bench.source_key()
res["head"] = os.environ.get("R_HEAD", "unknown")
json.dump(res, open(out + "/" + R + "_bench_pmc_hbm.json", "w"), indent=1)
sq, sq_step = collect(["/tmp/pr_s1", "/tmp/pr_s2", "/tmp/pr_s3"])
json.dump({"per_launch": sq, "per_step": {k: {"SQ_INSTS_VALU": v.get("SQ_INSTS_VALU", 0.0)} for k, v in sq_step.items()}, "_note": "SQ_* in quad-cycles / instructions summed over the chip, GRBM_GUI_ACTIVE summed over the 8 XCDs; "
           "three --pmc passes of the command above"}, open(out + "/" + R + "_bench_pmc_sq.json", "w"), indent=1)
PY
$B 2>/dev/null | tail -1 > $OUT/${R}_bench_line.json
# the bench line WITH the `configs` object and the CPU baseline (what the driver runs)
python $here/bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${R}_bench_line_full.json
if [ -n "$WITH_BER" ]; then timeout 900 python scripts/ber_curve.py --out $OUT/${R}_ber_curve.json > $OUT/${R}_ber_curve.txt 2>&1; fi
python - "$OUT" "$R" <<'PY'
import sys, json, csv
out, R = sys.argv[1], sys.argv[2]
d = json.load(open(out + "/" + R + "_bench_line.json"))
print("bench:", d["value"], "MS/s", round(d["ms_per_step"], 4), "ms/step; roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "alone", d["roofline"]["alone"]["avg_launch_ms"], "parity", d["parity"]["equal"], d["parity"]["bursts_checked"])
for r in list(csv.DictReader(open(out + "/" + R + "_bench_kernel_stats.csv")))[:16]:
    print("%-44s calls %4s avg %9.1f us %6s%%" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
pm = json.load(open(out + "/" + R + "_bench_pmc_hbm.json"))
for k in pm["FETCH_SIZE_KB_per_launch"]:
    print("%-40s FETCH KB %12.0f  WRITE KB %12.0f" % (k[:40], pm["FETCH_SIZE_KB_per_launch"][k], pm["WRITE_SIZE_KB_per_launch"].get(k, 0)))
PY
