#!/bin/bash
# profiles/${R}_<leg>_kernel_stats.csv + ${R}_<leg>_pmc.json: rocprofv3 kernel stats and counters (HBM traffic, vector / scalar
# instructions, LDS conflicts; separate --pmc passes) of bench.py's `configs` legs, each run alone through scripts/dev/legs.py:
#   R=r04 scripts/make_profiles_configs.sh [c3 c4 busy15 busy30]      (copy gpurun_out/profiles/* to profiles/ afterwards)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=${R:-r05}
OUT=gpurun_out/profiles
mkdir -p $OUT
here=$(pwd)
legs=${@:-c3 c4 busy15 busy30}
for leg in $legs; do
  case $leg in c3) name=config3;; c4) name=config4_share;; busy15) name=busy15;; busy30) name=busy30;; *) name=$leg;; esac
  rm -rf /tmp/prl_*
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prl_s -- python $here/scripts/dev/legs.py $leg > /tmp/prl_s.log 2>&1 )
  cp $(find /tmp/prl_s -name "*kernel_stats.csv" | head -1) $OUT/${R}_${name}_kernel_stats.csv
  tail -1 /tmp/prl_s.log | cut -c1-400
  ( cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prl_f -- python $here/scripts/dev/legs.py $leg > /tmp/prl_f.log 2>&1 )
  ( cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prl_w -- python $here/scripts/dev/legs.py $leg > /tmp/prl_w.log 2>&1 )
  ( cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/prl_q -- python $here/scripts/dev/legs.py $leg > /tmp/prl_q.log 2>&1 )
  python - "$OUT" "$R" "$name" <<'PY'
import sys, glob, csv, collections, json
out, R, name = sys.argv[1:4]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("/tmp/prl_f", "/tmp/prl_w", "/tmp/prl_q"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith(("k", "void k")):
                vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in sorted(vals.items()):
    res[k] = {}
    for c, v in sorted(cs.items()):
        top = [x for x in v if x >= 0.8 * max(v)] if max(v) > 0 else v    # full-size launches only (the first push runs in parts)
        res[k][c] = sum(top) / len(top)
    if "FETCH_SIZE" in res[k]:
        res[k]["hbm_bytes"] = (2.0 * res[k]["FETCH_SIZE"] + res[k].get("WRITE_SIZE", 0.0)) * 1024.0    # gfx950: FETCH_SIZE counts half of wide reads
    if "SQ_INSTS_VALU" in res[k] and res[k].get("GRBM_GUI_ACTIVE"):
        res[k]["valu_occupancy"] = res[k]["SQ_INSTS_VALU"] * 4 / 1024 / (res[k]["GRBM_GUI_ACTIVE"] / 8)
json.dump({"per_launch": res, "_note": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ set) over scripts/dev/legs.py " + name +
           "; per kernel: mean over its full-size launches; hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes); valu_occupancy = SQ_INSTS_VALU x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)"},
          open(f"{out}/{R}_{name}_pmc.json", "w"), indent=1)
for r in list(csv.DictReader(open(f"{out}/{R}_{name}_kernel_stats.csv")))[:8]:
    print("   %-40s calls %4s avg %9.1f us %6s%%" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
