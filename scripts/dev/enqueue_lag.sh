#!/bin/bash
# dev aid: for every kernel of two steady-state steps, when the host enqueued it (HIP API trace) against when it started on the GPU
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
rm -rf /tmp/prof_el
( cd /tmp && rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d /tmp/prof_el -- python "$OLDPWD/bench.py" --no-cpu --no-parity --no-extra --no-ring --steps 10 --warmup 2 "$@" > /tmp/prof_el.log 2>&1 )
python - <<'PY'
import glob, csv
kf = glob.glob("/tmp/prof_el/**/*kernel_trace.csv", recursive=True)[0]
af = glob.glob("/tmp/prof_el/**/*hip_api_trace.csv", recursive=True)
print("api files", af)
ks = list(csv.DictReader(open(kf)))
api = list(csv.DictReader(open(af[0]))) if af else []
print(api[0].keys() if api else None)
by_corr = {r["Correlation_Id"]: r for r in api if "Launch" in r.get("Function", "")}
ks.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(ks) if r["Kernel_Name"].startswith("k2a_probe")]
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
full = 0.35 * max(dur(ks[i]) for i in idx)
idx = [i for i in idx if dur(ks[i]) >= full]
a = idx[len(idx) // 2]; b = idx[min(len(idx) - 1, len(idx) // 2 + 2)]
t0 = int(ks[a]["Start_Timestamp"])
for r in ks[a - 3:b + 1]:
    c = by_corr.get(r["Correlation_Id"])
    enq = (int(c["Start_Timestamp"]) - t0) / 1e3 if c else float("nan")
    print("%9.1f us start  enq %9.1f us  lag %8.1f  +%7.1f  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, enq, (int(r["Start_Timestamp"]) - t0) / 1e3 - enq, dur(r) / 1e3, r.get("Queue_Id"), r["Kernel_Name"][:28]))
PY
python - <<'PY'
import glob, csv
af = glob.glob("/tmp/prof_el/**/*hip_api_trace.csv", recursive=True)[0]
kf = glob.glob("/tmp/prof_el/**/*kernel_trace.csv", recursive=True)[0]
api = list(csv.DictReader(open(af)))
ks = list(csv.DictReader(open(kf)))
ks.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(ks) if r["Kernel_Name"].startswith("k2a_probe")]
t0 = int(ks[idx[len(idx) // 2]]["Start_Timestamp"])
print("host API calls longer than 20 us around the same steps (start relative to the same probe, duration):")
for r in api:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if -700e3 < s - t0 < 1500e3 and e - s > 20e3:
        print("%9.1f us  %8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Function"]))
PY
