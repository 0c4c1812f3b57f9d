#!/bin/bash
# dev aid: kernel timeline of one steady-state step (start offset, duration, gap to previous end on the same queue)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf /tmp/prof_tl
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -- python "$OLDPWD/bench.py" --no-cpu --no-parity --no-extra --steps 6 --warmup 2 "$@" > /tmp/prof_tl.log 2>&1 )
python - <<'PY'
import glob, csv
f = glob.glob("/tmp/prof_tl/**/*kernel_trace.csv", recursive=True)[0]
import shutil, os
if os.path.isdir("gpurun_out"): shutil.copy(f, "gpurun_out/timeline_trace.csv")
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the 5th k2a_probe and print until the 6th
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k2a_probe")]
full = 0.35 * max(dur(rows[i]) for i in idx)
idx = [i for i in idx if dur(rows[i]) >= full]          # full-size pushes only
# a step of the timed region: the middle of the longest run of pushes that follow each other within 0.8 ms (warm-up and
# the "alone" pushes at the end are synchronous, the very first push runs as several parts)
st = [int(rows[i]["Start_Timestamp"]) for i in idx]
best, cur = (0, 0), 0
for k in range(1, len(idx) + 1):
    if k == len(idx) or st[k] - st[k - 1] > 800_000:
        if k - cur > best[1] - best[0]:
            best = (cur, k)
        cur = k
mid = (best[0] + best[1]) // 2
a, b = idx[mid], idx[min(mid + 2, best[1] - 1)]
t0 = int(rows[a]["Start_Timestamp"])
last_end = {}
for r in rows[a - 3:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "0")
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print("%9.1f us  +%7.1f us  gap %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, r["Kernel_Name"][:40]))
PY
