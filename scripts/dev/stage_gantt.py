"""dev aid: VDL2GPU_STAGE_DUMP=1 python bench.py ... 2> err.txt; python scripts/dev/stage_gantt.py err.txt
prints, for the pushes in the middle of the timed region, when each stage began and ended relative to the push's channeliser start,
the period between consecutive pushes, and how long each stream was busy per period."""
import re, sys
rows = {}
for ln in open(sys.argv[1]):
    m = re.match(r"vdl2gpu stage dump push (\d+):(.*)", ln)
    if m:
        rows[int(m.group(1))] = {int(k): float(v) for k, v in re.findall(r"e(\d+)=(-?[\d.]+)", m.group(2))}
ks = sorted(rows)
# steady state: the longest run of pushes whose K1 starts are < 0.8 ms apart
best, cur = (0, 0), 0
for i in range(1, len(ks) + 1):
    if i == len(ks) or rows[ks[i]][0] - rows[ks[i - 1]][0] > 800 or ks[i] != ks[i - 1] + 1:
        if i - cur > best[1] - best[0]:
            best = (cur, i)
        cur = i
sel = ks[best[0] + 3:best[1] - 3]
print("pushes", sel[0], "..", sel[-1], "period (K1 start to K1 start) us:", [round(rows[b][0] - rows[a][0]) for a, b in zip(sel, sel[1:])][:24])
names = [(0, "K1 begins"), (1, "K1 ends"), (10, "scan begins"), (4, "front ends"), (2, "clusters begin"), (13, "clusters end"), (23, "resolver begins"), (14, "resolver ends"),
         (12, "verify begins"), (15, "verify ends"), (16, "tail begins"), (17, "merge ends"), (18, "patch ends"), (5, "rounds end"), (20, "commit ends"),
         (21, "payload2 ends"), (22, "export ends"), (6, "commit..export end"), (7, "tail ends")]
for p in sel[4:10]:
    r = rows[p]
    print("push %d (K1 at %.0f):" % (p, r[0]), "  ".join("%s %+.0f" % (n, r[k] - r[0]) for k, n in names if r.get(k, -1) >= 0))
n = len(sel) - 1
span = rows[sel[-1]][0] - rows[sel[0]][0]
front = sum(rows[p][4] - rows[p][0] for p in sel[:-1])
back = sum(rows[p][15] - rows[p][2] for p in sel[:-1] if rows[p].get(15, -1) >= 0)
tail = sum(rows[p][7] - rows[p][15] for p in sel[:-1] if rows[p].get(15, -1) >= 0)
print("per period %.0f us: front stream busy %.0f, back (clusters..verify) %.0f, tail (verify end..tail end) %.0f; push latency %.0f" %
      (span / n, front / n, back / n, tail / n, sum(rows[p][7] - rows[p][0] for p in sel[:-1]) / n))
gaps = [rows[b][0] - rows[a][4] for a, b in zip(sel, sel[1:])]
print("front stream idle between a push's front end and the next K1: avg %.0f us" % (sum(gaps) / len(gaps)))
gaps2 = [rows[b][2] - rows[a][15] for a, b in zip(sel, sel[1:]) if rows[a].get(15, -1) >= 0]
print("main stream idle between a push's verify end and the next clusters: avg %.0f us" % (sum(gaps2) / len(gaps2)))
gaps3 = [rows[b][15] - rows[a][7] for a, b in zip(sel, sel[1:]) if rows[b].get(15, -1) >= 0]
print("tail stream idle between a push's tail end and the next verify end: avg %.0f us" % (sum(gaps3) / len(gaps3)))
