"""dev aid (round 5): for every burst of a few bench tiles, the class each of its (up to 8) trigger candidates enters the detector in
and the class / instant the idle search resumes in behind it (cluster heads).  How many DIFFERENT exits does a burst have?  How
many clusters would a resolver need if only the exits reachable from the previous burst's exits were made?"""
import sys, os, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from vdlm2dec_amd import synth
from vdlm2dec_amd.demod import Receiver, plan_channels

bps = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
HEDGE = len(sys.argv) > 2 and sys.argv[2] == 'hedge'
rate, fos = 2_000_000, synth.DEFAULT_FO_8CH
tiles = [bench.make_tile(1000 + i, "cs16", rate, fos, bps)[1] for i in range(4)]
raw = np.concatenate(tiles)
ns = len(raw) // 2
import torch
dev = torch.from_numpy(raw).to("cuda:0")
hist = collections.Counter(); nexit = collections.Counter(); npred = collections.Counter()
tot_b = tot_c = need = 0
pred_ok = pred_bad = 0
rounds = collections.Counter(); misses = collections.Counter(); chains = collections.Counter()
def pred_exit(nrel_i, p2, pe, er):
    of = np.float32(4.0) * (p2 - np.float32(4.0) * pe + np.float32(3.0) * er) / (p2 - np.float32(2.0) * pe + er)
    clk0 = int(np.floor(of + np.float32(0.5))) if np.isfinite(of) else 0
    clk0 = min(max(clk0, 0), 68)
    j = (32 - clk0 + 3) // 4
    j = max(j, 1)
    rb = clk0 + 4 * j - 32
    return rb * 2 + ((int(nrel_i) + j) & 1)
with Receiver(rate, plan_channels(136_975_000, fos), fmt="cs16", max_push=ns) as rx:
    rx.push_device(dev.data_ptr(), ns)
    rx.poll()
    for ch in range(8):
        c = rx.debug_cands(0, ch)
        hd = rx.debug_clheads(0, ch)
        nrel, r = c[:, 0], c[:, 1]
        fl = c[:, 2:6].copy().view(np.float32)
        stat = hd[:, 1] & 3
        made_total = globals().get("made_total", 0) + int((stat != 3).sum()); globals()["made_total"] = made_total
        cands_total = globals().get("cands_total", 0) + len(stat); globals()["cands_total"] = cands_total
        r_s = (hd[:, 1] >> 2) & 3
        n_s = hd[:, 0]
        order = np.argsort(nrel, kind="stable")
        groups, cur = [], [order[0]]
        for i in order[1:]:
            if nrel[i] - nrel[cur[0]] <= 40:
                cur.append(i)
            else:
                groups.append(cur); cur = [i]
        groups.append(cur)
        # ---- per burst: entering class -> (candidate, actual exit class, predicted exit class)
        bursts = []
        for g in groups:
            g = [i for i in g if stat[i] == 0]        # steady clusters only
            if not g:
                continue
            ent = {}
            for i in g:
                ent.setdefault(int(r[i]) * 2 + int(nrel[i] & 1), i)
            ex = {e: int(r_s[i]) * 2 + int(n_s[i] & 1) for e, i in ent.items()}
            px = {e: pred_exit(nrel[i], fl[i, 0], fl[i, 1], fl[i, 2]) for e, i in ent.items()}
            bursts.append((ent, ex, px))
            tot_b += 1; tot_c += len(ent); hist[len(ent)] += 1
            for e in ent:
                if px[e] == ex[e]: pred_ok += 1
                else: pred_bad += 1
        def propagate(have):
            """have[k] = classes of burst k whose cluster exists (their ACTUAL exit is known); returns needed[k] under propagation with
            actual exits where known, predicted ones otherwise.  HEDGE: a burst's needed classes also include the exits of EVERY
            candidate of the burst before (the chain may have met that one in a class it was not expected in)"""
            reach, needed = set(range(8)), []
            hedge = set()
            for k, (ent, ex, px) in enumerate(bursts):
                use = {e for e in ent if e in reach or (HEDGE and e in hedge)}
                needed.append(use)
                reach = {(ex[e] if e in have[k] else px[e]) for e in use if e in reach} | {c_ for c_ in reach if c_ not in ent}
                hedge = {(ex[e] if e in have[k] else px[e]) for e in ent} | {c_ for c_ in hedge if c_ not in ent}
            return needed
        have = [set() for _ in bursts]
        for rnd in range(3):
            needed = propagate(have)
            new = sum(len(n - h) for n, h in zip(needed, have))
            rounds[rnd] += new
            have = [h | n for h, n in zip(have, needed)]
            # the real chain through the tables as they stand after this round: replays where it meets a burst in a class without a cluster
            for start in range(8):
                cls = start
                for k, (ent, ex, px) in enumerate(bursts):
                    if cls in ent:
                        if cls not in have[k]:
                            misses[rnd] += 1
                        cls = ex[cls]
                chains[rnd] += len(bursts)
print("library made", globals().get("made_total"), "clusters for", globals().get("cands_total"), "candidates")
print("bursts", tot_b, "clusters made", tot_c, "(%.2f per burst)" % (tot_c / tot_b))
print("entering classes per burst:", sorted(hist.items()))
print("exit class predicted from the candidate record: right %d, wrong %d (%.1f %%)" % (pred_ok, pred_bad, 100.0 * pred_bad / (pred_ok + pred_bad)))
for rnd in range(3):
    print("round %d: %d new clusters (%.2f per burst, %.2f so far); the real chain (8 start classes) meets %.2f %% of its bursts without a cluster" %
          (rnd, rounds[rnd], rounds[rnd] / tot_b, sum(rounds[q] for q in range(rnd + 1)) / tot_b, 100.0 * misses[rnd] / max(1, chains[rnd])))

# ---- the library's own algorithm (k2s_sort / k2s_fix: groups by gaps of 16 samples, primaries = first of its class within 72 samples), emulated on
#      the complete tables (run with VDL2GPU_NO_REACH=1 so that every primary has a head)
if os.environ.get("VDL2GPU_NO_REACH"):
  for GAP in (16, 24, 40, 64):
    totA = totB = 0
    with Receiver(rate, plan_channels(136_975_000, fos), fmt="cs16", max_push=ns) as rx:
        rx.push_device(dev.data_ptr(), ns)
        rx.poll()
        for ch in range(8):
            c = rx.debug_cands(0, ch); hd = rx.debug_clheads(0, ch)
            fl = c[:, 2:6].copy().view(np.float32)
            key = c[:, 0].astype(np.int64) * 4 + c[:, 1]
            order = np.argsort(key, kind="stable")
            n = c[order, 0]; r = c[order, 1]; cls = r * 2 + (n & 1)
            stat = hd[order, 1] & 3; ex = ((hd[order, 1] >> 2) & 3) * 2 + (hd[order, 0] & 1)
            px = np.array([pred_exit(c[i, 0], fl[i, 0], fl[i, 1], fl[i, 2]) for i in order])
            prim = np.ones(len(n), bool)
            for j in range(len(n)):
                i = j - 1
                while i >= 0 and n[j] - n[i] < 72:
                    if cls[i] == cls[j]:
                        prim[j] = False; break
                    i -= 1
            first = np.ones(len(n), bool); first[1:] = (n[1:] - n[:-1]) >= GAP
            gid = np.cumsum(first) - 1
            def run(have):
                reach, hedge, want = 0xff, 0, np.zeros(len(n), bool)
                for g in range(gid[-1] + 1):
                    m = np.nonzero((gid == g) & prim)[0]
                    need = reach | hedge
                    nr, allex, present = 0, 0, 0
                    for j in m:
                        em = (1 << ex[j]) if (have[j] and stat[j] == 0) else (0xff if have[j] else (1 << px[j]))
                        if (need >> cls[j]) & 1: want[j] = True
                        if (reach >> cls[j]) & 1: nr |= em
                        allex |= em; present |= 1 << cls[j]
                    nr |= reach & ~present
                    reach = nr if nr else 0xff
                    hedge = allex | (hedge & ~present)
                return want
            have = np.zeros(len(n), bool)
            wA = run(have); have |= wA; totA += int(wA.sum())
            wB = run(have); totB += int((wB & ~have).sum())
    print("emulation of the library's algorithm, gap %d: first launch %d clusters, second %d" % (GAP, totA, totB))
