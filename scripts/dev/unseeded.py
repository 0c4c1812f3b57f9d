"""Design experiment (CPU, numpy, float64 approximation of the detector): which of the detector's triggers have no probe seed
(class 0 free-running `perr < 7 && err > perr`) within +-40 samples -- the events only the verify pass can find?"""
import re, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from vdlm2dec_amd import synth
from oracle import oracle as O
txt = open("vdlm2dec_amd/csrc/vdl2_tables.inc").read()
def tab(name):
    body = txt[txt.index("VDL2_TABLE_BEGIN(%s," % name):]
    body = body[:body.index("VDL2_TABLE_END")]
    return np.array([int(x, 16) for x in re.findall(r"VDL2_F32\(0x([0-9a-f]+)u\)", body)], np.uint32).view(np.float32)
mflt = np.concatenate([tab("mflt"), np.zeros(8, np.float32)]).astype(np.float64)
sw = tab("sw").astype(np.float64)
l8 = (np.arange(17) - 8.0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1077
bps = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
fos = synth.DEFAULT_FO_8CH
spec, raw = bench.make_tile(seed, "cs16", 2_000_000, fos, bps)
for ci, fo in enumerate(fos):
    ch = O.OracleChannel(spec.rate, fo, bench.FC + fo, tap_dec=True)
    ch.feed(raw, "cs16")
    x = ch.dec().astype(np.complex128)
    trig = ch.triggers(); blocks = ch.blocks(); ch.close()
    N = len(x)
    errs = {}
    for r in range(4):
        taps = mflt[r::4][:17]
        Sf = np.convolve(x, taps[::-1])[:N]
        P = np.angle(Sf)
        for par in range(2):
            n = np.arange(200 + par, N, 2)
            ph = np.stack([P[n - 8 * (16 - l)] - sw[l] for l in range(17)], 1)
            d = np.diff(ph, axis=1)
            k = np.where(d > np.pi, -1.0, np.where(d < -np.pi, 1.0, 0.0))
            pr = ph.copy(); pr[:, 1:] += np.cumsum(k, 1) * 2 * np.pi
            pr -= pr.mean(1, keepdims=True)
            fr = (pr * l8).sum(1) / 408.0
            errs[(r, par)] = (n, ((pr - l8 * fr[:, None]) ** 2).sum(1))
    n0, e0 = errs[(0, 0)]
    seeds = n0[1:][(e0[:-1] < 7.0) & (e0[1:] > e0[:-1])]
    bad = []
    for t in trig:
        nt = t["dec_index"]
        j = np.searchsorted(seeds, nt)
        dist = min(abs(int(seeds[j - 1]) - nt) if j > 0 else 1 << 30, abs(int(seeds[j]) - nt) if j < len(seeds) else 1 << 30)
        if dist > 40:
            # the class it fired in: best (lowest) error near nt over classes
            best = min((float(e[np.searchsorted(n, nt - 2)]), rp) for rp, (n, e) in errs.items() if np.searchsorted(n, nt - 2) < len(e))
            e0near = float(e0[max(0, np.searchsorted(n0, nt) - 25):np.searchsorted(n0, nt) + 25].min())
            bad.append((nt, dist, best, round(e0near, 2)))
    print("ch%d: %d triggers, %d bursts, %d seeds; triggers without a seed within 40 samples: %s" % (ci, len(trig), len(blocks), len(seeds), bad))
    if "--all" in sys.argv:
        # every free-running firing of every class with no probe seed within +-40 samples (an event the tables cannot know)
        for rp, (n, e) in sorted(errs.items()):
            f = n[1:][(e[:-1] < 4.0) & (e[1:] > e[:-1])]
            lone = [int(t) for t in f if len(seeds) == 0 or np.min(np.abs(seeds - t)) > 40]
            if lone:
                print("   class", rp, "lone firings at", lone)
