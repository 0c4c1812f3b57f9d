"""Design experiment (CPU, numpy): how wide is the first screen's peak R around a real trigger?  Decides whether a
burst FINDER may look at every 2nd / 4th evaluation instant of one class only (the probe only has to find
bursts; the verify pass is what makes the result exact)."""
import re, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vdlm2dec_amd import synth
from oracle import oracle as O
sys.path.insert(0, "tests")
import scenarios as S

txt = open("vdlm2dec_amd/csrc/vdl2_tables.inc").read()
def tab(name):
    body = txt[txt.index("VDL2_TABLE_BEGIN(%s," % name):]
    body = body[:body.index("VDL2_TABLE_END")]
    return np.array([int(x, 16) for x in re.findall(r"VDL2_F32\(0x([0-9a-f]+)u\)", body)], np.uint32).view(np.float32)
mflt = np.concatenate([tab("mflt"), np.zeros(8, np.float32)])
sw = tab("sw")
steps = np.diff(sw.astype(np.float64))          # template phase steps
c = np.exp(-1j * steps)                          # 16 rotations

dens = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
noise = float(sys.argv[2]) if len(sys.argv) > 2 else None
spec = synth.random_scenario(2_000_000, S.FO8[:2], 12_000_000, seed=5, bursts_per_s=dens, info_max=120)
if noise:
    spec.noise = noise
raw = synth.synth_stream(spec, "cs16")
for ci, fo in enumerate(spec.fo):
    ch = O.OracleChannel(spec.rate, fo, S.FC + fo, tap_dec=True)
    ch.feed(raw, "cs16")
    x = ch.dec().astype(np.complex128)
    trig = ch.triggers()
    ch.close()
    res = {}
    for r in range(4):
        taps = mflt[r::4][:17]
        # S[n] = sum_k x[n-16+k] * taps[k]
        Sf = np.convolve(x, taps[::-1])[: len(x)]        # Sf[n] = sum_k x[n-k] taps_rev[k] -> x[n-16+k]*taps[k] at index n
        u = Sf / np.maximum(np.abs(Sf), 1e-30)
        st = np.zeros_like(u); st[8:] = u[8:] * np.conj(u[:-8])
        # R at eval instant n (as perr-evaluation): sum_l c_l st[n - 8*(15-l)], l=0..15 -> step l+1 of the template
        R = np.zeros(len(x))
        acc = np.zeros(len(x), complex)
        for l in range(16):
            sh = 8 * (15 - l)
            acc[sh:] += c[l] * st[: len(x) - sh] if sh else c[l] * st
        res[r] = np.abs(acc)
    widths = {2: [], 4: [], 8: []}
    miss = {2: 0, 4: 0, 8: 0}
    thr = 9.0
    nreal = 0
    for t in trig:
        n = t["dec_index"] - 2          # the minimum (perr) evaluation
        # best R over all classes near n, and what a sub-sampled probe of class (r=0, even n) would see
        best = max(res[r][n - 3:n + 4].max() for r in range(4))
        if best < 11.0:
            continue        # not a sync word (stale-ring re-trigger, noise): nothing a finder has to find
        nreal = nreal + 1
        for stp in (2, 4, 8):
            grid = np.arange((n - 12) // stp * stp, n + 13, stp)
            seen = res[0][grid].max()
            if seen <= thr:
                miss[stp] += 1
            widths[stp].append(seen)
    ntr = len(trig)
    # noise false alarm rate per instant for thresholds
    Rn = res[0][::2]
    print(f"ch{ci}: {ntr} triggers, {nreal} sync words; probe step 2/4/8 samples: missed(R<={thr}) {miss}; "
          f"min seen R: { {k: round(float(np.min(v)), 2) for k, v in widths.items()} }; "
          f"share of instants with R>{thr}: {np.mean(Rn > thr):.5f}, R>7.5: {np.mean(Rn > 7.5):.4f}")
