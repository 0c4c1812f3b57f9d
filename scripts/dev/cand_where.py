"""Design experiment (CPU, numpy, float64 approximation of the detector): where do the free-running detector's
firings `perr < 4 && err > perr` of all 8 classes lie relative to a burst?  (17.8 candidates per burst fill the tables.)"""
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
mflt = np.concatenate([tab("mflt"), np.zeros(8, np.float32)]).astype(np.float64)
sw = tab("sw").astype(np.float64)
spec = synth.random_scenario(2_000_000, S.FO8[:1], 8_400_000, seed=1234, bursts_per_s=4.0, info_max=240)
raw = synth.synth_stream(spec, "cs16")
ch = O.OracleChannel(spec.rate, spec.fo[0], S.FC + spec.fo[0], tap_dec=True)
ch.feed(raw, "cs16")
x = ch.dec().astype(np.complex128)
blocks = ch.blocks(); trig = ch.triggers(); ch.close()
N = len(x)
l8 = (np.arange(17) - 8.0)
fires = []
for r in range(4):
    taps = mflt[r::4][:17]
    Sf = np.convolve(x, taps[::-1])[:N]
    P = np.angle(Sf)
    for par in range(2):
        n = np.arange(200 + par, N, 2)
        ph = np.stack([P[n - 8 * (16 - l)] - sw[l] for l in range(17)], 1)        # [inst, 17]
        d = np.diff(ph, axis=1)
        k = np.where(d > np.pi, -1.0, np.where(d < -np.pi, 1.0, 0.0))
        pr = ph.copy(); pr[:, 1:] += np.cumsum(k, 1) * 2 * np.pi
        pr -= pr.mean(1, keepdims=True)
        fr = (pr * l8).sum(1) / 408.0
        err = ((pr - l8 * fr[:, None]) ** 2).sum(1)
        f = np.where((err[:-1] < 4.0) & (err[1:] > err[:-1]))[0] + 1
        for i in f:
            fires.append((int(n[i]), r * 2 + par))
fires.sort()
fn = np.array([f[0] for f in fires])
print("bursts", len(blocks), "triggers", len(trig), "free-running firings of all classes", len(fires), "per burst %.1f" % (len(fires) / max(1, len(blocks))))
# position of firings relative to the nearest burst [trig, end]
rel = {"before trig-20": 0, "trig-20..trig+20": 0, "inside burst": 0, "end..end+150": 0, "elsewhere": 0}
tr = np.array([b.trig_dec for b in blocks]); en = np.array([b.end_dec for b in blocks])
for n0, c in fires:
    j = np.searchsorted(tr, n0 + 20) - 1
    if j >= 0 and abs(n0 - tr[j]) <= 20: rel["trig-20..trig+20"] += 1
    elif j >= 0 and tr[j] + 20 < n0 <= en[j]: rel["inside burst"] += 1
    elif j >= 0 and en[j] < n0 <= en[j] + 150: rel["end..end+150"] += 1
    else: rel["elsewhere"] += 1
print(rel)
