"""Dev aid: statistics of the sync-scan candidates (how they cluster in time / by hypothesis)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vdlm2dec_amd import synth
from vdlm2dec_amd.demod import Receiver, plan_channels
sys.path.insert(0, "tests"); 
RATE=2000000; FC=136975000; TILE=4200000
spec = synth.random_scenario(RATE, synth.DEFAULT_FO_8CH, TILE, seed=1234, bursts_per_s=4.0, info_max=240)
raw = synth.synth_stream(spec, "cs16")
rx = Receiver(RATE, plan_channels(FC, synth.DEFAULT_FO_8CH), fmt="cs16", max_push=TILE)
rx.push(raw); bursts = rx.poll()
print("bursts", len(bursts), rx.stats())
for ch in range(8):
    cd = rx.debug_cands(0, ch)
    n = cd[:,0]; r = cd[:,1]
    cls = r*2 + (n & 1)
    order = np.argsort(n); n=n[order]; cls=cls[order]
    # regions by class-0..7 individually: gaps > 200 samples
    gaps = np.diff(n)
    regions = 1 + int((gaps > 300).sum())
    # for each class: number of region starts where that class has a candidate within the region
    reg_id = np.concatenate([[0], np.cumsum(gaps > 300)])
    per_class_regs = [len(set(reg_id[cls==k])) for k in range(8)]
    reg_len = [n[reg_id==i].max()-n[reg_id==i].min() for i in range(regions)]
    print("ch", ch, "cands", len(n), "regions", regions, "regions seen per class", per_class_regs, "max region span", max(reg_len), "mean span", int(np.mean(reg_len)))
