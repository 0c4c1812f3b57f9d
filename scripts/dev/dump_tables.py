"""dev aid: dump the candidate tables and cluster heads of one push (VDL2GPU_NO_REACH=1: every primary has its cluster) to gpurun_out/tables_<bps>.npz"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, torch
from vdlm2dec_amd import synth
from vdlm2dec_amd.demod import Receiver, plan_channels
bps = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
rate, fos = 2_000_000, synth.DEFAULT_FO_8CH
raw = np.concatenate([bench.make_tile(1000 + i, "cs16", rate, fos, bps)[1] for i in range(4)])
ns = len(raw) // 2
dev = torch.from_numpy(raw).to("cuda:0")
out = {}
with Receiver(rate, plan_channels(136_975_000, fos), fmt="cs16", max_push=ns) as rx:
    rx.push_device(dev.data_ptr(), ns)
    rx.poll()
    for ch in range(8):
        out["c%d" % ch] = rx.debug_cands(0, ch)
        out["h%d" % ch] = rx.debug_clheads(0, ch)
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/tables_%g.npz" % bps, **out)
print("saved", sum(len(out["c%d" % c]) for c in range(8)), "candidates")
