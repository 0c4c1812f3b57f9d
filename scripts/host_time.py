"""dev aid: how much host time one bench step takes (push call, poll_ready call) vs the step itself"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench as B
from vdlm2dec_amd import lib as _lib
from vdlm2dec_amd.demod import Receiver, plan_channels
from vdlm2dec_amd import synth
RATE = 2_000_000
spec, tile = B.make_tile(0, "cs16", RATE, synth.DEFAULT_FO_8CH)
fos = list(spec.fo)
batch = 16 * len(tile) // 2
dev = torch.device("cuda:0")
dbatch = torch.from_numpy(np.tile(tile, 16)).to(dev)
rx = Receiver(RATE, [plan_channels(B.FC, fos)], fmt="cs16", max_push=batch, max_bursts=1 << 18)
buf = (_lib.BurstT * 16384)()
for _ in range(4):
    rx.push_device(dbatch.data_ptr(), batch, 0); rx.poll_raw(buf, 16384)
tp = tq = 0.0
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 32
for _ in range(N):
    a = time.perf_counter(); rx.push_device(dbatch.data_ptr(), batch, 0); b = time.perf_counter()
    rx.poll_ready_raw(buf, 16384); c = time.perf_counter()
    tp += b - a; tq += c - b
rx.poll_raw(buf, 16384); rx.sync()
dt = time.perf_counter() - t0
print("per step: total %.3f ms, push call %.3f ms, poll_ready call %.3f ms" % (dt / N * 1e3, tp / N * 1e3, tq / N * 1e3))
# the enqueue cost alone: the GPU idle before every push (nothing to wait for inside the call)
tp = 0.0
for _ in range(N):
    rx.sync()
    a = time.perf_counter(); rx.push_device(dbatch.data_ptr(), batch, 0); tp += time.perf_counter() - a
    rx.poll_raw(buf, 16384)
print("push call with an idle GPU (enqueue cost only): %.3f ms" % (tp / N * 1e3))
