"""dev aid: how vdl2gpu_decode_blocks scales with the batch size"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vdlm2dec_amd import synth
from vdlm2dec_amd.demod import Receiver, plan_channels
rng = np.random.default_rng(5)
blocks = []
for n in rng.integers(1, 900, 4096):
    info = bytes(rng.integers(0, 256, int(n), dtype=np.uint8).tolist())
    blocks.append(synth.received_rows(synth.hdlc_payload(synth.avlc_frame(info))))
with Receiver(2_000_000, plan_channels(136975000, [-50000]), fmt="cu8", max_push=4096) as rx:
    rx.decode_blocks(blocks[:8])
    for n in (1, 16, 64, 256, 1024, 4096):
        t0 = time.perf_counter()
        fr = rx.decode_blocks(blocks[:n])
        dt = time.perf_counter() - t0
        print(n, "blocks", len(fr), "frames", round(dt * 1e3, 3), "ms (incl. copies and ctypes)")
