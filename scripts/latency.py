"""Config 5 shape: the live path.  A producer hands over 32768-sample cu8 blocks (one RTL-SDR
USB buffer, 16.384 ms of air time at 2 MS/s) from host memory, as fast as the library takes them and then
paced at real time; reports what one hand-off costs end to end (vdl2gpu_push of a host block + vdl2gpu_poll
until its bursts are on the host).  Usage: python scripts/latency.py [nblocks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vdlm2dec_amd import synth
from vdlm2dec_amd.demod import Receiver, plan_channels

RATE, FC, BLK = 2_000_000, 136_975_000, 32768
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 600
spec = synth.random_scenario(RATE, synth.DEFAULT_FO_8CH, nblk * BLK, seed=77, bursts_per_s=8.0, info_max=200)
raw = synth.synth_stream(spec, "cu8")
with Receiver(RATE, plan_channels(FC, spec.fo), fmt="cu8", max_push=BLK, serial=bool(os.environ.get("SERIAL"))) as rx:
    for i in range(8):                                   # warm-up
        rx.push(raw[2 * i * BLK:2 * (i + 1) * BLK]); rx.poll()
    lat, nb = [], 0
    for i in range(8, nblk):
        t0 = time.perf_counter()
        rx.push(raw[2 * i * BLK:2 * (i + 1) * BLK])
        nb += len(rx.poll())
        lat.append(time.perf_counter() - t0)
    lat = np.array(lat) * 1e3
    print("blocks %d, bursts %d; hand-off -> bursts on the host: median %.3f ms, p99 %.3f ms, max %.3f ms "
          "(one block = 16.384 ms of air time: %.0fx faster than real time block by block)"
          % (len(lat), nb, np.median(lat), np.percentile(lat, 99), lat.max(), 16.384 / np.median(lat)))
