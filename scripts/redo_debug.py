import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,"tests")
import numpy as np, ctypes as C
from vdlm2dec_amd import synth
from vdlm2dec_amd.demod import Receiver, plan_channels
import scenarios as S
import bench
rate = 10_000_000
fos = tuple(int(f * rate / bench.RATE) // 25000 * 25000 for f in synth.DEFAULT_FO_8CH)
spec, tile = bench.make_tile(seed=1234, fmt="cs16", rate=rate, fos=fos)
raw = np.tile(tile, 8)
rx = Receiver(rate, plan_channels(S.FC, fos), fmt="cs16", max_push=raw.size // 2)
prev = rx.stats()
for step in range(3):
    rx.push(raw)
    rx.sync()
    st = rx.stats()
    fail = np.zeros(8, np.int32); rx.L.vdl2gpu_debug_fail(rx.h, fail.ctypes.data_as(C.c_void_p), 8)
    print("step", step, "redos", st["serial_redos"], "deferrals", st["deferrals"], "fail", [int(f) if f < 0x7f000000 else -1 for f in fail])
    for ch in range(8):
        if fail[ch] < 0x7f000000:
            segs = np.zeros((4096,4), np.int32); n = rx.L.vdl2gpu_debug_segs(rx.h, 0, ch, segs.ctypes.data_as(C.c_void_p), 4096)
            cd = rx.debug_cands(0, ch); cd = cd[np.argsort(cd[:,0])]
            near = cd[np.abs(cd[:,0]-fail[ch]) < 300]
            sg = segs[:n]; hit = sg[(sg[:,0] <= fail[ch]) & (sg[:,1] > fail[ch])]
            print("  ch", ch, "fail at", fail[ch], "nsegs", n, "segment containing fail:", hit.tolist())
            print("   cands near fail:", [(int(a),int(b)) for a,b in near[:,:2]])
            print("   first cands:", [(int(a),int(b)) for a,b in cd[:10,:2]], "last:", [(int(a),int(b)) for a,b in cd[-6:,:2]])
    prev = st
print(rx.stats())
