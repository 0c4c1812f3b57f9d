"""dev aid: ONE scenario of scripts/soak.py by seed, with what differs from the oracle printed (channel, trigger instant, which side has it):
   python scripts/dev/soak_one.py <seed> [VAR=value ...]    (VAR=value: environment overrides applied after the scenario's own)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from vdlm2dec_amd import synth
from vdlm2dec_amd.demod import Receiver, plan_channels
from oracle import oracle as O

FC = 136_975_000
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
rate = int(rng.choice([2_000_000, 2_000_000, 2_000_000, 5_000_000, 10_000_000]))
nch = int(rng.integers(1, 9))
fos = [int(f * rate / 2_000_000) // 25000 * 25000 for f in synth.DEFAULT_FO_8CH[:nch]]
fmt = str(rng.choice(["cs16", "cu8", "cs16"]))
ns = int(rng.integers(3, 14)) * 1_000_000 * (rate // 2_000_000 if rate > 2_000_000 else 1) // (2 if rate > 2_000_000 else 1)
dens = float(rng.choice([3.0, 8.0, 20.0, 40.0])) * rate / 2_000_000
spec = synth.random_scenario(rate, fos, ns, seed=seed, bursts_per_s=dens, info_max=int(rng.choice([60, 240, 900])))
raw = synth.synth_stream(spec, fmt)
ob = O.run_oracle(raw, fmt, rate, fos, FC)
block = int(rng.choice([ns, ns // 2 + 17, 1_234_567, 400_000, 2_000_000, 65536]))
frames_too = bool(seed & 1)
for k in ("VDL2GPU_REPAIR_ROUNDS", "VDL2GPU_SPLIT_SAMPLES", "VDL2GPU_PRIM_DROP"):
    os.environ.pop(k, None)
mode = int(rng.integers(0, 6))
if mode == 1:
    os.environ["VDL2GPU_REPAIR_ROUNDS"] = str(int(rng.integers(1, 4)))
elif mode == 2:
    os.environ["VDL2GPU_SPLIT_SAMPLES"] = str(int(rng.choice([262144, 524288, 1 << 20])))
elif mode == 3:
    os.environ["VDL2GPU_PRIM_DROP"] = str(int(rng.integers(2, 6)))
flags = 0
from vdlm2dec_amd import lib as _lib
if mode in (1, 4):
    flags = _lib.F_TEST_NOREGION
for a in sys.argv[2:]:
    k, v = a.split("=", 1)
    if k == "NOREGION":
        flags = _lib.F_TEST_NOREGION if int(v) else 0
    elif k == "PIPELINED":
        seed = (seed & ~2) | (2 if int(v) else 0)
    else:
        os.environ[k] = v
print("seed", sys.argv[1], "rate", rate, "ch", nch, fmt, "ns", ns, "dens", dens, "block", block, "mode", mode, "flags", flags,
      {k: os.environ.get(k) for k in ("VDL2GPU_REPAIR_ROUNDS", "VDL2GPU_SPLIT_SAMPLES", "VDL2GPU_PRIM_DROP")}, "pipelined" if seed & 2 else "")
with Receiver(rate, plan_channels(FC, fos), fmt=fmt, max_push=max(block, 1 << 16), frames=frames_too, flags=flags, testhooks=True) as rx:
    if seed & 2:
        per = O.PER_SAMPLE[fmt]
        bl, nsm = [], raw.size // per
        for s0 in range(0, nsm, block):
            rx.push(raw[s0 * per:min(nsm, s0 + block) * per])
            bl += rx.poll_ready()
        bl += rx.poll()
    elif os.environ.get("PERPUSH"):
        per = O.PER_SAMPLE[fmt]
        bl, nsm, prev = [], raw.size // per, None
        for i, s0 in enumerate(range(0, nsm, block)):
            rx.push(raw[s0 * per:min(nsm, s0 + block) * per])
            got1 = rx.poll()
            bl += got1
            stn = rx.stats()
            d = {k: stn[k] - (prev[k] if prev else 0) for k in ("repairs", "serial_redos", "triggers", "bursts", "header_rejects", "deferrals", "candidates", "serial_samples")}
            prev = stn
            lo, hi = int(os.environ.get("PERPUSH_LO", "0")), int(os.environ.get("PERPUSH_HI", "1000000"))
            if lo <= i <= hi:
                print("   push", i, d, [(b.chn, b.trig_dec) for b in got1])
            if os.environ.get("DBG_PUSH") and i == int(os.environ["DBG_PUSH"]):
                ch = int(os.environ.get("DBG_CHN", "0"))
                cd = rx.debug_cands(0, ch)
                hd = rx.debug_clheads(0, ch)
                base = 84000 * (s0 // 1) // rate * 0      # (printed relative: nrel is relative to the push's dec_base)
                print("   cands of chn", ch, "after push", i, ":", len(cd))
                for k in range(len(cd)):
                    print("      #%d nrel %d r %d  p2err/perr/err bits %08x %08x %08x  head x %d y %08x (status %d r_s %d ns %d)" % (
                        k, cd[k][0], cd[k][1], int(cd[k][2]) & 0xffffffff, int(cd[k][3]) & 0xffffffff, int(cd[k][4]) & 0xffffffff,
                        hd[k][0] if k < len(hd) else -1, (int(hd[k][1]) & 0xffffffff) if k < len(hd) else 0,
                        (int(hd[k][1]) & 3) if k < len(hd) else -1, ((int(hd[k][1]) >> 2) & 3) if k < len(hd) else -1, ((int(hd[k][1]) >> 4) & 15) if k < len(hd) else -1))
    else:
        bl = list(rx.run(raw, block=block))
    st = rx.stats()
g = {(b.chn, b.trig_dec): (b.nbrow, b.nlbyte, b.data) for b in bl}
e = {(b.chn, b.trig_dec): (b.nbrow, b.nlbyte, b.data) for b in ob}
print("gpu", len(bl), "(distinct", len(g), ") oracle", len(ob), "stats", {k: st[k] for k in ("repairs", "serial_redos", "triggers", "bursts", "header_rejects")})
tile = 84000 * block // rate      # 84 kS/s frames per push
for k in sorted(set(g) - set(e)):
    print("  GPU only   chn %d trig %d (push %d + %d) nbrow %d nlbyte %d" % (k[0], k[1], k[1] // max(1, tile), k[1] % max(1, tile), g[k][0], g[k][1]))
for k in sorted(set(e) - set(g)):
    print("  oracle only chn %d trig %d (push %d + %d) nbrow %d nlbyte %d" % (k[0], k[1], k[1] // max(1, tile), k[1] % max(1, tile), e[k][0], e[k][1]))
for k in sorted(set(e) & set(g)):
    if e[k] != g[k]:
        print("  differ     chn %d trig %d" % k)
dup = len(bl) - len(g)
if dup:
    import collections
    cnt = collections.Counter((b.chn, b.trig_dec) for b in bl)
    print("  duplicates:", [k for k, v in cnt.items() if v > 1][:10])
print("OK" if sorted((b.chn, b.nbrow, b.nlbyte, b.data) for b in bl) == sorted(b.key() for b in ob) else "MISMATCH")
