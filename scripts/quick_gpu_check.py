"""Quick end-to-end parity check on a GPU box (dev aid; the real tests live in tests/)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vdlm2dec_amd import synth
from vdlm2dec_amd.demod import Receiver, plan_channels
from oracle import oracle as O

def scenario(rate, fo, seed, fmt):
    rng = np.random.default_rng(seed)
    bursts = []; t = 0.003
    for i, n in enumerate([1, 2, 3, 28, 31, 60, 66, 70, 200, 247, 250, 497, 900]):
        b = synth.Burst(chan=i % len(fo), t0=t, info=bytes(rng.integers(0, 256, n, dtype=np.uint8).tolist()),
                        amp=float(rng.uniform(8, 60)), cfo=float(rng.uniform(-400, 400)))
        bursts.append(b); t += b.duration() + 0.002 + rng.uniform(0, 1e-3)
    ns = int((t + 0.01) * rate); ns = (ns + 32767) // 32768 * 32768
    spec = synth.StreamSpec(rate=rate, fo=fo, nsamples=ns, bursts=bursts, noise=1.7, seed=seed)
    return synth.synth_stream(spec, fmt)

def main():
    rate = 2000000; fo = [-50000, 250000]; fc = 136975000
    for fmt in ("cu8", "cs16"):
        x = scenario(rate, fo, 3, fmt)
        ob = O.run_oracle(x, fmt, rate, fo, fc)
        for block in (None, 32768, 100000):
            rx = Receiver(rate, plan_channels(fc, fo), fmt=fmt, max_push=1 << 22, keep_dec=True)
            t0 = time.time()
            gb = rx.run(x, block=block)
            dt = time.time() - t0
            ok = sorted(b.key() for b in ob)
            gk = sorted((b.chn, b.nbrow, b.nlbyte, b.data) for b in gb)
            print(fmt, "block", block, "oracle bursts", len(ob), "gpu bursts", len(gb), "equal:", ok == gk, "%.3fs" % dt, rx.stats(), rx.timing())
            if block is None:
                for c, f in enumerate(fo):
                    ch = O.OracleChannel(rate, f, fc + f, tap_dec=True); ch.feed(x, fmt)
                    d = ch.dec(); g = rx.debug_dec(0, c)
                    n = min(len(d), len(g))
                    print("  dec ch", c, len(d), len(g), "bit-equal:", np.array_equal(d[:n].view(np.uint32), g[:n].view(np.uint32)))
                    # df/ppm/trig equality
                om = {(b.trig_dec): b for b in ch.blocks()}
                # atan2 device check
                rng = np.random.default_rng(5)
                y = rng.standard_normal(1 << 20).astype(np.float32) * 100; xx = rng.standard_normal(1 << 20).astype(np.float32) * 100
                ga = rx.debug_atan2f(y, xx)
                L = O.lib()
                import ctypes as C
                ca = np.array([L.vo_atan2f(float(a), float(b)) for a, b in zip(y[:20000], xx[:20000])], np.float32)
                print("  atan2 device==libm on 20000:", np.array_equal(ca.view(np.uint32), ga[:20000].view(np.uint32)))
            for a in gb[:3]:
                print("   ", a.chn, a.nbrow, a.nlbyte, a.df, a.ppm, a.trig_dec, a.end_dec, a.trig_sample)
            rx.close()
        for b in ob[:3]:
            print("  o ", b.chn, b.nbrow, b.nlbyte, b.df, b.ppm, b.trig_dec, b.end_dec)

main()
