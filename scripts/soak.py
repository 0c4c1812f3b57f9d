"""dev aid: randomised parity soak -- many seeds, rates, formats, push sizes; GPU bursts must equal the oracle's"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vdlm2dec_amd import synth
from vdlm2dec_amd.demod import Receiver, plan_channels
from oracle import oracle as O

FC = 136_975_000
n_ok = n_bad = 0
t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    rate = int(rng.choice([2_000_000, 2_000_000, 2_000_000, 5_000_000, 10_000_000]))
    nch = int(rng.integers(1, 9))
    fos = [int(f * rate / 2_000_000) // 25000 * 25000 for f in synth.DEFAULT_FO_8CH[:nch]]
    fmt = str(rng.choice(["cs16", "cu8", "cs16"]))
    ns = int(rng.integers(3, 14)) * 1_000_000 * (rate // 2_000_000 if rate > 2_000_000 else 1) // (2 if rate > 2_000_000 else 1)
    dens = float(rng.choice([3.0, 8.0, 20.0, 40.0])) * rate / 2_000_000
    spec = synth.random_scenario(rate, fos, ns, seed=seed, bursts_per_s=dens, info_max=int(rng.choice([60, 240, 900])),
                                 )
    raw = synth.synth_stream(spec, fmt)
    ob = O.run_oracle(raw, fmt, rate, fos, FC)
    want = sorted(b.key() for b in ob)
    block = int(rng.choice([ns, ns // 2 + 17, 1_234_567, 400_000, 2_000_000, 65536]))
    frames_too = bool(seed & 1)  # every other scenario also runs the block path in the pipeline
    # the knobs the library adapts by itself, exercised from the start: repair rounds (the last one is the complete
    # rescan), short parts for long pushes, clusters withheld from K2b, the region scan dropped (verify must catch up)
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
    if mode in (1, 4):
        from vdlm2dec_amd import lib as _lib
        flags = _lib.F_TEST_NOREGION
    with Receiver(rate, plan_channels(FC, fos), fmt=fmt, max_push=max(block, 1 << 16), frames=frames_too, flags=flags, testhooks=True) as rx:
        if seed & 2:
            # two pushes in flight: hand-offs back to back, what has finished is taken in between (vdl2gpu_poll_ready),
            # everything else at the end -- the front stage of one push runs beside the back stage of the one before
            per = O.PER_SAMPLE[fmt]
            bl, nsm = [], raw.size // per
            for s0 in range(0, nsm, block):
                rx.push(raw[s0 * per:min(nsm, s0 + block) * per])
                bl += rx.poll_ready()
            bl += rx.poll()
            got = sorted((b.chn, b.nbrow, b.nlbyte, b.data) for b in bl)
        else:
            got = sorted((b.chn, b.nbrow, b.nlbyte, b.data) for b in rx.run(raw, block=block))
        gotf = sorted(rx.poll_frames()) if frames_too else []
        st = rx.stats()
    ok = got == want
    if frames_too:
        wantf = sorted((0, b.chn, f) for b in ob for f in O.frames_of_block(b.nbrow, b.nlbyte, b.data))
        ok = ok and gotf == wantf
    n_ok += ok; n_bad += (not ok)
    print("seed %d rate %d ch %d %s ns %d dens %.0f block %d mode %d%s: %d bursts %s redos %d" % (seed, rate, nch, fmt, ns, dens, block, mode, " pipelined" if seed & 2 else "", len(want), "OK" if ok else "MISMATCH", st["serial_redos"]), flush=True)
    seed += 1
print("soak: %d ok, %d bad" % (n_ok, n_bad))
sys.exit(1 if n_bad else 0)
