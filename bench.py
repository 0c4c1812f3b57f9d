#!/usr/bin/env python3
"""bench.py -- throughput of the IQ->bursts hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4]

Workloads (BASELINE.json `configs`; config.workload carries the string):
  --config 2 (default)  configs[1]: 8 channels @ 2 MS/s, one wideband cs16 stream per GPU, 67.2 MS per step
  --config 3            configs[2]: 8 channels @ 10 MS/s (SDRCLK 2500, LO table 400), 268.8 MS per step
  --config 4            configs[3]: 512 replayed channels = 64 streams x 8 channels over 8 GPUs: 8 streams per GPU
                        (weak scaling: every rank decodes 8 streams whatever N is)
Synthetic D8PSK bursts (vdlm2dec_amd.synth: Poisson arrivals per channel, 8..60 LSB, +-400 Hz carrier offset,
AWGN).  A "step" is one vdl2gpu_push() of samples already resident in HBM plus delivery of the decoded msgblk
records to the host; three input buffers at different addresses are pushed in turn (nothing of the 268.8 MB a
push reads is in the 256 MB Infinity Cache when it is read again).  `value` = wideband input samples consumed
per second, whole job, all channels demodulated.

N > 1: `--gpus N` starts N ranks itself (python -m torch.distributed.run, 127.0.0.1) unless it is already running
under one (WORLD_SIZE set, as the driver launches it), and fails if fewer than N GPUs are visible.  The path shards
by independent wideband stream (vdlm2dec_amd/shard.py): no data-path collective; RCCL carries the
timing barrier / max, the parity verdicts and the gather of the packed burst records to rank 0.

Extra objects on the JSON line:
  roofline      the channeliser's full-rate kernel (k1_fast at 2 MS/s, k1_pp otherwise -- the only kernel that touches
                the wideband stream): algorithmic bytes = sample bytes x samples, read once for all 8 channels (SURVEY.md
                8d), over its mean launch time from HIP events on the library's own stream, inside the timed region.
                traffic_from_profiles = HBM bytes per launch from the committed rocprofv3 PMC passes of this command.
  kernels_ms    mean per-step device time of each stage, same events (kernel intervals only).
  parity        EVERY burst of the timed region compared with the oracle: the stream is tile-periodic, so the oracle's
                bursts of one steady-state tile are what every tile of the timed region must contain, record for record
                (channel, instant, nbrow, nlbyte, carrier estimate, all 2040 data bytes).  A mismatch nulls `value` and
                the process exits 1.
  cpu_baseline  the real reference (oracle/_ref: its own sources, its own compile flags) on this host's cores on a bounded
                sample; beside it ("port") the oracle, the CPU restatement pinned bit-equal to it.
  configs       (N = 1, default workload only; --no-extra skips it) the other BASELINE.json configurations measured in the
                same run, each checked against the oracle like the headline (value withheld on a mismatch):
                config3_8ch_10MSps, config4_share_8x8ch (one GPU's share of the 512 channels), config2_busy_15 / _30
                (the headline's workload at 15 and 30 bursts per second and channel offered), config5_live_ring (32768-sample
                cu8 blocks through vdl2gpu_ring_*, paced every 16.384 ms: p50 / p99 / max of commit -> bursts on the host).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FC = 136_975_000
TILE = 4_200_000          # samples per generated tile: a whole number of schedule periods at every rate
HBM_PEAK_GBS = 8000.0     # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
FO8_10MS = (-2_250_000, -1_750_000, -1_250_000, -300_000, 475_000, 1_000_000, 1_525_000, 2_300_000)

CONFIGS = {
    2: dict(workload="configs[1]: 8 channels @ 2 MS/s on 1xMI355X, synthetic D8PSK bursts, LDS-staged FIR",
            rate=2_000_000, streams=1, tiles=16, fos=None),
    3: dict(workload="configs[2]: 8 channels @ 10 MS/s (airspy-rate) on 1xMI355X, wider decimation chain",
            rate=10_000_000, streams=1, tiles=64, fos=FO8_10MS),
    4: dict(workload="configs[3]: 512 replayed channels sharded across 8xMI355X via RCCL/xGMI (offline bulk-decode): "
                     "8 streams x 8 channels per GPU",
            rate=2_000_000, streams=8, tiles=16, fos=None),
}


def source_key() -> str:
    """sha256 over the sources libvdl2gpu.so is built from (what __graft_entry__.build_hip watches): the key a committed PMC profile
    carries (scripts/make_profiles.sh) -- bench.py quotes a profile's traffic only for the tree it was measured on"""
    import hashlib
    csrc = os.path.join(ROOT, "vdlm2dec_amd", "csrc")
    files = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h", ".inc"))] + [os.path.join(ROOT, "include", "vdl2gpu.h")]
    hsh = hashlib.sha256()
    for f in files:
        hsh.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return hsh.hexdigest()[:16]


def synth_default():
    from vdlm2dec_amd import synth
    return synth.DEFAULT_FO_8CH


def make_tile(seed: int, fmt: str, rate: int, fos, bursts_per_s: float = 4.0):
    from vdlm2dec_amd import synth
    spec = synth.random_scenario(rate, fos, TILE, seed=seed, bursts_per_s=bursts_per_s, info_max=240)   # the same traffic per channel-second at every rate
    return spec, synth.synth_stream(spec, fmt)


def make_variants(seeds, fmt: str, rate: int, fos, bursts_per_s: float = 4.0):
    """several different recordings of one tile length (numpy releases the GIL: one thread each)"""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max(1, min(len(seeds), os.cpu_count() or 1))) as ex:
        return list(ex.map(lambda sd: make_tile(sd, fmt, rate, fos, bursts_per_s)[1], seeds))


def tile_order(i: int, ntiles: int, nbuf: int, nvar: int) -> int:
    """which of a stream's `nvar` recordings is tile i of the stream: a push is `ntiles` tiles, recordings in turn; the
    `nbuf` device buffers pushed in turn start one recording apart, so consecutive pushes differ as well"""
    return ((i % ntiles) + (i // ntiles) % nbuf) % nvar


def device_buffers(variants, ntiles, nbuf, dev):
    """variants[stream][recording] -> nbuf tensors [stream, ntiles * tile] laid out as tile_order() says"""
    import torch
    dv = [[torch.from_numpy(v).to(dev) for v in per] for per in variants]
    return [torch.stack([torch.cat([per[tile_order(b * ntiles + k, ntiles, nbuf, len(per))] for k in range(ntiles)]) for per in dv]).contiguous()
            for b in range(nbuf)]


def cpu_baseline(raw: np.ndarray, fmt: str, fos, rate: int, budget_s: float = 12.0):
    """Oracle ("port" of the reference path) on host cores: one thread per channel, like the reference's one
    rcv_thread per channel (main.c:228-231)."""
    from oracle import oracle as O
    O.lib()
    per = O.PER_SAMPLE[fmt]
    n_total = raw.size // per
    probe = min(n_total, 1_000_000)
    ch = O.OracleChannel(rate, fos[0], FC + fos[0])
    t0 = time.perf_counter()
    ch.feed(raw[:probe * per], fmt)
    one = (time.perf_counter() - t0) / probe
    ch.close()
    ncores = os.cpu_count() or 1
    nthreads = min(len(fos), ncores)
    passes = int(np.ceil(len(fos) / nthreads))
    reps = max(1, int(budget_s / (one * n_total * passes)))
    n = n_total * reps
    chans = [O.OracleChannel(rate, fo, FC + fo, chn=i) for i, fo in enumerate(fos)]

    def work(idx):
        for c in range(idx, len(chans), nthreads):
            for _ in range(reps):
                chans[c].feed(raw, fmt)       # ctypes releases the GIL

    th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    nb = sum(len(c.blocks()) for c in chans)
    for c in chans:
        c.close()
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": n / dt / 1e6, "unit": "MS/s", "cores": nthreads, "kind": "port",
            "sample": f"{n} samples ({reps} x the 4.2 MS tile of the same recording), {len(fos)} channels, {nthreads} threads "
                      f"(1 thread/channel), {dt:.1f} s wall, {nb} bursts; single-thread single-channel "
                      f"{1.0 / one / 1e6:.1f} MS/s; host CPU: {model} x{ncores}"}


def cpu_baseline_reference(raw: np.ndarray, fmt: str, fos, rate: int, budget_s: float = 10.0):
    """The REAL reference on host cores: oracle/_ref/ref_rtl[_ofast] is vdlm2dec's own d8psk.c/viterbi.c/vdlm2.c/crc.c/
    rs.c compiled where they lie (oracle/Makefile; the _ofast build with the reference's own -Ofast -march=native)
    behind a harness that does what rtl.c's in_callback and main.c do: one producer mixing and decimating blocks of
    RTLINBUFSZ for every channel, one rcv_thread per channel, two barriers per block.  Prebuilt in the container
    (the GPU box has no /root/reference); returns None where the binary is missing or does not run on this host."""
    import subprocess
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    if fmt == "f32":
        names = ["ref_air_ofast", "ref_air"]
    elif fmt in ("cu8", "cs16", "cf32"):
        names = ["ref_rtl_ofast", "ref_rtl"]
    else:
        return None
    per = {"cu8": 2, "cs16": 2, "cf32": 2, "f32": 1}[fmt]
    n_tile = raw.size // per
    reps = max(1, min(32, (512 << 20) // max(1, raw.nbytes)))       # at most 512 MB of input
    for tmpdir in (["/dev/shm"] if os.path.isdir("/dev/shm") else []) + [None]:
        try:
            return _reference_run(raw, fmt, fos, rate, budget_s, names, n_tile, reps, tmpdir, here)
        except OSError:
            continue        # e.g. a 64 MB /dev/shm: try the ordinary temporary directory, then give up
        except Exception:   # a baseline leg must never take the bench down (a timed-out run, an unreadable output)
            return None
    return None


def _reference_run(raw, fmt, fos, rate, budget_s, names, n_tile, reps, tmpdir, here):
    import subprocess
    import tempfile
    res = None
    with tempfile.TemporaryDirectory(dir=tmpdir) as td:
        path = os.path.join(td, "iq.bin")
        with open(path, "wb") as f:
            for _ in range(reps):
                f.write(raw.tobytes())
        n = n_tile * reps
        fo = ",".join(str(int(x)) for x in fos)
        fr = ",".join(str(int(FC + x)) for x in fos)
        for name in names:
            exe = os.path.join(here, "oracle", "_ref", name)
            if not os.path.exists(exe):
                continue
            cmd = [exe, path, fmt, str(rate), fo, fr, os.path.join(td, "out.txt")]
            try:
                t0 = time.perf_counter()
                subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
                one = time.perf_counter() - t0
            except (subprocess.SubprocessError, OSError):
                continue                                          # e.g. -march=native of another machine
            runs, total = 1, one
            while total < budget_s and runs < 8:
                t0 = time.perf_counter()
                subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
                total += time.perf_counter() - t0
                runs += 1
            nb = sum(1 for ln in open(os.path.join(td, "out.txt")) if ln.startswith("B"))
            dropin = _dropin_replay(here, td, path, fmt, rate, fo, fr, n)
            res = {"value": n * runs / total / 1e6, "unit": "MS/s", "cores": len(fos) + 1, "kind": "reference", "dropin_replay": dropin,
                   "sample": f"{runs} runs of oracle/_ref/{name} (vdlm2dec's own sources"
                             f"{', its own -Ofast -march=native' if name.endswith('_ofast') else ', -O2'}; producer + one rcv_thread per "
                             f"channel as in rtl.c/main.c) over {n} samples ({reps} x the 4.2 MS tile) from a file in "
                             f"{'memory' if tmpdir else 'tmp'}, {len(fos)} channels, {total:.1f} s wall, {nb} bursts in the last run (all channels in "
                             f"one process is this harness's timing mode: the count varies by a few per cent from run to run; "
                             f"parity is pinned one channel per process, tests/test_oracle_vs_ref.py)"}
            break
    return res


def _dropin_replay(here, td, path, fmt, rate, fo, fr, n):
    """The literal drop-in over the very file the CPU reference was just timed on: oracle/_ref/ref_rtl_gpu = the same harness
    (producer converting blocks of RTLINBUFSZ into Cbuff, two barriers per block) with the reference's UNCHANGED vdlm2.c /
    rs.c / crc.c behind dropin/vdl2gpu_rcv.c + libvdl2gpu.so instead of d8psk.c / viterbi.c.  The harness reports the time from
    the first hand-off to the last burst delivered (process start and HIP initialisation are not replay time)."""
    import re
    import subprocess
    exe = os.path.join(here, "oracle", "_ref", "ref_rtl_gpu")
    if not os.path.exists(exe):
        return None
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = os.path.join(td, "out_gpu.txt")
    best = None
    try:
        for _ in range(3):
            r = subprocess.run([exe, path, fmt, str(rate), fo, fr, out], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=180, env=env)
            m = re.search(r"replay (\d+) samples ([0-9.]+) s", r.stderr or "")
            if r.returncode or not m:
                return {"error": (r.stderr or "")[-300:]}
            ns, sec = int(m.group(1)), float(m.group(2))
            if best is None or sec < best[1]:
                best = (ns, sec)
        nb = sum(1 for ln in open(out) if ln.startswith("B"))
        nf = sum(1 for ln in open(out) if ln.startswith("F"))
        # what the hand-off protocol alone allows on this host: the same executable with the shim doing nothing but the barriers
        ceil = None
        r = subprocess.run([exe, path, fmt, str(rate), fo, fr, out], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=180,
                           env={**env, "VDL2GPU_RCV_NULL": "1"})
        m = re.search(r"replay (\d+) samples ([0-9.]+) s", r.stderr or "")
        if not r.returncode and m:
            ceil = int(m.group(1)) / float(m.group(2)) / 1e6
    except (subprocess.SubprocessError, OSError) as e:
        return {"error": repr(e)}
    return {"value": best[0] / best[1] / 1e6, "unit": "MS/s", "samples": best[0], "seconds": best[1], "bursts": nb, "frames": nf,
            "protocol_ceiling": ceil,
            "what": "oracle/_ref/ref_rtl_gpu (the reference's unchanged host path behind the drop-in shim) over the file the CPU reference was timed on, "
                    "best of 3; first Cbuff hand-off -> last burst through decodeVdlm2(); the producer (sample conversion into Cbuff, two barriers "
                    "per 32768 samples among nbch + 1 threads, rtl.c:283-294) is the harness's, on one host thread; protocol_ceiling = the same executable with the shim "
                    "doing nothing but the barriers (VDL2GPU_RCV_NULL=1): what the reference's hand-off protocol allows on this host"}


def oracle_stream(tiles, fmt: str, rate: int, fos, ntiles: int, order=None):
    """The oracle over the very stream the GPU is fed -- `ntiles` tiles, tile i being recording order(i) of `tiles`, from the first sample on
    (the sync detector's timing class is carried from burst to burst, so what a tile decodes to depends on everything
    before it: there is no shortcut through periodicity).  One thread per channel, like the reference's one rcv_thread
    per channel (main.c:228-231).  Returns (packed records with absolute instants, seconds of wall time)."""
    from oracle import oracle as O
    from vdlm2dec_amd import shard
    O.lib()
    out = [None] * len(fos)

    def work(c):
        ch = O.OracleChannel(rate, fos[c], FC + fos[c], chn=c)
        for i in range(ntiles):
            ch.feed(tiles[order(i)] if order else tiles, fmt)          # ctypes releases the GIL
        out[c] = ch.blocks()
        ch.close()

    th = [threading.Thread(target=work, args=(c,)) for c in range(len(fos))]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    bl = [b for per in out for b in per]
    rec = np.zeros(len(bl), shard.REC_DTYPE)
    for i, b in enumerate(bl):
        rec[i] = (0, b.chn, b.nbrow, b.nlbyte, int(np.float32(b.df).view(np.uint32)), 0, b.trig_dec, b.end_dec,
                  np.frombuffer(b.data, np.uint8))
    return rec, dt


def canon(recs: np.ndarray) -> bytes:
    order = np.lexsort((recs["trig_dec"], recs["chn"], recs["stream"]))
    return recs[order].tobytes()


def oracle_streams(tiles, fmt, rate, fos, ntiles, order=None):
    """oracle_stream() for several streams side by side (one thread per channel and stream)"""
    res = [None] * len(tiles)

    def work(i):
        res[i] = oracle_stream(tiles[i], fmt, rate, fos, ntiles, order)

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(tiles))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return [r[0] for r in res], max(r[1] for r in res)



def prefault(buf) -> None:
    """Touch every page of a host buffer the records will be written into: a fresh ctypes array is lazily mapped zero pages, and
    vdl2gpu_poll*() writing 2 MB of records per step into untouched pages paid ~500 first-touch page faults per step inside the timed
    region (in_poll_ready 0.10-0.25 ms by box) -- the cost of THIS script's allocation, not of the hand-off; a consumer that reuses
    its buffers never sees it."""
    v = np.frombuffer(buf, dtype=np.uint8)
    v[::4096] = 0
    v[-1:] = 0

def run_leg(name, workload, local, rate, fos, fmt, nstr, ntiles, bursts_per_s, steps, warmup, seed0, check_streams=4, stream_base=0, fence=None,
            want_records=False, repeats=3):
    """One more workload inside the same `bench.py --gpus 1` run (the `configs` object of the JSON line): resident input,
    pushes of ntiles x 4.2 MS per stream, bursts delivered to the host -- measured like the headline (pipelined, everything
    drained before the clock stops) and checked like it: every burst of the WHOLE run (first push included) against the
    oracle run over the same stream; `value` is withheld on a mismatch.  Also reported: the very first push of the handle
    (synchronous: what a cold start or a sudden load costs), the slowest of three synchronous pushes afterwards, and how
    much went through the serial machine."""
    import torch
    from vdlm2dec_amd import lib as _lib
    from vdlm2dec_amd import shard
    from vdlm2dec_amd.demod import Receiver, plan_channels
    dev = torch.device("cuda", local)
    sample_bytes = {"cs16": 4, "cu8": 2}[fmt]
    batch = ntiles * TILE
    tile_dec = TILE * 21 // (rate // 4000)
    NBUF = 2
    nvar = min(16 if nstr == 1 else 4, ntiles)        # different recordings in turn (see tile_order): no recording twice in a push of one stream (several
                                                          # streams: four each -- 128 recordings would take the leg's synthesis past a minute)
    # (stream_base: the global index of this rank's first stream in an N > 1 run -- every stream of the job has its own recordings)
    flat = make_variants([seed0 + stream_base + g + 1000 * v for g in range(nstr) for v in range(nvar)], fmt, rate, fos, bursts_per_s)
    variants = [flat[g * nvar:(g + 1) * nvar] for g in range(nstr)]
    order = lambda i: tile_order(i, ntiles, NBUF, nvar)
    dbufs = device_buffers(variants, ntiles, NBUF, dev)
    stride_bytes = dbufs[0].stride(0) * dbufs[0].element_size()
    cap = 1 << 19
    store = (_lib.BurstT * cap)()
    prefault(store)
    nrec = 0
    npush = 0
    with Receiver(rate, [plan_channels(FC, fos)] * nstr, fmt=fmt, max_push=batch, device=local, max_bursts=1 << 18) as rx:
        def drain(ready_only):
            nonlocal nrec
            while True:
                room = min(16384, cap - nrec)
                if room <= 0:
                    raise RuntimeError("bench.py: record store full")
                ptr = C.cast(C.byref(store, nrec * C.sizeof(_lib.BurstT)), C.POINTER(_lib.BurstT))
                n = rx.poll_ready_raw(ptr, room) if ready_only else rx.poll_raw(ptr, room)
                nrec += n
                if n < room:
                    break

        def push():
            nonlocal npush
            rx.push_device(dbufs[npush % NBUF].data_ptr(), batch, stride_bytes)
            npush += 1

        def sync_push():
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            push()
            drain(False)
            return (time.perf_counter() - t0) * 1e3

        # (the headline run before this leg has loaded the library's code objects: the first push is a cold HANDLE, not a cold process)
        first_ms = sync_push()
        for _ in range(max(0, warmup - 1)):
            push()
            drain(False)
        rx.sync()
        torch.cuda.synchronize(dev)
        if fence:
            fence()
        # `repeats` timed runs of `steps` pushes back to back (each drained and fenced): the line carries their median with min / max --
        # boxes differ by 5-20 % from lease to lease and one run in ten is 10-20 % slow on the same box; a single shot says little
        dts = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            for _ in range(steps):
                push()
                if not os.environ.get("BENCH_LEG_NO_DRAIN"):      # development: how fast without the record read-back in the loop
                    drain(True)
            drain(False)
            rx.sync()
            torch.cuda.synchronize(dev)
            if fence:
                fence()
            dts.append(time.perf_counter() - t0)
        dt = float(np.median(dts))
        slow = [sync_push() for _ in range(3)]
        st = rx.stats()
    total_tiles = npush * ntiles
    got = shard.pack_records(np.frombuffer(store, dtype=shard.BURST_DTYPE, count=nrec))
    tidx = got["trig_dec"] // tile_dec
    t_hi = total_tiles - 1
    ncheck = min(nstr, check_streams)
    exp, osec = oracle_streams(variants[:ncheck], fmt, rate, fos, total_tiles, order)
    nb, bad, first_bad = 0, 0, None
    for sidx in range(ncheck):
        e = exp[sidx]
        e = e[(e["trig_dec"] // tile_dec) < t_hi]
        g = got[(got["stream"] == sidx) & (tidx < t_hi)].copy()
        g["stream"] = 0
        nb += len(g)
        if canon(g) != canon(e):
            bad += 1
            if first_bad is None:       # what differs, for whoever reads the line: records only one side has, the first few by (channel, instant)
                gk = {(int(r["chn"]), int(r["trig_dec"])): r.tobytes() for r in g}
                ek = {(int(r["chn"]), int(r["trig_dec"])): r.tobytes() for r in e}
                first_bad = {"stream": sidx, "gpu_only": sorted(set(gk) - set(ek))[:6], "oracle_only": sorted(set(ek) - set(gk))[:6],
                             "different": sorted(k for k in set(gk) & set(ek) if gk[k] != ek[k])[:6],
                             "counts": [len(gk), len(ek)], "tile_dec": tile_dec}
                for k in first_bad["different"][:1]:
                    a, b = g[(g["chn"] == k[0]) & (g["trig_dec"] == k[1])][0], e[(e["chn"] == k[0]) & (e["trig_dec"] == k[1])][0]
                    first_bad["fields"] = {n: ([int(x) for x in np.flatnonzero(np.asarray(a[n]) != np.asarray(b[n]))[:8]] if np.ndim(a[n]) else [int(a[n]), int(b[n])])
                                           for n in g.dtype.names if not np.array_equal(a[n], b[n])}
    equal = bad == 0 and nb > 0
    value = nstr * batch * steps / dt / 1e6
    dec_total = st["dec_samples"] * 8 * nstr
    extra = {}
    if want_records:       # for the gather of an N > 1 run: this rank's records with GLOBAL stream indices, and its wall time
        pk = got[tidx < t_hi].copy()
        pk["stream"] += stream_base
        extra = {"_records": pk, "_dt": dt}
    return {**extra, "workload": workload, "value": value if equal else None, "unit": "MS/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "repeats": {"n": repeats, "what": "timed runs of `steps` pushes, back to back on this box; value / ms_per_step are their median",
                        "values": [nstr * batch * steps / x / 1e6 for x in dts] if equal else None,
                        "min": (nstr * batch * steps / max(dts) / 1e6) if equal else None, "max": (nstr * batch * steps / min(dts) / 1e6) if equal else None},
            "warmup": warmup, "fmt": fmt, "sdrinrate": rate, "streams": nstr, "channels": 8 * nstr, "samples_per_step": batch * nstr,
            "bursts_per_s_per_channel_offered": bursts_per_s, "recordings_per_stream": nvar, "bursts_per_step": int(round(nrec / max(1, npush))),
            "first_push_ms": first_ms, "max_push_ms": max(slow),
            "serial_samples_frac": st["serial_samples"] / max(1, dec_total), "repairs": st["repairs"], "serial_redos": st["serial_redos"], "overflowed": st["overflowed"],
            "parity": {"equal": equal, "bursts_checked": nb, "streams_checked": ncheck, "tiles_checked": t_hi, "oracle_seconds": osec,
                       "what": "every burst from the first sample of the run on, vs the oracle over the same stream",
                       **({"mismatch": first_bad} if first_bad else {})}}


def live_leg(local=0, nblocks=300, paced=True, nslots=8, bursts_per_s=8.0, seed=77):
    """configs[4]: the live path.  A producer hands 32768-sample cu8 blocks (one RTL-SDR USB transfer,
    RTLINBUFSZ = 65536 bytes, vdlm2.h:35: 16.384 ms of air time at 2 MS/s) to the ingest ring -- acquire a page-locked slot,
    fill it in place (what rtlsdr_read_async's buffer copy / in_callback do, rtl.c:274-295; 8 slots like its 8 asynchronous
    buffers, rtl.c:302), commit -- paced at real time, and a consumer takes the bursts: latency = from vdl2gpu_ring_commit()
    of a block until every burst that ends in it is on the host (vdl2gpu_poll returns).  The bursts must be the oracle's."""
    import torch
    from oracle import oracle as O
    from vdlm2dec_amd import lib as _lib
    from vdlm2dec_amd import synth
    from vdlm2dec_amd.demod import Receiver, plan_channels
    rate, blk = 2_000_000, 32768
    fos = synth.DEFAULT_FO_8CH
    spec = synth.random_scenario(rate, fos, nblocks * blk, seed=seed, bursts_per_s=bursts_per_s, info_max=200)
    raw = synth.synth_stream(spec, "cu8")
    want = sorted(b.key() for b in O.run_oracle(raw, "cu8", rate, fos, FC))
    rawb = raw.view(np.uint8).reshape(-1)
    period = blk / rate
    buf = (_lib.BurstT * 4096)()
    got, lat, late = [], [], 0
    with Receiver(rate, plan_channels(FC, fos), fmt="cu8", max_push=blk, device=local) as rx:
        rx.ring_init(blk, nslots=nslots)
        t_start = time.perf_counter()
        for i in range(nblocks):
            if paced:
                due = t_start + i * period
                while True:
                    now = time.perf_counter()
                    if now >= due:
                        break
                    time.sleep(min(0.002, due - now))
                late += (time.perf_counter() - due) > period
            slot = rx.ring_acquire()
            slot[0, :2 * blk] = rawb[2 * i * blk:2 * (i + 1) * blk]     # the producer's fill
            t0 = time.perf_counter()
            rx.ring_commit(blk)
            n = rx.poll_raw(buf, 4096)                                  # waits for everything committed so far
            lat.append((time.perf_counter() - t0) * 1e3)
            for k in range(n):
                b = buf[k]
                got.append((b.chn, b.nbrow, b.nlbyte, bytes(b.data)))
        wall = time.perf_counter() - t_start
        st = rx.stats()
    lat = np.array(lat[8:])             # the first blocks load the kernels' code objects
    equal = sorted(got) == want and len(want) > 0
    return {"workload": "configs[4]: 8 ch live rtl_sdr USB -> pinned ring -> GPU demod, real-time latency path (USB simulated: "
                        "32768-sample cu8 blocks every 16.384 ms)",
            "blocks": nblocks, "paced": bool(paced), "ring_slots": nslots, "block_air_time_ms": period * 1e3,
            "latency_ms": {"p50": float(np.median(lat)), "p99": float(np.percentile(lat, 99)), "max": float(lat.max()),
                           "definition": "vdl2gpu_ring_commit(block) -> vdl2gpu_poll() has returned every burst ending in it"} if equal else None,
            "blocks_started_late": int(late), "wall_s": wall, "bursts": len(got), "serial_samples": st["serial_samples"],
            "parity": {"equal": equal, "bursts_checked": len(want), "what": "every burst of the run vs the oracle over the same recording"}}


def ring_rate(local, rate, fos, fmt, batch, recs, ntiles):
    """PCIe-inclusive rate for one format: pushes of `batch` samples from page-locked ring slots, bursts delivered."""
    from vdlm2dec_amd import lib as _lib
    from vdlm2dec_amd.demod import Receiver, plan_channels
    buf = (_lib.BurstT * 16384)()
    raw = np.concatenate([recs[k % len(recs)] for k in range(ntiles)]).view(np.uint8).reshape(1, -1)
    with Receiver(rate, plan_channels(FC, fos), fmt=fmt, max_push=batch, device=local, max_bursts=1 << 18) as rx:
        rx.ring_init(batch, nslots=3)
        nb = batch * rx.sample_bytes
        for _ in range(3):
            slot = rx.ring_acquire()
            slot[:, :nb] = raw[:, :nb]
            rx.ring_commit(batch)
        while rx.poll_raw(buf, 16384) == 16384:
            pass
        rx.sync()
        t0 = time.perf_counter()
        for _ in range(6):
            rx.ring_acquire()
            rx.ring_commit(batch)
            while rx.poll_ready_raw(buf, 16384) == 16384:
                pass
        while rx.poll_raw(buf, 16384) == 16384:
            pass
        rx.sync()
        dt = time.perf_counter() - t0
    return {"value": 6 * batch / dt / 1e6, "unit": "MS/s", "pushes": 6, "fmt": fmt, "bytes_per_sample": 2 if fmt == "cu8" else 4,
            "host_GBps": 6 * batch * (2 if fmt == "cu8" else 4) / dt / 1e9,
            "note": "as host_ring, for the reference's native cu8 (rtl.c:285-292): half the bytes per sample over the same link"}


def extra_legs(local):
    """The `configs` object: configs[2], the per-GPU share of configs[3], configs[1] on busy channels, configs[4]."""
    from vdlm2dec_amd import synth
    fo2 = synth.DEFAULT_FO_8CH
    legs = {}
    plan = [
        ("config3_8ch_10MSps", dict(workload=CONFIGS[3]["workload"], rate=10_000_000, fos=FO8_10MS, fmt="cs16", nstr=1, ntiles=64,
                                    bursts_per_s=4.0, steps=8, warmup=3, seed0=1234)),
        ("config4_share_8x8ch", dict(workload=CONFIGS[4]["workload"] + " -- one GPU's share", rate=2_000_000, fos=fo2, fmt="cs16", nstr=8,
                                     ntiles=16, bursts_per_s=4.0, steps=6, warmup=3, seed0=1234)),
        ("config2_busy_15", dict(workload=CONFIGS[2]["workload"] + " -- 15 bursts/s/channel offered", rate=2_000_000, fos=fo2, fmt="cs16",
                                 nstr=1, ntiles=16, bursts_per_s=15.0, steps=8, warmup=3, seed0=77)),
        ("config2_busy_30", dict(workload=CONFIGS[2]["workload"] + " -- 30 bursts/s/channel offered (channels saturated)", rate=2_000_000,
                                 fos=fo2, fmt="cs16", nstr=1, ntiles=16, bursts_per_s=30.0, steps=8, warmup=3, seed0=77)),
    ]
    for name, kw in plan:
        try:
            legs[name] = run_leg(name, local=local, **kw)
        except Exception as e:      # a leg must not cost the run its line
            legs[name] = {"error": repr(e)}
    try:
        legs["config5_live_ring"] = live_leg(local)
    except Exception as e:
        legs["config5_live_ring"] = {"error": repr(e)}
    return legs


def respawn(args, argv):
    """`python bench.py --gpus N` on its own: become N ranks."""
    import torch
    have = torch.cuda.device_count()
    if have < (1 if args.share_gpu else args.gpus):
        sys.exit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible (no CPU fallback; ranks share a GPU only with --backend gloo --share-gpu)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--tiles", type=int, default=0, help="tiles of 4.2 MS per step and stream (0 = the config's)")
    ap.add_argument("--fmt", default="cs16", choices=["cs16", "cu8"])
    ap.add_argument("--rate", type=int, default=0, help="override the config's SDRINRATE (5000000 / 6000000)")
    ap.add_argument("--streams", type=int, default=0, help="override the config's streams per GPU")
    ap.add_argument("--recordings", type=int, default=0, help="different synthetic recordings per stream, pushed in turn (0 = 16 for one stream, 4 otherwise)")
    ap.add_argument("--bursts-per-s", type=float, default=4.0, help="offered load per channel (development: the headline is 4)")
    ap.add_argument("--frames", action="store_true",
                    help="also run the block path (RS, HDLC, FCS: SURVEY 8f-1) on every push's bursts and collect the frames")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-ring", action="store_true", help="skip the PCIe-inclusive extra pass (ingest ring from pinned host memory)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the `configs` object (configs[2], [3]-share, busy channels, [4] live ring)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process group of an N > 1 run: nccl (= RCCL over xGMI, the default) or gloo (host sockets; with --share-gpu it lets the "
                         "whole N > 1 path run on a box with ONE GPU: RCCL refuses two ranks on one device)")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses cuda:0 (needs --backend gloo): the multi-rank path on a one-GPU box")
    ap.add_argument("--gather-out", default="", help="rank 0 writes the gathered records of the timed region here (.npy, shard.REC_DTYPE): for tests")
    args = ap.parse_args()
    if args.share_gpu and args.backend != "gloo":
        sys.exit("bench.py: --share-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn(args, sys.argv[1:])

    import torch
    import torch.distributed as dist
    from vdlm2dec_amd import lib as _lib
    from vdlm2dec_amd import shard, synth
    from vdlm2dec_amd.demod import Receiver, plan_channels

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.share_gpu:
        local = 0
    if torch.cuda.device_count() <= local:
        sys.exit(f"bench.py: rank {rank} has no GPU (LOCAL_RANK {local}, {torch.cuda.device_count()} visible)")
    if world > 1:
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    dev = torch.device("cuda", local)
    cdev = dev if (world > 1 and args.backend == "nccl") else torch.device("cpu")     # where the collectives' tensors live

    cfg = CONFIGS[args.config]
    rate = args.rate or cfg["rate"]
    nstr = args.streams or cfg["streams"]
    ntiles = args.tiles or cfg["tiles"]
    if rate == 2_000_000:
        fos = synth.DEFAULT_FO_8CH
    elif cfg["fos"] and rate == cfg["rate"]:
        fos = cfg["fos"]
    else:
        fos = tuple(int(f * rate / 2_000_000) // 25000 * 25000 for f in synth.DEFAULT_FO_8CH)
    nstreams_total = nstr * world
    mine = shard.shard_streams(nstreams_total, rank, world)        # this rank's global stream indices
    sample_bytes = 4 if args.fmt == "cs16" else 2
    batch = ntiles * TILE
    tile_dec = TILE * 21 // (rate // 4000)

    # Different 2.1 s recordings per stream, in turn (tile_order): a push of 16 tiles is 16 different recordings (33.6 s of air time in
    # which nothing repeats -- a repeated recording repeats its rare events with it: a noise trigger that exists in one timing class
    # only, one per ~100 channel-seconds, came back in every tile, each time behind the repair of the one before), and the three
    # device buffers pushed in turn start one recording apart; what a push reads was last touched two pushes ago.
    NBUF = 3
    nvar = max(1, args.recordings or min(16 if nstr == 1 else 4, ntiles))
    flat = make_variants([1234 + g + 1000 * v for g in mine for v in range(nvar)], args.fmt, rate, fos, args.bursts_per_s)
    variants = [flat[k * nvar:(k + 1) * nvar] for k in range(len(mine))]
    tiles_np = [per[0] for per in variants]
    order = lambda i: tile_order(i, ntiles, NBUF, nvar)
    dbufs = device_buffers(variants, ntiles, NBUF, dev)
    stride_bytes = dbufs[0].stride(0) * dbufs[0].element_size()

    rx = Receiver(rate, [plan_channels(FC, fos)] * nstr, fmt=args.fmt, max_push=batch, device=local, max_bursts=1 << 18,
                  frames=args.frames)
    cap = 1 << 20
    store = (_lib.BurstT * cap)()           # every record of the run lands here, straight from vdl2gpu_poll*()
    prefault(store)
    nrec = 0
    framebuf = (_lib.FrameT * 4096)() if args.frames else None
    nframes = [0]
    npush = [0]

    def drain(ready_only):
        nonlocal nrec
        while True:
            room = min(16384, cap - nrec)
            if room <= 0:
                raise RuntimeError("bench.py: record store full")
            ptr = C.cast(C.byref(store, nrec * C.sizeof(_lib.BurstT)), C.POINTER(_lib.BurstT))
            n = rx.poll_ready_raw(ptr, room) if ready_only else rx.poll_raw(ptr, room)
            nrec += n
            if n < room:
                break
        if args.frames:
            while True:
                m = (rx.L.vdl2gpu_poll_frames_ready if ready_only else rx.L.vdl2gpu_poll_frames)(rx.h, framebuf, 4096)
                if m < 0:
                    raise RuntimeError("vdl2gpu_poll_frames failed")
                nframes[0] += m
                if m < 4096:
                    break

    host_s = [0.0, 0.0]     # seconds this thread spent inside vdl2gpu_push / inside the poll calls (timed region)

    def step(pipelined):
        # one hand-off of resident samples + delivery of decoded msgblk records to the host.  pipelined: take what
        # earlier pushes have finished (vdl2gpu_poll_ready) while this push runs; everything is drained inside the
        # timed region after the last step.
        ta = time.perf_counter()
        rx.push_device(dbufs[npush[0] % NBUF].data_ptr(), batch, stride_bytes)
        tb = time.perf_counter()
        npush[0] += 1
        drain(ready_only=pipelined)
        host_s[0] += tb - ta
        host_s[1] += time.perf_counter() - tb

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False)
    rx.sync()
    rx.timing(reset=True)
    rec0 = nrec
    first_timed_tile = npush[0] * ntiles

    fence()
    host_s[0] = host_s[1] = 0.0
    rx.host_profile(reset=True)
    t0 = time.perf_counter()
    t_steps = []
    for _ in range(args.steps):
        step(True)
        t_steps.append(time.perf_counter())
    drain(False)                        # every burst of every step is on the host before the clock stops
    rx.sync()
    fence()
    dt = time.perf_counter() - t0
    hprof = rx.host_profile()
    rec1 = nrec
    last_timed_tile = npush[0] * ntiles            # exclusive
    tm = rx.timing(reset=True)
    st = rx.stats()
    # outside the timed region: the same hand-off a few times with nothing else on the GPU (each push finished before
    # the next starts), to tell what the channeliser does alone from what it does while it shares the GPU with the
    # previous push's demodulator (the timed region above)
    for _ in range(4):
        rx.push_device(dbufs[npush[0] % NBUF].data_ptr(), batch, stride_bytes)
        npush[0] += 1
        rx.sync()
        drain(False)
    tm_iso = rx.timing(reset=True)
    # also outside the timed region: the same hand-off from page-locked host memory through the ingest ring (SURVEY 8
    # f-2) -- the PCIe-inclusive rate.  Reported beside `value`, never as `value`.
    host_ring = None
    host_ring_cu8 = None
    if rank == 0 and world == 1 and not args.no_ring and args.config == 2:
        try:
            if args.fmt == "cs16":      # the reference's native RTL format through the same ring, on a handle of its own (2 bytes per sample)
                host_ring_cu8 = ring_rate(local, rate, fos, "cu8", batch, [v for v in make_variants([4321 + k for k in range(4)], "cu8", rate, fos, args.bursts_per_s)], ntiles)
            else:
                host_ring_cu8 = None
        except Exception as e:
            host_ring_cu8 = {"error": str(e)}
        try:
            hb = dbufs[0].cpu().numpy().view(np.uint8).reshape(nstr, -1)
            rx.ring_init(batch, nslots=3)
            nb = batch * rx.sample_bytes
            for _ in range(3):          # fill the three slots once: the producer's work is not what is measured
                slot = rx.ring_acquire()
                slot[:, :nb] = hb[:, :nb]
                rx.ring_commit(batch)
            drain(False)
            rx.sync()
            th = time.perf_counter()
            for _ in range(6):
                rx.ring_acquire()
                rx.ring_commit(batch)
                drain(True)
            drain(False)
            rx.sync()
            dth = time.perf_counter() - th
            host_ring = {"value": 6 * batch * nstr / dth / 1e6, "unit": "MS/s", "pushes": 6,
                         "note": "samples start in page-locked host memory (vdl2gpu_ring_acquire/commit): H2D copy "
                                 "on its own stream beside the previous push's kernels, bursts delivered to the host"}
            rx.timing(reset=True)
        except Exception as e:      # the extra measurement must not cost the run its line
            host_ring = {"error": str(e)}

    # ---- parity: every record of the timed region against the oracle run over the same stream from its first sample
    timed = np.frombuffer(store, dtype=shard.BURST_DTYPE, count=rec1)[rec0:]
    parity = None
    ok_local = True
    oracle_time = None
    if not args.no_parity:
        # bounded: the oracle handles ~25 MS/s per channel and thread; BENCH_ORACLE_TILES (1200) tiles per stream and at most four streams
        # per rank keep the check near a minute.  The default run (and the driver's) is covered completely.
        ncheck_tiles = min(last_timed_tile, int(os.environ.get("BENCH_ORACLE_TILES", "1200")))
        ncheck_streams = min(nstr, 4)
        packed = shard.pack_records(timed)
        tidx = packed["trig_dec"] // tile_dec
        # tiles wholly inside the timed region and the oracle's run, except the last one (a burst that is still open when
        # the stream ends is not delivered)
        t_lo, t_hi = first_timed_tile, min(last_timed_tile, ncheck_tiles) - 1
        nb_checked, bad, per_stream = 0, [], []
        oracle_time = 0.0
        for s in range(ncheck_streams):
            e, dt_o = oracle_stream(variants[s], args.fmt, rate, fos, ncheck_tiles, order)
            oracle_time += dt_o
            et = e["trig_dec"] // tile_dec
            e = e[(et >= t_lo) & (et < t_hi)]
            g = packed[(packed["stream"] == s) & (tidx >= t_lo) & (tidx < t_hi)].copy()
            g["stream"] = 0
            nb_checked += len(g)
            per_stream.append((len(g), len(e)))
            if canon(g) != canon(e):
                ge = {(int(r["chn"]), int(r["trig_dec"])): r.tobytes() for r in g}
                ee = {(int(r["chn"]), int(r["trig_dec"])): r.tobytes() for r in e}
                diff = sorted(k for k in set(ge) | set(ee) if ge.get(k) != ee.get(k))
                bad.append({"stream": s, "gpu": len(g), "oracle": len(e), "first_differences": [list(k) for k in diff[:6]]})
        ok_local = not bad and t_hi > t_lo and nb_checked > 0
        parity = {"level": "msgblk_t (pre-RS) bit-exact: every burst of the timed region vs the oracle run over the same stream "
                           "from its first sample (chn, instant, nbrow, nlbyte, df bits, 2040 data bytes)",
                  "tiles_in_timed_region": last_timed_tile - first_timed_tile, "tiles_checked": max(0, t_hi - t_lo),
                  "streams_checked": ncheck_streams, "streams": nstr, "bursts_checked": nb_checked,
                  "gpu_vs_oracle_bursts": per_stream, "mismatches": bad, "equal": ok_local,
                  "oracle_seconds": oracle_time}

    # ---- collection across ranks: counts, verdicts, and the packed records on rank 0
    if world > 1:
        t = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        okt = torch.tensor([1 if ok_local else 0], device=cdev, dtype=torch.int32)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok_all = bool(okt.item())
        g = shard.run_sharded(nstreams_total, lambda idx: shard.pack_records(timed, stream_offset=idx.start))
        allrecs, counts, _ = g
        total_bursts = int(sum(counts))
        # counts + per-rank digests reach every rank (40 bytes each); the records only rank 0, point to point in chunks
        gathered = {"records_on_rank0": int(len(allrecs)), "per_rank": counts,
                    "digest": shard.digest(allrecs).hex()[:16] if rank == 0 else None,
                    "digest_of_rank_digests": g.combined.hex()[:16], "backend": args.backend,
                    "ranks_share_one_gpu": bool(args.share_gpu)}
        if rank == 0 and args.gather_out:
            np.save(args.gather_out, allrecs)
    else:
        ok_all = ok_local
        total_bursts = int(len(timed))
        gathered = None
        if args.gather_out:
            np.save(args.gather_out, shard.pack_records(timed))

    rc = 0
    if rank == 0:
        pushes = max(1, tm["pushes"])
        k1_ms = tm["channelise_ms"] / pushes
        k2_ms = tm["demod_ms"] / pushes
        k2a_ms = tm["scan_ms"] / pushes
        k2b_ms = tm["cluster_ms"] / pushes
        k2c_ms = tm["resolve_ms"] / pushes
        k3_ms = tm["other_ms"] / pushes
        # the full-rate kernel covers all whole periods of a push (1 ms; k1_fast: superperiods of 4 ms) but the first and the last
        sdrclk = rate // 4000
        periods = (batch * 21 // sdrclk) // 84
        if rate == 2_000_000:
            fast_samples = (periods // 4 - 2) * 16 * sdrclk * nstr
        else:
            fast_samples = (periods - 2) * 4 * sdrclk * nstr
        kname = "k1_fast" if rate == 2_000_000 else "k1_pp"
        fast_ms = tm["channelise_fast_ms"] / max(1, tm["fast_pushes"])
        alg_bytes = float(fast_samples) * sample_bytes
        achieved = alg_bytes / (fast_ms * 1e-3) / 1e9 if fast_ms > 0 else 0.0
        iso_ms = tm_iso["channelise_fast_ms"] / max(1, tm_iso["fast_pushes"])
        value = world * nstr * batch * args.steps / dt / 1e6
        ms_step = dt / args.steps * 1e3
        # Steady state: `value` is defined over exactly --steps steps between two fences, so it contains the pipeline's fill and drain
        # (three pushes deep: about two steps' worth, 7 % of a 20-step run and 4 % of a 32-step one).  vdl2gpu_push() returns when
        # the push is enqueued and -- three being in flight -- the push three back has been collected, so the times the calls
        # return, from the fourth step to the last, are the pipeline's own rate whatever --steps is.
        steady = None
        if len(t_steps) >= 8:
            ss_ms = (t_steps[-1] - t_steps[3]) / (len(t_steps) - 4) * 1e3
            steady = {"ms_per_step": ss_ms, "value": world * nstr * batch / (ss_ms * 1e-3) / 1e6, "unit": "MS/s", "steps_used": len(t_steps) - 4,
                      "definition": "(return of the last vdl2gpu_push+poll_ready - return of the fourth) / steps between them: the pipelined rate without the fill "
                                    "and the final drain that `value` (exactly --steps steps between two fences, the contract's figure) contains"}
        traffic, traffic_src, whole_step = None, None, None
        try:
            pmf = [f for f in ("r06_bench_pmc_hbm.json", "r05_bench_pmc_hbm.json", "r04_bench_pmc_hbm.json") if os.path.exists(os.path.join(ROOT, "profiles", f))][0]
            pm = json.load(open(os.path.join(ROOT, "profiles", pmf)))
            key_now = source_key()
            if pm.get("source_key") != key_now:
                # a stale replay is worse than none: the counters of another build say nothing about this one's traffic
                traffic_src = (f"profiles/{pmf} was measured on sources {pm.get('source_key')} (commit {pm.get('head')}), this run is {key_now}: "
                               "not quoted -- regenerate with scripts/make_profiles.sh")
            elif args.config == 2 and not args.tiles and args.fmt == "cs16" and not args.rate and not args.streams:
                fk = [k for k in pm["FETCH_SIZE_KB_per_launch"] if kname in k][0]
                traffic = (2.0 * pm["FETCH_SIZE_KB_per_launch"][fk] + pm["WRITE_SIZE_KB_per_launch"][fk]) * 1024.0
                traffic_src = f"profiles/{pmf} (commit {pm.get('head')}, sources {key_now} = this run's): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this very command " \
                              "(2 x FETCH_SIZE + WRITE_SIZE per launch, the gfx950 correction of MI355X_MICROARCH.md), replayed from the committed file -- PMC counters cannot be read by the run itself"
                # the whole step from the same committed passes: every kernel's HBM traffic and vector instructions, launches per step as in the trace
                sqf = pmf.replace("_hbm", "_sq")
                sqj = json.load(open(os.path.join(ROOT, "profiles", sqf)))
                sq = sqj["per_launch"]
                per_step = pm.get("launches_per_step", {})
                if "per_step_bytes" in pm and "per_step" in sqj:      # every launch's counters summed, over the pushes of the profiled run
                    hbm = sum(pm["per_step_bytes"].values())
                    vinst = sum(v.get("SQ_INSTS_VALU", 0.0) for v in sqj["per_step"].values())
                else:
                    hbm = sum((2.0 * pm["FETCH_SIZE_KB_per_launch"].get(k, 0.0) + pm["WRITE_SIZE_KB_per_launch"].get(k, 0.0)) * 1024.0 * per_step.get(k, 1.0)
                              for k in pm["FETCH_SIZE_KB_per_launch"])
                    vinst = sum(v.get("SQ_INSTS_VALU", 0.0) * per_step.get(k, 1.0) for k, v in sq.items())
                floor_ms = vinst * 4 / 1024 / 2.4e9 * 1e3
                # per kernel: the SIMDs' time its vector instructions take (4 cycles a wave instruction, 1024 SIMDs, the 2.1 GHz the part sustains
                # under this load) against the time its launches last in the traced run of the same command: what share of the issue slots it
                # holds it uses (the rest it waits: LDS and memory round trips, barriers, other kernels' wavefronts)
                issue = None
                try:
                    import csv as _csv
                    ksf = pmf.replace("_pmc_hbm.json", "_kernel_stats.csv")
                    ks = {r["Name"].split("(")[0]: (int(r["Calls"]), float(r["AverageNs"])) for r in _csv.DictReader(open(os.path.join(ROOT, "profiles", ksf)))}
                    npush = max(1, ks.get("k2a_probe", (1, 0.0))[0])
                    issue = {}
                    for k, v in sqj.get("per_step", {}).items():
                        if k in ks and v.get("SQ_INSTS_VALU", 0.0) > 0:
                            busy_us = v["SQ_INSTS_VALU"] * 4 / 1024 / 2.1e9 * 1e6
                            held_us = ks[k][0] / npush * ks[k][1] / 1e3
                            issue[k.replace("void ", "")] = {"valu_us": round(busy_us, 1), "kernel_us": round(held_us, 1), "frac": round(busy_us / held_us, 3) if held_us > 0 else None}
                except (OSError, KeyError, ValueError):
                    issue = None
                whole_step = {"issue_utilisation": issue, "issue_utilisation_source": f"SQ_INSTS_VALU per step (profiles/{sqf}) x 4 cycles / 1024 SIMDs / 2.1 GHz over launches per step x "
                                                                                      f"AverageNs (profiles/{pmf.replace('_pmc_hbm.json', '_kernel_stats.csv')})",
                              "valu_wave_insts": vinst, "valu_floor_ms": floor_ms, "hbm_traffic_bytes": hbm, "step_over_valu_floor": ms_step / floor_ms if floor_ms > 0 else None,
                              "hbm_traffic_over_algorithmic": hbm / (batch * nstr * sample_bytes) if batch else None,
                              "source": f"profiles/{sqf} (SQ_INSTS_VALU) and profiles/{pmf}, kernels x launches per step; 4 cycles per wave instruction, 1024 SIMDs, 2.4 GHz"}
        except (OSError, KeyError, ValueError, IndexError):
            pass
        parity_ok = ok_all or args.no_parity
        # what actually bounds the channeliser: VALU issue.  The reference's D += x*w is eight separately rounded operations = four
        # packed FP32 instructions per sample and channel (no FMA: it rounds differently); a SIMD-32 issues a wave's packed
        # instruction in 4 cycles, a plain one in 2 (MI355X_MICROARCH.md: 157 TFLOP/s FP32 = 64 flop/clk/SIMD).
        valu = None
        if kname == "k1_fast" and iso_ms > 0:
            packed = fast_samples * 8 * 4 / 64.0                    # wave instructions: samples x channels x 4 / 64 lanes
            other = packed * (41.8 - 33.6) / 33.6                   # conversion, addressing, division (SQ_INSTS_VALU, profiles/r03_bench_pmc_sq.json)
            cyc = (packed * 4 + other * 2) / (256 * 4)              # per SIMD
            floor_ms = cyc / 2.4e9 * 1e3
            valu = {"packed_fp32_wave_insts": packed, "other_valu_wave_insts": other, "simd_cycles": cyc, "floor_ms_at_2400MHz": floor_ms,
                    "frac_alone": floor_ms / iso_ms, "frac_live": floor_ms / fast_ms if fast_ms > 0 else None,
                    "note": "share of the launch the SIMDs need for the mixer's instructions alone at the peak clock (the part sustains "
                            "2.0-2.1 GHz under this load): the kernel is within a third of its instruction-issue bound while it uses a "
                            "quarter of the HBM bandwidth -- the bound the `roofline` contract asks for (hbm) is not the one that binds"}
        out = {
            "metric": "IQ MS/s demodulated (8 ch, 2 MS/s cs16) + CRC-pass frame parity vs ref",
            "value": value if parity_ok else None, "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "steady_state": steady,
            "config": {"workload": (cfg["workload"] if world == 1 else
                                    (f"{world} x configs[1]: one 8-channel 2 MS/s stream per GPU, {world} GPUs, stream-sharded (weak scaling; the 512-channel job is configs.config4_512ch)"
                                     if args.config == 2 else cfg["workload"] + f" -- {world} of 8 GPUs" * (world != 8))) + ("" if not (args.rate or args.streams or args.tiles or args.bursts_per_s != 4.0) else
                                                       f" [overridden: {nstr} stream(s)/GPU, {rate / 1e6:g} MS/s, {ntiles} tiles, {args.bursts_per_s:g} bursts/s/channel offered]"),
                       "fmt": args.fmt, "samples_per_step": batch * nstr, "air_time_s_per_step": batch / rate,
                       "channels": 8, "streams_per_gpu": nstr, "streams_total": nstreams_total, "sdrinrate": rate,
                       "bursts_per_step": total_bursts / max(1, args.steps * world),
                       "x_real_time": value * 1e6 / rate, "parallelism": f"stream-sharded x{world}",
                       "rccl_world_size": dist.get_world_size() if world > 1 else 1,
                       "input_buffers": NBUF, "recordings_per_stream": nvar,
                       "input_note": f"{nvar} different synthetic recordings of {TILE / rate:.1f} s per stream, in turn; the {NBUF} device "
                                     "buffers pushed in turn start one recording apart (bench.py tile_order)"},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_from_profiles": traffic,
                         "whole_step": whole_step,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": fast_ms,
                         "whole_path_frac": (batch * nstr * sample_bytes / (ms_step * 1e-3) / 1e9) / HBM_PEAK_GBS,
                         "valu_issue": valu,
                         "alone": {"avg_launch_ms": iso_ms, "achieved": alg_bytes / (iso_ms * 1e-3) / 1e9 if iso_ms > 0 else 0.0,
                                   "frac": (alg_bytes / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if iso_ms > 0 else 0.0,
                                   "how": "4 pushes after the timed region, each synchronised before the next: no other kernel on the GPU"},
                         "note": "live: HIP events around the one full-rate launch of each push inside the timed region (every fourth push "
                                 "carries them), where it runs beside the PREVIOUS push's back stage -- its cluster kernel or verify pass, "
                                 "both of which fill the GPU by themselves: three pushes are in the pipeline (DESIGN.md 4, Host side), so the "
                                 "kernel takes longer here than alone (`alone`) while the step as a whole got shorter; algorithmic bytes = sample "
                                 "bytes x samples, read once for all 8 channels; the kernel also writes the 84 kS/s planes (2.7 B per "
                                 "input sample at 2 MS/s), which is intermediate traffic, not algorithmic (SURVEY.md 8d); `traffic` is "
                                 "a REPLAY of the committed rocprofv3 PMC passes over this command (traffic_source), not a measurement of this run"},
            "kernels_ms": {"k1_channelise": k1_ms, "k2a_scan": k2a_ms, "k2b_clusters": k2b_ms,
                           "k2c_resolve+k2d_gather": k2c_ms, "k3_compact": k3_ms, "demod_chain": k2_ms,
                           "note": "kernel intervals from HIP events; the front stage of one push (channeliser, scan) runs beside the back "
                                   "stage of the one before (clusters, resolver, verify) and the tail of the one before that: the intervals overlap and are longer than the "
                                   "kernels alone -- their sum (demod_chain + k1) exceeds ms_per_step, it is not a critical path"},
            "stats": {k: st[k] for k in ("sync_evals", "triggers", "header_rejects", "bursts", "deferrals",
                                           "candidates", "repairs", "serial_redos", "serial_samples", "overflowed")},
            "parity": parity,
            "host_ms_per_step": {"in_push": host_s[0] / args.steps * 1e3, "in_poll_ready": host_s[1] / args.steps * 1e3,
                                 "enqueue": hprof["enqueue"] / args.steps * 1e3, "wait_for_ring": hprof["wait_for_ring"] / args.steps * 1e3,
                                 "wait_input": hprof["wait_input"] / args.steps * 1e3, "spill": hprof["spill"] / args.steps * 1e3,
                                 "note": "wall time of the calling thread inside vdl2gpu_push and inside vdl2gpu_poll_ready (record read-back). "
                                         "in_push = enqueue (the step's launches and event operations: what the thread must do) + wait_for_ring "
                                         "(blocked until the GPU has finished the push three back, whose ring, tables and planes this push reuses: "
                                         "the GPU is the slower side) + wait_input + spill, from vdl2gpu_get_host_profile(); the step is host-bound "
                                         "only if `enqueue` + in_poll_ready approach ms_per_step"},
        }
        if gathered is not None:
            out["gather"] = gathered
        if args.frames:
            out["frames"] = {"collected": nframes[0], "note": "k4_frames ran on every push's records (VDL2GPU_F_FRAMES); "
                             "frames are collected like the bursts: what is ready after every push, everything before the clock stops"}
        if host_ring is not None:
            out["host_ring"] = host_ring
        if host_ring_cu8 is not None:
            out["host_ring_cu8"] = host_ring_cu8
        if os.environ.get("VDL2GPU_DEBUG_COUNTERS") or os.environ.get("VDL2GPU_K1_PROF"):
            out["dbg"] = rx.debug_counters(64)
        if not args.no_cpu and world == 1:
            port = cpu_baseline(tiles_np[0], args.fmt, fos, rate)
            ref = cpu_baseline_reference(tiles_np[0], args.fmt, fos, rate)
            if ref is not None:
                ref["port"] = {k: port[k] for k in ("value", "unit", "cores", "sample")}   # the oracle, for comparison
                out["cpu_baseline"] = ref
            else:
                out["cpu_baseline"] = port
        if not parity_ok:
            print("bench.py: PARITY FAILED -- value withheld", file=sys.stderr)
            rc = 1
    rx.close()
    leg4 = None
    if world > 1 and args.config == 2 and not args.no_extra and not (args.rate or args.streams):
        # configs[3] across ALL ranks: 8 streams x 8 channels per rank (64 streams = 512 channels on 8 GPUs), every stream its own
        # recordings, the timed region fenced across the ranks, results collected as SURVEY 8e says.  (The headline above stays the
        # config-2 workload per GPU, so that an N = 1 line and BENCH agree.)
        del dbufs
        torch.cuda.empty_cache()
        from vdlm2dec_amd import synth as _sy
        lt = args.tiles or 16
        leg = run_leg("config4_512ch", workload=CONFIGS[4]["workload"], local=local, rate=2_000_000, fos=_sy.DEFAULT_FO_8CH, fmt="cs16", nstr=8,
                      ntiles=lt, bursts_per_s=4.0, steps=min(6, args.steps), warmup=min(3, max(1, args.warmup)), seed0=1234, check_streams=2,
                      stream_base=8 * rank, fence=fence, want_records=True, repeats=1)
        recs4, dt4 = leg.pop("_records"), leg.pop("_dt")
        t = torch.tensor([dt4], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        okt = torch.tensor([1 if leg["parity"]["equal"] else 0], device=cdev, dtype=torch.int32)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        g4 = shard.run_sharded(8 * world, lambda idx: recs4)
        if rank == 0:
            all_ok = bool(okt.item())
            leg4 = dict(leg)
            leg4.update({"workload": CONFIGS[4]["workload"] + (f" -- here {world} of the 8 GPUs: {8 * world} streams, {64 * world} channels" if world != 8 else ""),
                         "value": (8 * world * lt * TILE * leg["steps"] / float(t.item()) / 1e6) if all_ok else None, "unit": "MS/s",
                         "ms_per_step": float(t.item()) / leg["steps"] * 1e3, "n_gpus": world, "streams": 8 * world, "channels": 64 * world,
                         "samples_per_step": 8 * world * lt * TILE, "rank0": {k: leg[k] for k in ("first_push_ms", "max_push_ms", "repairs", "serial_redos")},
                         "gather": {"records_on_rank0": int(len(g4[0])), "per_rank": g4[1], "digest": shard.digest(g4[0]).hex()[:16],
                                    "digest_of_rank_digests": g4.combined.hex()[:16], "backend": args.backend, "ranks_share_one_gpu": bool(args.share_gpu)},
                         "parity": {**leg["parity"], "equal": all_ok, "what": "every rank checks two of its eight streams burst for burst against the oracle; verdicts MIN-reduced"}})
            if args.gather_out:
                np.save(args.gather_out.replace(".npy", "") + "_config4.npy", g4[0])
    if rank == 0:
        if leg4 is not None:
            out.setdefault("configs", {})["config4_512ch"] = leg4
            if leg4["value"] is None:
                rc = 1
        if world == 1 and not args.no_extra and args.config == 2 and not (args.rate or args.streams or args.tiles):
            del dbufs
            torch.cuda.empty_cache()
            out["configs"] = extra_legs(local)
            if isinstance(out.get("cpu_baseline"), dict) and out["cpu_baseline"].get("dropin_replay") is not None:
                dr = out["cpu_baseline"].pop("dropin_replay")
                if isinstance(dr, dict) and dr.get("value"):
                    dr["over_cpu_reference"] = dr["value"] / out["cpu_baseline"]["value"]
                out["configs"]["dropin_replay"] = dr
            if any(isinstance(v, dict) and (v.get("parity") or {}).get("equal") is False for v in out["configs"].values()):
                print("bench.py: a `configs` leg differs from the oracle -- its value is withheld", file=sys.stderr)
                rc = 1
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
