#!/usr/bin/env python3
"""bench.py -- throughput of the IQ->bursts hot path on MI355X (BASELINE.json metric).

Workload (configs[1] of BASELINE.json): one wideband stream, 8 VDL2 channels on the 25 kHz
grid, SDRINRATE 2 MS/s, cs16 interleaved IQ, synthetic D8PSK bursts (Poisson arrivals per
channel, 8..60 LSB amplitude, +-400 Hz carrier offset, AWGN) from vdlm2dec_amd.synth.
A "step" is one vdl2gpu_push() of `--batch` samples (default 16 x 4.2 MS = 67.2 MS = 33.6 s
of air time) that is already resident in HBM, plus vdl2gpu_poll() of the decoded bursts.
`value` = input samples consumed per second with all 8 channels demodulated, whole job.

N > 1 (weak scaling): the path shards by independent wideband stream, so every
rank decodes its own stream of the same size; there is no data-path collective.  RCCL is used
only for the timing barrier / max-reduce and a gather of per-rank burst counts.

Extra objects on the JSON line:
  roofline      channeliser kernel (the only kernel that touches the full-rate stream):
                algorithmic bytes = 4 B per cs16 sample, read once for all 8 channels
                (SURVEY.md 8d), divided by its mean launch time from HIP events recorded on
                the library's stream (vdl2gpu_get_timing).
  kernels_ms    mean per-step device time of each kernel, same events.
  cpu_baseline  the oracle (CPU restatement, verified bit-equal to the reference) timed on
                this host on a bounded sample of the same recording, one thread per channel.
  parity        GPU bursts of the first tile compared with the oracle's, msgblk_t level.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

RATE = 2_000_000          # default; --rate overrides (10 MS/s = config 3)
FC = 136_975_000
TILE = 4_200_000          # samples per generated tile (multiple of the 2000-sample LO/decimator period)
HBM_PEAK_GBS = 8000.0     # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def make_tile(seed: int, fmt: str, rate: int = RATE, fos=None):
    from vdlm2dec_amd import synth
    fos = fos or synth.DEFAULT_FO_8CH
    spec = synth.random_scenario(rate, fos, TILE, seed=seed, bursts_per_s=4.0 * rate / RATE, info_max=240)
    return spec, synth.synth_stream(spec, fmt)


def cpu_baseline(raw: np.ndarray, fmt: str, fos, budget_s: float = 12.0, rate: int = RATE):
    """Oracle ("port" of the reference path) on host cores: one thread per channel, like the
    reference's one rcv_thread per channel (main.c:228-231)."""
    from oracle import oracle as O
    O.lib()
    per = O.PER_SAMPLE[fmt]
    n_total = raw.size // per
    # size the sample so the run takes roughly budget_s: probe one channel on 1 MS first
    probe = min(n_total, 1_000_000)
    ch = O.OracleChannel(rate, fos[0], FC + fos[0])
    t0 = time.perf_counter()
    ch.feed(raw[:probe * per], fmt)
    one = (time.perf_counter() - t0) / probe
    ch.close()
    ncores = os.cpu_count() or 1
    nthreads = min(len(fos), ncores)
    passes = int(np.ceil(len(fos) / nthreads))
    # the recording is one tile; feed it repeatedly (the stream simply continues) until ~budget_s
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
            "sample": f"{n} samples ({reps} x the 4.2 MS tile of the same recording), 8 channels, {nthreads} threads "
                      f"(1 thread/channel), {dt:.1f} s wall, {nb} bursts; single-thread single-channel "
                      f"{1.0 / one / 1e6:.1f} MS/s; host CPU: {model} x{ncores}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--tiles", type=int, default=16, help="tiles of 4.2 MS per step (batch = tiles*4.2 MS)")
    ap.add_argument("--fmt", default="cs16", choices=["cs16", "cu8"])
    ap.add_argument("--rate", type=int, default=RATE, help="SDRINRATE (config 3: 10000000)")
    ap.add_argument("--streams", type=int, default=1, help="independent wideband streams per GPU (config 4: 8)")
    ap.add_argument("--frames", action="store_true",
                    help="also run the block path (RS, HDLC, FCS: SURVEY 8f-1) on every push's bursts and collect the frames")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-ring", action="store_true", help="skip the PCIe-inclusive extra pass (ingest ring from pinned host memory)")
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from vdlm2dec_amd import synth
    from vdlm2dec_amd.demod import Receiver, plan_channels

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    dev = torch.device("cuda", local)

    rate = args.rate
    fos = synth.DEFAULT_FO_8CH if rate == RATE else tuple(int(f * rate / RATE) // 25000 * 25000 for f in synth.DEFAULT_FO_8CH)
    nstr = args.streams
    tiles_np = []
    for st_i in range(nstr):
        spec, tile = make_tile(seed=1234 + rank * 64 + st_i, fmt=args.fmt, rate=rate, fos=fos)
        tiles_np.append(tile)
    tile = tiles_np[0]
    batch = args.tiles * TILE
    dbatch = torch.stack([torch.from_numpy(t).to(dev).repeat(args.tiles) for t in tiles_np]).contiguous()  # [streams, batch*2]
    sample_bytes = 4 if args.fmt == "cs16" else 2
    stride_bytes = dbatch.stride(0) * dbatch.element_size()

    rx = Receiver(rate, [plan_channels(FC, fos)] * nstr, fmt=args.fmt, max_push=batch, device=local, max_bursts=1 << 18,
                  frames=args.frames)
    first = []
    nbursts = 0

    from vdlm2dec_amd import lib as _lib
    from vdlm2dec_amd.demod import Burst
    rawbuf = (_lib.BurstT * 16384)()
    framebuf = (_lib.FrameT * 4096)() if args.frames else None
    nframes = [0]

    def drain(collect, ready_only):
        nonlocal nbursts
        while True:
            n = rx.poll_ready_raw(rawbuf, 16384) if ready_only else rx.poll_raw(rawbuf, 16384)
            nbursts += n
            if collect is not None:
                for i in range(n):
                    b = rawbuf[i]
                    collect.append(Burst(b.stream, b.chn, b.Fr, b.nbrow, b.nlbyte, b.df, b.ppm, b.trig_dec,
                                         b.end_dec, b.trig_sample, b.end_sample, bytes(b.data)))
            if n < 16384:
                break
        if args.frames:
            while True:
                m = (rx.L.vdl2gpu_poll_frames_ready if ready_only else rx.L.vdl2gpu_poll_frames)(rx.h, framebuf, 4096)
                if m < 0:
                    raise RuntimeError("vdl2gpu_poll_frames failed")
                nframes[0] += m
                if m < 4096:
                    break

    def step(collect=None, pipelined=False):
        # one hand-off of resident samples + delivery of decoded msgblk records to the host.
        # pipelined: take what earlier pushes have finished (vdl2gpu_poll_ready) while this push
        # runs; everything is drained inside the timed region after the last step.
        rx.push_device(dbatch.data_ptr(), batch, stride_bytes)
        drain(collect, ready_only=pipelined)

    for i in range(args.warmup):
        step(first if i == 0 else None)
    if args.warmup == 0:
        pass
    rx.timing(reset=True)
    nbursts = 0

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(first if (args.warmup == 0 and i == 0) else None, pipelined=not (args.warmup == 0 and i == 0))
    drain(None, ready_only=False)       # every burst of every step is on the host before the clock stops
    rx.sync()
    fence()
    dt = time.perf_counter() - t0
    nbursts_timed = nbursts             # the extra passes below deliver bursts too
    tm = rx.timing(reset=True)
    st = rx.stats()
    # outside the timed region: the same hand-off a few times with nothing else on the GPU (each push
    # finished before the next starts), to tell what the channeliser does alone from what it does
    # while it shares the GPU with the previous push's demodulator (the timed region above)
    for _ in range(4):
        rx.push_device(dbatch.data_ptr(), batch, stride_bytes)
        rx.sync()
        drain(None, ready_only=False)
    tm_iso = rx.timing(reset=True)
    # also outside the timed region: the same hand-off from page-locked host memory through the ingest
    # ring -- the PCIe-inclusive rate.  Reported beside `value`, never as `value`.
    host_ring = None
    if rank == 0 and world == 1 and not args.no_ring:
        try:
            hb = dbatch.cpu().numpy().view(np.uint8).reshape(args.streams, -1)
            rx.ring_init(batch, nslots=3)
            nb = batch * rx.sample_bytes
            for k in range(3):          # fill the three slots once: the producer's work is not what is measured
                slot = rx.ring_acquire()
                slot[:, :nb] = hb[:, :nb]
                rx.ring_commit(batch)
            drain(None, ready_only=False)
            rx.sync()
            th = time.perf_counter()
            for k in range(6):
                rx.ring_acquire()
                rx.ring_commit(batch)
                drain(None, ready_only=True)
            drain(None, ready_only=False)
            rx.sync()
            dth = time.perf_counter() - th
            host_ring = {"value": 6 * batch * args.streams / dth / 1e6, "unit": "MS/s", "pushes": 6,
                         "note": "samples start in page-locked host memory (vdl2gpu_ring_acquire/commit): H2D copy "
                                 "on its own stream beside the previous push's kernels, bursts delivered to the host"}
            rx.timing(reset=True)
        except Exception as e:      # the extra measurement must not cost the run its line
            host_ring = {"error": str(e)}
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        cnt = torch.tensor([nbursts_timed], device=dev, dtype=torch.int64)
        allc = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(allc, cnt)
        total_bursts = int(sum(int(c.item()) for c in allc))
    else:
        total_bursts = nbursts_timed

    parity = None
    if rank == 0 and not args.no_parity:
        from oracle import oracle as O     # checker only
        want = sorted(b.key() for b in O.run_oracle(tile, args.fmt, rate, fos, FC))
        got = sorted((b.chn, b.nbrow, b.nlbyte, b.data) for b in first if b.end_sample < TILE and b.stream == 0)
        # oracle bursts still open at the tile end have no counterpart; both lists hold completed ones
        parity = {"level": "msgblk_t (pre-RS) bit-exact, first tile", "oracle_bursts": len(want),
                  "gpu_bursts": len(got), "equal": want == got}

    if rank == 0:
        k1_ms = tm["channelise_ms"] / max(1, tm["pushes"])
        k2_ms = tm["demod_ms"] / max(1, tm["pushes"])
        k2a_ms = tm["scan_ms"] / max(1, tm["pushes"])
        k2b_ms = tm["cluster_ms"] / max(1, tm["pushes"])
        k2c_ms = tm["resolve_ms"] / max(1, tm["pushes"])
        k3_ms = tm["other_ms"] / max(1, tm["pushes"])
        # dominant full-rate kernel: k1_fast (all whole 1 ms periods but the first and last of a push)
        # (a push may split it into two launches that run beside different stretches of the previous push's
        #  demodulator: bytes and time are per launch, averaged over all launches, like rocprof's average)
        fast_ms = tm["channelise_fast_ms"] / max(1, tm["fast_pushes"])
        fast_samples = (batch * 21 // 500 // 84 - 2) * 2000 * nstr
        alg_bytes = float(fast_samples) * sample_bytes * tm["pushes"] / max(1, tm["fast_pushes"])
        achieved = alg_bytes / (fast_ms * 1e-3) / 1e9 if fast_ms > 0 else 0.0
        value = world * nstr * batch * args.steps / dt / 1e6
        # HBM traffic of the same kernel from the committed PMC passes of this very command
        # (profiles/r01_bench_pmc_hbm.json: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc runs;
        #  FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction, WRITE_SIZE as reported)
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_pmc_hbm.json")))
            if args.tiles == 16 and args.fmt == "cs16" and nstr == 1 and rate == RATE:
                fk = [k for k in pm["FETCH_SIZE_KB_per_launch"] if "k1_fast" in k][0]
                traffic = (2.0 * pm["FETCH_SIZE_KB_per_launch"][fk] + pm["WRITE_SIZE_KB_per_launch"][fk]) * 1024.0
        except (OSError, KeyError, ValueError, IndexError):
            pass
        out = {
            "metric": "IQ MS/s demodulated (8 ch, 2 MS/s cs16) + CRC-pass frame parity vs ref",
            "value": value, "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("configs[1]: 8 channels @ 2 MS/s on 1xMI355X, synthetic D8PSK bursts" if (rate == RATE and nstr == 1)
                                    else f"non-default: {nstr} stream(s) x 8 channels @ {rate / 1e6:g} MS/s"),
                       "fmt": args.fmt, "samples_per_step": batch, "air_time_s_per_step": batch / RATE,
                       "channels": 8, "streams_per_gpu": nstr, "sdrinrate": rate,
                       "bursts_per_step": total_bursts / max(1, args.steps * world),
                       "x_real_time": value * 1e6 / rate, "parallelism": f"stream-sharded x{world}"},
            "roofline": {"bound": "hbm", "kernel": "k1_fast", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": fast_ms,
                         "alone": (lambda ms, by: {"avg_launch_ms": ms, "achieved": by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                                                   "frac": (by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms > 0 else 0.0,
                                                   "how": "4 pushes after the timed region, each synchronised before the next: "
                                                          "no other kernel on the GPU"})(
                             tm_iso["channelise_fast_ms"] / max(1, tm_iso["fast_pushes"]),
                             float(fast_samples) * sample_bytes * tm_iso["pushes"] / max(1, tm_iso["fast_pushes"])),
                         "note": "live: HIP events around the k1_fast launches inside the timed region, where they run beside the "
                                 "previous push's demodulator kernels on a second stream (two launches per push); "
                                 "algorithmic bytes = 4 B per cs16 input sample, read once for all 8 channels; the "
                                 "kernel also writes the 84 kS/s planes (2.7 B per input sample), which is "
                                 "intermediate traffic, not algorithmic (SURVEY.md 8d)"},
            "kernels_ms": {"k1_channelise": k1_ms, "k2a_scan": k2a_ms, "k2b_clusters": k2b_ms,
                           "k2c_resolve+k2d_gather": k2c_ms, "k3_compact": k3_ms},
            "whole_path_GBps": alg_bytes / ((k1_ms + k2_ms + k3_ms) * 1e-3) / 1e9,
            "stats": {k: st[k] for k in ("sync_evals", "triggers", "header_rejects", "bursts", "deferrals",
                                           "candidates", "serial_redos", "serial_samples", "overflowed")},
            "parity": parity,
        }
        if args.frames:
            out["frames"] = {"collected": nframes[0], "note": "k4_frames ran on every push's records (VDL2GPU_F_FRAMES); "
                             "frames are collected like the bursts: what is ready after every push, everything before the clock stops"}
        if host_ring is not None:
            out["host_ring"] = host_ring
        if os.environ.get("VDL2GPU_DEBUG_COUNTERS") or os.environ.get("VDL2GPU_K1_PROF"):
            out["dbg"] = rx.debug_counters(64)
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(tile, args.fmt, fos, rate=rate)
        print(json.dumps(out))
    rx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
