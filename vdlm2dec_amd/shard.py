"""Multi-GPU: shard independent wideband streams across ranks.

The path has no exchange step: channels only share the read-only input and streams share nothing,
so every rank decodes its own contiguous block of streams and the only communication is result
collection -- per-rank counts/digests (a few bytes) and, if wanted, the packed burst records on rank 0.
On the GPU box the process group is "nccl" (= RCCL over xGMI); the CPU tests use "gloo".
"""
from __future__ import annotations

import hashlib
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

REC_DTYPE = np.dtype([("stream", "<i4"), ("chn", "<i4"), ("nbrow", "<i4"), ("nlbyte", "<i4"),
                      ("df_bits", "<u4"), ("pad", "<u4"), ("trig_dec", "<i8"), ("end_dec", "<i8"),
                      ("data", "u1", (8 * 255,))])


# vdl2gpu_burst_t (include/vdl2gpu.h) as a numpy record: what vdl2gpu_poll*() writes, read without a Python loop
BURST_DTYPE = np.dtype({"names": ["stream", "chn", "Fr", "nbrow", "nlbyte", "df", "ppm", "trig_dec", "end_dec",
                                  "trig_sample", "end_sample", "data"],
                        "formats": ["<i4", "<i4", "<i4", "<i4", "<i4", "<f4", "<f4", "<i8", "<i8", "<i8", "<i8", ("u1", (8 * 255,))],
                        "offsets": [0, 4, 8, 12, 16, 20, 24, 32, 40, 48, 56, 64], "itemsize": 2104})


def pack_records(raw: np.ndarray, stream_offset: int = 0) -> np.ndarray:
    """vdl2gpu_burst_t records (BURST_DTYPE) -> the packed form that travels between ranks."""
    out = np.zeros(len(raw), REC_DTYPE)
    out["stream"] = raw["stream"] + stream_offset
    for k in ("chn", "nbrow", "nlbyte", "trig_dec", "end_dec", "data"):
        out[k] = raw[k]
    out["df_bits"] = raw["df"].view(np.uint32)
    return out


def shard_streams(nstreams: int, rank: int, world: int) -> range:
    """Contiguous, balanced block of stream indices owned by `rank` (first ranks get the remainder)."""
    base, rem = divmod(nstreams, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def pack_bursts(bursts: Sequence, stream_offset: int = 0) -> np.ndarray:
    """Burst objects (demod.Burst or anything with the same fields) -> structured array."""
    out = np.zeros(len(bursts), REC_DTYPE)
    for i, b in enumerate(bursts):
        out[i] = (getattr(b, "stream", 0) + stream_offset, b.chn, b.nbrow, b.nlbyte,
                  int(np.float32(b.df).view(np.uint32)), 0, b.trig_dec, b.end_dec,
                  np.frombuffer(b.data, np.uint8))
    return out


def digest(recs: np.ndarray) -> bytes:
    """Order-independent digest of a set of burst records (for cross-rank / cross-run comparison)."""
    order = np.lexsort((recs["trig_dec"], recs["chn"], recs["stream"]))
    return hashlib.sha256(recs[order].tobytes()).digest()


class Gathered(tuple):
    """(records on dst / empty elsewhere, per-rank counts, this rank's stream range) -- what run_sharded() always
    returned -- plus what EVERY rank now knows: `digests` (one sha256 per rank, of its own records) and `combined`
    (sha256 of those in rank order: equal on every rank and reproducible from a single-process decode with
    combined_digest())."""
    digests: List[bytes]
    combined: bytes
    peak_extra_bytes: int


def combined_digest(recs: np.ndarray, nstreams: int, world: int) -> bytes:
    """What `Gathered.combined` of a `world`-rank run over `nstreams` streams must be, computed in one process."""
    h = hashlib.sha256()
    for r in range(world):
        mine = shard_streams(nstreams, r, world)
        sel = recs[(recs["stream"] >= mine.start) & (recs["stream"] < mine.stop)] if len(mine) else recs[:0]
        h.update(digest(sel))
    return h.digest()


def run_sharded(nstreams: int, decode, dst: int = 0, group=None, chunk_records: int = 4096):
    """The N > 1 path: every rank decodes its own contiguous block of the `nstreams` wideband
    streams -- no data-path collective, streams share nothing -- and the results are collected on `dst`.

    decode(stream_indices: range) -> packed records (REC_DTYPE, `stream` = GLOBAL stream index) of those streams;
    on the GPU box that is a vdlm2dec_amd.demod.Receiver over the rank's device (bench.py, tests/test_gpu_multi.py),
    in the CPU tests the oracle stands in (tests/test_dist_cpu.py): the sharding and the gather are the same code.
    Returns a `Gathered`: (records on dst / empty elsewhere, per-rank counts, this rank's stream range) with the
    per-rank digests and their combination as attributes."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = shard_streams(nstreams, rank, world)
    recs = decode(mine)
    allrecs, counts, digests, peak = gather_bursts(recs, dst, group, chunk_records)
    g = Gathered((allrecs, counts, mine))
    g.digests = digests
    h = hashlib.sha256()
    for d in digests:
        h.update(d)
    g.combined = h.digest()
    g.peak_extra_bytes = peak
    return g


def _device(group=None) -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def gather_counts(n_local: int, group=None) -> List[int]:
    """all_gather of one integer per rank (burst count, frame count, ...)."""
    dev = _device(group)
    mine = torch.tensor([n_local], dtype=torch.int64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, mine, group=group)
    return [int(t.item()) for t in out]


def gather_meta(recs: np.ndarray, group=None) -> Tuple[List[int], List[bytes]]:
    """What every rank learns about every other: its record count and the digest of its records -- one
    all_gather of 40 bytes per rank ("counts + digest everywhere")."""
    dev = _device(group)
    mine = np.zeros(40, np.uint8)
    mine[:8] = np.frombuffer(np.int64(len(recs)).tobytes(), np.uint8)
    mine[8:] = np.frombuffer(digest(recs), np.uint8)
    t = torch.from_numpy(mine).to(dev)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t, group=group)
    rows = [o.cpu().numpy() for o in out]
    return [int(np.frombuffer(r[:8].tobytes(), np.int64)[0]) for r in rows], [r[8:].tobytes() for r in rows]


def gather_bursts(recs: np.ndarray, dst: int = 0, group=None, chunk_records: int = 4096):
    """Variable-size gather of packed burst records to `dst`: counts and digests go to every rank (gather_meta),
    the RECORDS only to `dst`, point to point (send / recv: over RCCL that is one xGMI link per pair), in chunks of
    `chunk_records`, in rank order.  No padding and nothing replicated: a rank other than `dst` never holds more than
    one chunk of its own records beside them, `dst` its own records, the result and one chunk.  (Round 2's padded
    all_gather gave every rank world x max(count) records to use them on one.)

    Returns (all records on dst in rank order / empty elsewhere, per-rank counts, per-rank digests, peak extra bytes
    this rank allocated for the exchange)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    counts, digests = gather_meta(recs, group)
    dev = _device(group)
    width = REC_DTYPE.itemsize
    chunk_records = max(1, int(chunk_records))
    peak = 0
    if rank != dst:
        flat = recs.view(np.uint8).reshape(-1)
        for lo in range(0, len(recs), chunk_records):
            hi = min(len(recs), lo + chunk_records)
            t = torch.from_numpy(flat[lo * width:hi * width]).to(dev)    # (a view of the caller's array on CPU, one chunk in HBM under RCCL)
            peak = max(peak, 0 if dev.type == "cpu" else t.numel())
            dist.send(t, dst, group=group)
        return np.zeros(0, REC_DTYPE), counts, digests, peak
    total = sum(counts)
    out = np.zeros(total, REC_DTYPE)
    peak = out.nbytes
    flat = out.view(np.uint8).reshape(-1)
    at = 0
    buf = None
    for src in range(world):
        n = counts[src]
        if src == dst:
            out[at:at + n] = recs
            at += n
            continue
        for lo in range(0, n, chunk_records):
            m = min(chunk_records, n - lo)
            if dev.type == "cpu":
                t = torch.from_numpy(flat[(at + lo) * width:(at + lo + m) * width])    # received in place
                dist.recv(t, src, group=group)
            else:
                if buf is None:
                    buf = torch.empty(chunk_records * width, dtype=torch.uint8, device=dev)
                    peak += buf.numel()
                dist.recv(buf[:m * width], src, group=group)
                flat[(at + lo) * width:(at + lo + m) * width] = buf[:m * width].cpu().numpy()
        at += n
    return out, counts, digests, peak
