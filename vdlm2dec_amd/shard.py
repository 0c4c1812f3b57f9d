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


def run_sharded(nstreams: int, decode, dst: int = 0, group=None):
    """The N > 1 path: every rank decodes its own contiguous block of the `nstreams` wideband
    streams -- no data-path collective, streams share nothing -- and the results are collected on `dst`.

    decode(stream_indices: range) -> packed records (REC_DTYPE, `stream` = GLOBAL stream index) of those streams;
    on the GPU box that is a vdlm2dec_amd.demod.Receiver over the rank's device (bench.py, tests/test_gpu_multi.py),
    in the CPU tests the oracle stands in (tests/test_dist_cpu.py): the sharding and the gather are the same code.
    Returns (records on dst / empty elsewhere, per-rank counts, this rank's stream range)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = shard_streams(nstreams, rank, world)
    recs = decode(mine)
    allrecs, counts = gather_bursts(recs, dst, group)
    return allrecs, counts, mine


def _device(group=None) -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def gather_counts(n_local: int, group=None) -> List[int]:
    """all_gather of one integer per rank (burst count, frame count, ...)."""
    dev = _device(group)
    mine = torch.tensor([n_local], dtype=torch.int64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, mine, group=group)
    return [int(t.item()) for t in out]


def gather_bursts(recs: np.ndarray, dst: int = 0, group=None) -> Tuple[np.ndarray, List[int]]:
    """Variable-size gather of packed burst records to `dst` (padded all_gather; KBs of data).

    Returns (all records on dst / empty elsewhere, per-rank counts)."""
    counts = gather_counts(len(recs), group)
    dev = _device(group)
    width = REC_DTYPE.itemsize
    mx = max(max(counts), 1)
    buf = torch.zeros(mx * width, dtype=torch.uint8, device=dev)
    if len(recs):
        buf[:len(recs) * width] = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).to(dev)
    parts = [torch.zeros_like(buf) for _ in counts]
    dist.all_gather(parts, buf, group=group)
    if dist.get_rank(group) != dst:
        return np.zeros(0, REC_DTYPE), counts
    chunks = [p[:c * width].cpu().numpy().view(REC_DTYPE) for p, c in zip(parts, counts) if c]
    return (np.concatenate(chunks) if chunks else np.zeros(0, REC_DTYPE)), counts
