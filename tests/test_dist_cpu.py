"""N > 1 path on CPU: world-size-2 gloo.  Streams are sharded across ranks (no data-path collective),
each rank decodes its shard -- with the oracle standing in for the GPU decoder, which these CPU
tests cannot run -- and the results are gathered; the union must equal the single-process decode."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import scenarios as S
from vdlm2dec_amd import shard, synth

NSTREAMS = 3


def _streams():
    specs = [S.eight_channels(seed=700 + i, dur=0.05, info=(3, 9, 5, 12, 7, 4, 6, 8)) for i in range(NSTREAMS)]
    return specs, [synth.synth_stream(sp, "cs16") for sp in specs]


class _B:
    pass


def _wrap(b, stream):
    w = _B()
    w.stream = stream
    for k in ("chn", "nbrow", "nlbyte", "df", "trig_dec", "end_dec", "data"):
        setattr(w, k, getattr(b, k))
    return w


def _decode(O, specs, raws, idxs):
    out = []
    for s in idxs:
        out += [_wrap(b, s) for b in O.run_oracle(raws[s], "cs16", specs[s].rate, specs[s].fo[:3], S.FC)]
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    specs, raws = _streams()
    # the decoder is injected: here the oracle, on the GPU box a Receiver (bench.py) -- sharding + gather are shared
    allrecs, counts, mine = shard.run_sharded(NSTREAMS, lambda idx: shard.pack_bursts(_decode(O, specs, raws, idx)), dst=0)
    assert list(mine) == list(shard.shard_streams(NSTREAMS, rank, world))
    dist.barrier()
    if rank == 0:
        q.put((counts, shard.digest(allrecs), len(allrecs)))
    dist.destroy_process_group()


def test_packed_record_layout_matches_the_c_struct(built):
    """shard.BURST_DTYPE is vdl2gpu_burst_t byte for byte (ctypes mirror in vdlm2dec_amd.lib)."""
    import ctypes as C
    from vdlm2dec_amd import lib
    assert shard.BURST_DTYPE.itemsize == C.sizeof(lib.BurstT) == 2104
    for name, _ in lib.BurstT._fields_:
        assert shard.BURST_DTYPE.fields[name][1] == getattr(lib.BurstT, name).offset, name
    buf = (lib.BurstT * 3)()
    buf[1].stream, buf[1].chn, buf[1].nbrow, buf[1].nlbyte, buf[1].df, buf[1].trig_dec, buf[1].end_dec = 2, 5, 3, 77, -0.4, 123456789012, 123456789999
    buf[1].data[7][254] = 0xAB
    raw = np.frombuffer(buf, dtype=shard.BURST_DTYPE)
    r = shard.pack_records(raw, stream_offset=10)[1]
    assert (r["stream"], r["chn"], r["nbrow"], r["nlbyte"], r["trig_dec"], r["end_dec"]) == (12, 5, 3, 77, 123456789012, 123456789999)
    assert r["df_bits"] == np.float32(-0.4).view(np.uint32) and r["data"][8 * 255 - 1] == 0xAB


def test_shard_partition_is_exact():
    for n in (1, 2, 3, 8, 64, 65):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in shard.shard_streams(n, r, world)]
            assert got == list(range(n))
            sizes = [len(shard.shard_streams(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_gather_equals_single_process(oracle):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    counts, dig, n = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    specs, raws = _streams()
    ref = _decode(oracle, specs, raws, range(NSTREAMS))
    recs = shard.pack_bursts(ref)
    assert n == len(ref) and sum(counts) == n and n >= 6
    assert counts == [sum(1 for b in ref if b.stream in shard.shard_streams(NSTREAMS, r, 2)) for r in range(2)]
    assert dig == shard.digest(recs)
