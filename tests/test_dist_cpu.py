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


def _worker(rank, world, port, q, chunk):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    specs, raws = _streams()
    # the decoder is injected: here the oracle, on the GPU box a Receiver (bench.py) -- sharding + gather are shared
    own = {}

    def decode(idx):
        own["recs"] = shard.pack_bursts(_decode(O, specs, raws, idx))
        return own["recs"]

    g = shard.run_sharded(NSTREAMS, decode, dst=0, chunk_records=chunk)
    allrecs, counts, mine = g
    assert list(mine) == list(shard.shard_streams(NSTREAMS, rank, world))
    if rank != 0:
        # a rank that is not the destination gets no records back and allocates nothing beyond its own for the exchange
        assert len(allrecs) == 0 and g.peak_extra_bytes == 0
    assert g.digests[rank] == shard.digest(own["recs"]) and counts[rank] == len(own["recs"])
    dist.barrier()
    q.put((rank, counts, g.combined, [d.hex() for d in g.digests], shard.digest(allrecs) if rank == 0 else None, len(allrecs)))
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


@pytest.mark.parametrize("world,chunk", [(2, 4096), (3, 5), (2, 1)])
def test_gloo_gather_equals_single_process(oracle, world, chunk):
    """world-size 2 and 3 over gloo: counts and digests reach EVERY rank (40 bytes each), the records only rank 0,
    point to point and in chunks (chunk = 5 and 1 records: several sends per rank); the gathered set, its digest and
    the digest of digests equal the single-process decode."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, chunk)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    specs, raws = _streams()
    ref = _decode(oracle, specs, raws, range(NSTREAMS))
    recs = shard.pack_bursts(ref)
    want_counts = [sum(1 for b in ref if b.stream in shard.shard_streams(NSTREAMS, r, world)) for r in range(world)]
    want_combined = shard.combined_digest(recs, NSTREAMS, world)
    for rank, counts, combined, digests, dig0, n in res:
        assert counts == want_counts and combined == want_combined      # every rank, not only the destination
        assert digests == res[0][3]
        if rank == 0:
            assert n == len(ref) and n >= 6 and dig0 == shard.digest(recs)
        else:
            assert n == 0 and dig0 is None
