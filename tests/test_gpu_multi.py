"""N > 1 on real GPUs.  The driver's box has one GPU: the two-device test skips there, the launcher test
does not (it checks that `bench.py --gpus N` refuses to run N ranks on fewer than N devices instead of silently
running one)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import scenarios as S
from vdlm2dec_amd import shard, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_more_ranks_than_gpus(built):
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "GPU(s) visible" in (r.stderr + r.stdout)


def test_two_receivers_on_two_devices_equal_one(built, oracle):
    """Streams sharded over two devices (one Receiver each, vdlm2dec_amd.shard partition) decode to the same records as
    all streams on one device and as the oracle."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from vdlm2dec_amd.demod import Receiver, plan_channels
    nstreams = 3
    specs = [S.eight_channels(seed=800 + i, dur=0.05, info=(3, 9, 5, 12, 7, 4, 6, 8)) for i in range(nstreams)]
    raws = [synth.synth_stream(sp, "cs16") for sp in specs]
    fos = specs[0].fo

    def decode(dev, idx):
        if not len(idx):
            return np.zeros(0, shard.REC_DTYPE)
        with Receiver(specs[0].rate, [plan_channels(S.FC, fos)] * len(idx), fmt="cs16", max_push=specs[0].nsamples, device=dev) as rx:
            rx.push(np.stack([raws[i] for i in idx]))
            return shard.pack_bursts(rx.poll(), stream_offset=idx.start)

    parts = [decode(r, shard.shard_streams(nstreams, r, 2)) for r in range(2)]
    both = np.concatenate(parts)
    one = decode(0, range(nstreams))
    assert shard.digest(both) == shard.digest(one)
    nwant = sum(len(oracle.run_oracle(raws[s], "cs16", specs[s].rate, fos, S.FC)) for s in range(nstreams))
    assert len(both) == nwant >= 6
