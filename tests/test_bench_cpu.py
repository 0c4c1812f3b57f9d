"""bench.py's CPU legs (no GPU): the reference-kind baseline runs the prebuilt real reference, the port-kind one the
oracle; both must see the same traffic."""
import os
import sys

import numpy as np
import pytest

import scenarios as S
from vdlm2dec_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def tile():
    spec = synth.random_scenario(2_000_000, S.FO8[:3], 1_050_000, seed=3, bursts_per_s=8.0, info_max=120)
    return spec, synth.synth_stream(spec, "cs16")


def test_port_baseline(oracle, tile):
    import bench
    spec, raw = tile
    r = bench.cpu_baseline(raw, "cs16", spec.fo, spec.rate, budget_s=0.5)
    assert r["kind"] == "port" and r["value"] > 1.0 and r["cores"] == 3 and r["unit"] == "MS/s"


def test_reference_baseline(oracle, tile):
    import bench
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_rtl")):
        pytest.skip("the real reference is built only where /root/reference exists")
    spec, raw = tile
    r = bench.cpu_baseline_reference(raw, "cs16", spec.fo, spec.rate, budget_s=0.5)
    assert r is not None and r["kind"] == "reference" and r["value"] > 1.0 and r["cores"] == 4
    assert "oracle/_ref/ref_rtl" in r["sample"]
    assert bench.cpu_baseline_reference(raw, "nosuchformat", spec.fo, spec.rate) is None


def test_share_gpu_needs_gloo():
    """`--share-gpu` (two ranks on cuda:0: the N > 1 path on a one-GPU box) is only accepted with `--backend gloo`: RCCL
    refuses two ranks on one device, and the refusal must come from bench.py's argument check, not from a hang in init."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--backend gloo" in (r.stderr + r.stdout)


def test_dropin_replay_without_a_gpu_reports_an_error_not_a_number(tile, tmp_path):
    """bench.py's `dropin_replay` leg runs oracle/_ref/ref_rtl_gpu; where there is no GPU the shim's vdl2gpu_create() fails
    (there is no CPU fallback) and the leg must say so instead of producing a figure."""
    import bench
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_rtl_gpu")):
        pytest.skip("the drop-in executable is built only where /root/reference exists")
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    spec, raw = tile
    path = str(tmp_path / "iq.bin")
    raw.tofile(path)
    fo = ",".join(str(int(x)) for x in spec.fo)
    fr = ",".join(str(int(bench.FC + x)) for x in spec.fo)
    r = bench._dropin_replay(ROOT, str(tmp_path), path, "cs16", spec.rate, fo, fr, raw.size // 2)
    assert r is None or ("error" in r and "value" not in r)
