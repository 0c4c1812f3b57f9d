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
