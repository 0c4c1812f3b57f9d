"""What the compiler made of the channeliser: its speed hangs on register counts (wavefronts per SIMD) and on not
spilling (scratch traffic breaks the counted waits' budget), neither of which a parity test notices."""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def resources():
    import __graft_entry__ as g
    g.build_hip()
    out = {}
    for ln in open(g.RES):
        if ln.startswith("#") or not ln.strip():
            continue
        name, rest = ln.split(" ", 1)
        # field names contain spaces: split on the '=' boundaries
        parts = rest.strip().split("=")
        keys = [parts[0]]
        vals = []
        for mid in parts[1:-1]:
            v, k = mid.split(" ", 1)
            vals.append(int(v))
            keys.append(k)
        vals.append(int(parts[-1]))
        out[name] = dict(zip(keys, vals))
    return out


def test_every_kernel_is_listed(resources):
    names = " ".join(resources)
    for k in ("k1_fast", "k1_pp", "k1_channelise", "k2a_probe", "k2a_region", "k2a_verify", "k2s_sort", "k2s_merge", "k2b_clusters",
              "k2c_resolve", "k2f_commit", "k2d_payload", "k3_carry", "k3_rebase", "k_export_records", "k4_frames"):
        assert k in names, k


@pytest.mark.parametrize("fmt,vgprs,waves", [(0, 96, 5), (1, 96, 5), (2, 128, 4), (3, 96, 5)])
def test_k1_fast_registers(resources, fmt, vgprs, waves):
    r = resources[f"_Z7k1_fastILi{fmt}EEv8K1Params"]
    assert r["VGPRs Spill"] == 0 and r["ScratchSize [bytes/lane]"] == 0
    assert r["VGPRs"] <= vgprs and r["Occupancy [waves/SIMD]"] >= waves
    assert r["LDS Size [bytes/block]"] * 2 * waves <= 160 * 1024      # two wavefronts per workgroup


def test_no_channeliser_kernel_uses_scratch(resources):
    for k, r in resources.items():
        if "k1_" in k:
            assert r["VGPRs Spill"] == 0 and r["ScratchSize [bytes/lane]"] == 0, k


def test_the_critical_kernels_touch_no_scratch(resources):
    """the cluster kernel spilled 12 registers (72 bytes of scratch per lane), the resolver passed the serial
    machine's context through 264 bytes of scratch; build() refuses either now, and this is what it reads."""
    for k, r in resources.items():
        if any(n in k for n in ("k2b_clusters", "k2c_resolve", "k2f_commit")):
            assert r["VGPRs Spill"] == 0 and r["ScratchSize [bytes/lane]"] == 0, (k, r)


def test_scan_kernels_keep_their_wavefronts(resources):
    """Round 4: the scan kernels at six wavefronts per SIMD (80 registers, no scratch); the sparse stages at five -- 96 registers, of
    which the compiler parks three around the exact stage's batch loop (16 bytes: build() allows exactly that): a latency-bound
    kernel's cost to the step is the register space its waiting wavefronts hold."""
    for k in ("_Z9k2a_probe8K2Params", "_Z10k2a_verify8K2Params"):
        r = resources[k]
        assert r["VGPRs"] <= 80 and r["Occupancy [waves/SIMD]"] >= 6 and r["ScratchSize [bytes/lane]"] == 0, (k, r)
        assert r["LDS Size [bytes/block]"] * 6 <= 160 * 1024, (k, r)
    r = resources["_Z10k2x_second8K2Params"]
    assert r["VGPRs"] <= 96 and r["Occupancy [waves/SIMD]"] >= 5 and r["ScratchSize [bytes/lane]"] <= 16, r
