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
    """The scan kernels at six wavefronts per SIMD (80 registers).  Since round 5 a scan workgroup takes what passed its first screen
    through the sparse stages itself, behind its last tile (k2a_tail): those stages may park loop invariants in scratch (bounded:
    build() checks the size), the tile loop may not touch it (next test)."""
    import __graft_entry__ as g
    for k, regs, waves in (("_Z9k2a_probe8K2Params", 80, 6), ("_Z10k2a_verify8K2Params", 80, 6), ("_Z10k2a_region8K2Params", 96, 5)):
        r = resources[k]
        assert r["VGPRs"] <= regs and r["Occupancy [waves/SIMD]"] >= waves and r["ScratchSize [bytes/lane]"] <= g.SCAN_TAIL_SCRATCH, (k, r)
        assert r["LDS Size [bytes/block]"] * waves <= 160 * 1024, (k, r)
    assert not any("k2x_second" in k for k in resources)      # four launches a push until round 4


def test_scan_tile_loops_touch_no_scratch(tmp_path):
    """The ISA of the scan kernels: every scratch access lies behind the marker k2a_tail() plants where it reads the kernel's
    parameters again -- i.e. in the sparse stages a workgroup runs once, not in the tile loop it runs eight times."""
    import subprocess
    import __graft_entry__ as g
    asm = tmp_path / "vdl2gpu.s"
    flags = [f for f in g.HIPCC_FLAGS if f not in ("-shared", "-fPIC") and not f.startswith("-Wl,")]
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["-S", "--cuda-device-only", "-w",
                           os.path.join(g.CSRC, "vdl2gpu.hip"), "-o", str(asm)])
    text = asm.read_text().splitlines()
    for kern in ("_Z9k2a_probe8K2Params", "_Z10k2a_verify8K2Params", "_Z10k2a_region8K2Params"):
        start = text.index(next(ln for ln in text if ln.startswith(kern + ":")))
        end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
        body = text[start:end]
        marks = [i for i, ln in enumerate(body) if "kernarg again" in ln]
        scratch = [i for i, ln in enumerate(body) if ln.strip().startswith(("scratch_", "buffer_store", "buffer_load"))]
        assert marks, kern
        # the tile loop = from the filter pass's first packed multiply-add to the marker; what the compiler parks for the tail it may
        # park in front of the loop (once per workgroup) or in the straight-line code right in front of the marker
        first_fma = next(i for i, ln in enumerate(body) if "v_pk_fma_f32" in ln)
        assert first_fma < marks[0]
        assert not [i for i in scratch if first_fma <= i < marks[0] - 32], (kern, "scratch access inside the tile loop")
        # and the FIRST chunk of the sparse stages (all there is, as a rule) is straight-line code without scratch stores: the
        # loop over further chunks, where the compiler parks the chunk's invariants, begins behind it
        stores = [i for i in scratch if i > marks[0] and body[i].strip().startswith("scratch_store")]
        assert not stores or stores[0] - marks[0] > 1500, (kern, "the first chunk spills")
