"""End-to-end drop-in: the reference's UNCHANGED host path (vdlm2.c, crc.c, rs.c, compiled from the
reference's own sources into oracle/_ref/ref_rtl_gpu in the build container) linked with
dropin/vdl2gpu_rcv.c + libvdl2gpu.so instead of d8psk.c/viterbi.c.  Its msgblk_t hand-offs
(decodeVdlm2) and CRC-clean frames (out) must equal what the all-CPU reference produced for the
same recordings (tests/golden)."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_rtl_gpu")


@pytest.mark.parametrize("name", ["regimes_cu8_2ms", "eight_cs16_2ms", "cs16_10ms", "short_cf32_2ms"])
def test_reference_host_path_with_gpu_front_end(built, tmp_path, name):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/ref_rtl_gpu not built (needs /root/reference at build time)")
    meta = json.load(open(os.path.join(HERE, "golden", name + ".json")))
    raw = np.load(os.path.join(HERE, "golden", meta["iq"] + ".npz"))["raw"]
    iq = str(tmp_path / "iq.raw")
    raw.tofile(iq)
    out = str(tmp_path / "out.txt")
    fos = ",".join(str(f) for f in meta["fo"])
    frs = ",".join(str(meta["fc"] + f) for f in meta["fo"])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    subprocess.run([EXE, iq, meta["fmt"], str(meta["rate"]), fos, frs, out, "0", ""], check=True, env=env, timeout=300)
    multi = len(meta["fo"]) > 1
    blocks, frames = {}, {}
    for line in open(out):
        p = line.split()
        chn = int(p[4 if p[0] == "F" else 5][1:]) if multi else 0
        if p[0] == "B":
            blocks.setdefault(chn, []).append((int(p[1]), int(p[2]), int(p[4], 16), p[-1]))
        elif p[0] == "F":
            frames.setdefault(chn, []).append(p[-1])
    for c in meta["channels"]:
        want_b = [(b["nbrow"], b["nlbyte"], b["df_bits"], b["data"]) for b in c["blocks"]]
        got_b = [(a, b, d, x) for (a, b, d, x) in blocks.get(c["chn"], [])]
        assert got_b == want_b, (name, c["chn"])
        assert frames.get(c["chn"], []) == c["frames"], (name, c["chn"])


EXEF = os.path.join(ROOT, "oracle", "_ref", "ref_rtl_gpuf")


@pytest.mark.parametrize("name", ["regimes_cu8_2ms", "eight_cs16_2ms", "cs16_10ms", "short_cf32_2ms"])
def test_gpu_front_end_and_gpu_block_path_feed_out(built, tmp_path, name):
    """-DVDL2GPU_FRAMES build of the shim: no d8psk.c, viterbi.c, vdlm2.c, rs.c or crc.c at all --
    demodulator and block path both on the GPU, the harness's out() receives the frames.  They must be
    the frames the all-CPU reference passed to its out() for the same recording."""
    if not os.path.exists(EXEF):
        pytest.skip("oracle/_ref/ref_rtl_gpuf not built (needs /root/reference at build time)")
    meta = json.load(open(os.path.join(HERE, "golden", name + ".json")))
    raw = np.load(os.path.join(HERE, "golden", meta["iq"] + ".npz"))["raw"]
    iq = str(tmp_path / "iq.raw")
    raw.tofile(iq)
    out = str(tmp_path / "out.txt")
    fos = ",".join(str(f) for f in meta["fo"])
    frs = ",".join(str(meta["fc"] + f) for f in meta["fo"])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    subprocess.run([EXEF, iq, meta["fmt"], str(meta["rate"]), fos, frs, out, "0", ""], check=True, env=env, timeout=300)
    multi = len(meta["fo"]) > 1
    frames = {}
    for line in open(out):
        p = line.split()
        if p[0] == "F":
            frames.setdefault(int(p[4][1:]) if multi else 0, []).append((int(p[1]), int(p[2]), p[-1]))
    for c in meta["channels"]:
        assert [f for _, _, f in frames.get(c["chn"], [])] == c["frames"], (name, c["chn"])
