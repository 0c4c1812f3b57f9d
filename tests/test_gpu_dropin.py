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
        chn = int(p[4 if p[0] == "F" else 6][1:]) if multi else 0
        if p[0] == "B":
            sec, usec = p[5][1:].split(".")         # msgblk_t.tv as the shim set it (the harness's clock counts samples)
            blocks.setdefault(chn, []).append((int(p[1]), int(p[2]), int(p[4], 16), p[-1], int(sec) * 1_000_000 + int(usec)))
        elif p[0] == "F":
            frames.setdefault(chn, []).append(p[-1])
    for c in meta["channels"]:
        want_b = [(b["nbrow"], b["nlbyte"], b["df_bits"], b["data"]) for b in c["blocks"]]
        got_b = [(a, b, d, x) for (a, b, d, x, _) in blocks.get(c["chn"], [])]
        assert got_b == want_b, (name, c["chn"])
        # d8psk.c:295: every block carries the time its sync trigger was seen at -- the all-CPU reference's stamp
        assert [t for (_, _, _, _, t) in blocks.get(c["chn"], [])] == [b["tv"] for b in c["blocks"]], (name, c["chn"])
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


def _fnv(data: bytes) -> int:
    h = 1469598103934665603
    for x in data:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_c_speed_caller_ring_and_frames_are_deterministic(built, oracle, tmp_path):
    """A C program drives the ingest ring (32768-sample blocks, 3 slots) with the block path in the pipeline
    and collects with the never-waiting calls -- nothing slows the caller down, so copies, kernels of several
    pushes and read-backs overlap as much as they ever will.  Ten runs must give the oracle's bursts and
    frames every time."""
    import scenarios as S
    from vdlm2dec_amd import synth
    from vdlm2dec_amd.lib import LIB_PATH
    exe = str(tmp_path / "ring_stress")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "ctests", "ring_stress.c"), LIB_PATH,
                           "-Wl,-rpath," + os.path.dirname(LIB_PATH)])
    spec = synth.random_scenario(2_000_000, S.FO8[:4], 3_000_000, seed=4711, bursts_per_s=30.0, info_max=200)
    raw = synth.synth_stream(spec, "cu8")
    iq = str(tmp_path / "iq.raw")
    raw.tofile(iq)
    ob = oracle.run_oracle(raw, "cu8", spec.rate, spec.fo, S.FC)
    want = sorted(["B %d %d %d %016x" % (b.chn, b.nbrow, b.nlbyte, _fnv(bytes(b.data[:b.nbrow * 255]))) for b in ob] +
                  ["F %d %d %016x" % (b.chn, len(f), _fnv(f)) for b in ob
                   for f in oracle.frames_of_block(b.nbrow, b.nlbyte, b.data)])
    assert len(want) >= 60
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    for rep in range(10):
        out = subprocess.run([exe, iq, "cu8", str(spec.rate), str(len(spec.fo))] + [str(f) for f in spec.fo] +
                             ["32768", "3"], capture_output=True, text=True, env=env, timeout=120)
        assert out.returncode == 0, out.stderr[-2000:]
        assert sorted(out.stdout.split("\n")[:-1]) == want, rep


def test_producer_and_consumer_threads_on_one_handle(built, oracle, tmp_path):
    """The reference's threading shape (include/vdl2gpu.h "Threads"): one pthread commits blocks through the ingest ring like
    the SDR library's callback (rtl.c:274-295, 302), another collects bursts meanwhile with vdl2gpu_poll_ready() and, every
    eighth turn, the waiting vdl2gpu_poll() -- which must not hold the producer up and must not lose or reorder anything.  Ten
    runs, every one the oracle's bursts, each channel's in time order."""
    import scenarios as S
    from vdlm2dec_amd import synth
    from vdlm2dec_amd.lib import LIB_PATH
    exe = str(tmp_path / "thread_stress")
    subprocess.check_call(["gcc", "-O2", "-pthread", "-I", os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "ctests", "thread_stress.c"), LIB_PATH,
                           "-Wl,-rpath," + os.path.dirname(LIB_PATH)])
    spec = synth.random_scenario(2_000_000, S.FO8[:4], 3_000_000, seed=815, bursts_per_s=30.0, info_max=200)
    raw = synth.synth_stream(spec, "cu8")
    iq = str(tmp_path / "iq.raw")
    raw.tofile(iq)
    ob = oracle.run_oracle(raw, "cu8", spec.rate, spec.fo, S.FC)
    want = sorted("B %d %d %d %016x" % (b.chn, b.nbrow, b.nlbyte, _fnv(bytes(b.data[:b.nbrow * 255]))) for b in ob)
    assert len(want) >= 40
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    for rep in range(10):
        blk, slots = (("32768", "3"), ("65536", "4"), ("8000", "8"))[rep % 3]
        out = subprocess.run([exe, iq, "cu8", str(spec.rate), str(len(spec.fo))] + [str(f) for f in spec.fo] + [blk, slots],
                             capture_output=True, text=True, env=env, timeout=120)
        assert out.returncode == 0, out.stderr[-2000:]
        assert sorted(out.stdout.split("\n")[:-1]) == want, rep
