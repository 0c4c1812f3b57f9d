"""The C-ABI shared library: loads, exports everything include/vdl2gpu.h declares, host-only helpers
agree with the oracle, and it refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_every_declared_symbol(built):
    from vdlm2dec_amd import lib
    L = lib.load()
    hdr = open(os.path.join(ROOT, "include", "vdl2gpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vdl2gpu_[a-z0-9_]+|reversebits)\s*\(", hdr))
    declared = {d for d in declared if not d.endswith("_t")}
    assert declared == set(lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.vdl2gpu_abi_version() == 6


def test_exports_nothing_but_the_c_abi(built):
    """The product library is linked into C programs (dropin/): it exports the C ABI and the one symbol of d8psk.c
    that out.c still calls -- no kernel stubs, no C++ runtime symbols (-fvisibility=hidden + csrc/vdl2gpu.map; round 5 exported
    53 symbols, every __device_stub__ among them)."""
    import subprocess
    from vdlm2dec_amd import lib
    for so in ("libvdl2gpu.so", "libvdl2gpu_test.so"):
        out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "vdlm2dec_amd", so)], text=True)
        names = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
        assert names == set(lib.EXPORTS), (so, sorted(names ^ set(lib.EXPORTS)))


def test_struct_layouts_match_header(built):
    from vdlm2dec_amd import lib
    assert C.sizeof(lib.BurstT) == 64 + 8 * 255          # data at offset 64, see vdl2gpu_kernels.h
    assert lib.BurstT.data.offset == 64
    assert C.sizeof(lib.ChanT) == 12                      # == thread_param_t, vdlm2.h:49-52
    assert C.sizeof(lib.ConfigT) == 56


def test_reversebits_exported_for_host_path(built, oracle):
    from vdlm2dec_amd import lib
    L, O = lib.load(), oracle.lib()
    for v, n in ((0x2A, 6), (0x55, 7), (0x12345, 17), (1, 1)):
        assert L.reversebits(v, n) == O.vo_reversebits(v, n)


def test_burst_to_msgblk_layout(built):
    """msgblk_t on LP64 (vdlm2.h:39-47): chn@8 Fr@12 ppm@32 nbrow@36 nlbyte@40 data@44, 16624 bytes."""
    from vdlm2dec_amd import lib
    L = lib.load()
    b = lib.BurstT()
    b.chn, b.Fr, b.nbrow, b.nlbyte, b.ppm = 3, 136975000, 2, 17, -4.75
    for r in range(8):
        for i in range(255):
            b.data[r][i] = (r * 255 + i) & 0xFF
    blk = (C.c_uint8 * 16624)()
    assert L.vdl2gpu_burst_to_msgblk(C.byref(b), blk, 16624) == 0
    raw = bytes(blk)
    assert np.frombuffer(raw[8:16], "<i4").tolist() == [3, 136975000]
    assert np.frombuffer(raw[32:36], "<f4")[0] == np.float32(-4.75)
    assert np.frombuffer(raw[36:44], "<i4").tolist() == [2, 17]
    assert raw[44:44 + 8 * 255] == bytes(b.data)
    assert raw[:8] == b"\0" * 8 and raw[16:32] == b"\0" * 16 and set(raw[44 + 2040:]) == {0}
    assert L.vdl2gpu_burst_to_msgblk(C.byref(b), blk, 100) == -1


@pytest.mark.parametrize("rate", [2_000_000, 5_000_000, 6_000_000, 10_000_000])
def test_lo_table_equals_reference_formula(built, oracle, rate):
    """Host LO table (sincosf) == the oracle's cexpf table (d8psk.c:353-357), bit for bit."""
    from vdlm2dec_amd.demod import lo_table
    rng = np.random.default_rng(rate)
    fos = [-450000, -50000, 100000, 25000, 975000, -123457, 1] + [int(v) for v in rng.integers(-rate // 2, rate // 2, 40)]
    for fo in fos:
        ch = oracle.OracleChannel(rate, fo, 136_000_000 + fo)
        assert np.array_equal(ch.lo_table().view(np.uint32), lo_table(rate, fo).view(np.uint32)), fo
        ch.close()


def test_create_rejects_bad_config_and_missing_gpu(built):
    import torch
    from vdlm2dec_amd import lib
    from vdlm2dec_amd.demod import Receiver, ThreadParam
    L = lib.load()
    assert L.vdl2gpu_create(None, None) == -1
    with pytest.raises(lib.Vdl2GpuError):
        Receiver(2_000_000, [ThreadParam(0, 1, 1)], fmt="cu8", max_push=0)
    with pytest.raises(lib.Vdl2GpuError):
        Receiver(2_000_001, [ThreadParam(0, 1, 1)], fmt="cu8")
    if not torch.cuda.is_available():
        # the product must fail loudly, never fall back to a CPU path
        with pytest.raises(lib.Vdl2GpuError, match="no HIP device"):
            Receiver(2_000_000, [ThreadParam(0, 136975000, -50000)], fmt="cu8")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "vdlm2dec_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.replace("the oracle", "").replace("oracle's", "") or f in ("vdl2_math.h",), \
                    f"{f} mentions the oracle"


def test_test_handicaps_exist_only_in_the_test_build(built):
    """libvdl2gpu.so (the product) rejects VDL2GPU_F_TEST_NOREGION and has no code for the VDL2GPU_PRIM_DROP /
    VDL2GPU_SPLIT_SAMPLES handicaps; libvdl2gpu_test.so (-DVDL2GPU_TESTHOOKS) is what the tests load for them.
    The flag is checked before any device is touched, so this runs without a GPU."""
    from vdlm2dec_amd import lib
    prod, test = lib.load(), lib.load(testhooks=True)
    for name in lib.EXPORTS:
        assert hasattr(test, name), name
    chan = (lib.ChanT * 1)(lib.ChanT(0, 136_975_000, -50_000))
    cfg = lib.ConfigT(struct_size=C.sizeof(lib.ConfigT), sdrinrate=2_000_000, fmt=0, nbch=1, nstreams=1, chan=chan,
                      max_push=32768, flags=lib.F_TEST_NOREGION)
    h = C.c_void_p()
    assert prod.vdl2gpu_create(C.byref(cfg), C.byref(h)) == -1          # VDL2GPU_EINVAL, whatever the machine
    rc = test.vdl2gpu_create(C.byref(cfg), C.byref(h))
    assert rc in (0, -5)                                                    # accepted: a handle, or ENODEV without a GPU
    if rc == 0:
        test.vdl2gpu_destroy(h)
    blob = open(lib.LIB_PATH, "rb").read()
    assert b"VDL2GPU_PRIM_DROP" not in blob and b"VDL2GPU_SPLIT_SAMPLES" not in blob and b"VDL2GPU_TEST_ITEM" not in blob
    tblob = open(lib.LIB_TEST_PATH, "rb").read()
    assert b"VDL2GPU_PRIM_DROP" in tblob and b"VDL2GPU_TEST_ITEM_GRID" in tblob and b"VDL2GPU_TEST_ITEM_COMMON" in tblob
    src = open(os.path.join(ROOT, "vdlm2dec_amd", "csrc", "vdl2gpu.hip")).read()
    push = src[src.index("static int push_impl(vdl2gpu_t *h, const void *iq, size_t nsamples, size_t stream_stride_bytes, int memkind, bool wait_copy)\n{"):]
    push = push[:push.index("extern \"C\" int vdl2gpu_sync")]
    assert "getenv" not in push                                             # every knob is read once, in create_impl


def test_msgblk_offsets_are_checked_against_the_reference_header(tmp_path):
    """oracle/ref_layout_check.c (_Static_assert against the reference's own vdlm2.h) is part of `make -C oracle ref`;
    with one constant of include/vdl2gpu.h changed it must refuse to compile."""
    import subprocess
    if not os.path.isdir("/root/reference"):
        pytest.skip("needs /root/reference")
    src = os.path.join(ROOT, "oracle", "ref_layout_check.c")
    ok = subprocess.run(["gcc", "-std=c11", "-DWITH_RTL", "-I/root/reference", "-I" + os.path.join(ROOT, "include"), "-w",
                         "-c", "-o", str(tmp_path / "a.o"), src], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr
    hdr = open(os.path.join(ROOT, "include", "vdl2gpu.h")).read().replace("VDL2GPU_MSGBLK_OFF_NBROW 36", "VDL2GPU_MSGBLK_OFF_NBROW 40")
    (tmp_path / "vdl2gpu.h").write_text(hdr)
    bad = subprocess.run(["gcc", "-std=c11", "-DWITH_RTL", "-I/root/reference", "-I" + str(tmp_path), "-w",
                          "-c", "-o", str(tmp_path / "b.o"), src], capture_output=True, text=True)
    assert bad.returncode != 0 and "msgblk_t.nbrow" in bad.stderr
