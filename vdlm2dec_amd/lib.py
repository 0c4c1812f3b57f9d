"""ctypes binding of libvdl2gpu.so (the C ABI in include/vdl2gpu.h).

The library is the product; this module only loads it.  There is no Python or
CPU fallback: if the shared object is missing, or no HIP device is present when a
handle is created, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvdl2gpu.so")
LIB_TEST_PATH = os.path.join(_HERE, "libvdl2gpu_test.so")   # the same sources with -DVDL2GPU_TESTHOOKS (tests only)

FMT = {"cu8": 0, "cs16": 1, "cf32": 2, "f32": 3}
SAMPLE_BYTES = {"cu8": 2, "cs16": 4, "cf32": 8, "f32": 4}
MEM_HOST, MEM_DEVICE = 0, 1
F_KEEP_DEC = 1
F_SERIAL = 2
F_FULLSCAN = 4
F_TEST_NOREGION = 8
F_FRAMES = 16
F_RTL_QUIRK = 32
F_DEBUG_HEADS = 64

# every symbol include/vdl2gpu.h declares
EXPORTS = (
    "vdl2gpu_abi_version", "vdl2gpu_create", "vdl2gpu_destroy", "vdl2gpu_push", "vdl2gpu_sync",
    "vdl2gpu_ring_init", "vdl2gpu_ring_acquire", "vdl2gpu_ring_commit",
    "vdl2gpu_poll", "vdl2gpu_poll_ready", "vdl2gpu_pending", "vdl2gpu_inflight", "vdl2gpu_get_stats", "vdl2gpu_get_timing", "vdl2gpu_get_host_profile", "vdl2gpu_last_error",
    "vdl2gpu_strerror", "vdl2gpu_burst_to_msgblk", "vdl2gpu_decode_blocks", "vdl2gpu_poll_frames", "vdl2gpu_poll_frames_ready", "reversebits", "vdl2gpu_lo_table", "vdl2gpu_plan",
    "vdl2gpu_choose_fc_rtl", "vdl2gpu_choose_fc_air",
    "vdl2gpu_debug_dec", "vdl2gpu_debug_lo", "vdl2gpu_debug_atan2f", "vdl2gpu_debug_counters", "vdl2gpu_debug_cands", "vdl2gpu_debug_clheads", "vdl2gpu_debug_fail", "vdl2gpu_debug_segs", "vdl2gpu_debug_heads",
)


class ChanT(C.Structure):
    _fields_ = [("chn", C.c_int32), ("Fr", C.c_int32), ("Fo", C.c_int32)]


class ConfigT(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("sdrinrate", C.c_uint32), ("sdrclk", C.c_uint32),
                ("fmt", C.c_int32), ("nbch", C.c_int32), ("nstreams", C.c_int32),
                ("chan", C.POINTER(ChanT)), ("max_push", C.c_uint64), ("device", C.c_int32),
                ("max_bursts", C.c_uint32), ("flags", C.c_uint32)]


class BurstT(C.Structure):
    _fields_ = [("stream", C.c_int32), ("chn", C.c_int32), ("Fr", C.c_int32), ("nbrow", C.c_int32),
                ("nlbyte", C.c_int32), ("df", C.c_float), ("ppm", C.c_float),
                ("trig_dec", C.c_int64), ("end_dec", C.c_int64), ("trig_sample", C.c_int64),
                ("end_sample", C.c_int64), ("data", (C.c_uint8 * 255) * 8)]


class FrameT(C.Structure):
    _fields_ = [("stream", C.c_int32), ("chn", C.c_int32), ("Fr", C.c_int32), ("nbrow", C.c_int32),
                ("nlbyte", C.c_int32), ("len", C.c_int32), ("block", C.c_int32), ("seq", C.c_int32),
                ("df", C.c_float), ("ppm", C.c_float), ("trig_dec", C.c_int64), ("end_dec", C.c_int64),
                ("data", C.c_uint8 * 2000)]


class StatsT(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples_in", "dec_samples", "sync_evals", "triggers",
                                          "header_rejects", "bursts", "deferrals", "candidates", "serial_redos", "serial_samples", "overflowed",
                                          "frames_dropped", "repairs")]


class TimingT(C.Structure):
    _fields_ = [("channelise_ms", C.c_double), ("demod_ms", C.c_double), ("scan_ms", C.c_double),
                ("cluster_ms", C.c_double), ("resolve_ms", C.c_double), ("other_ms", C.c_double),
                ("channelise_fast_ms", C.c_double), ("fast_pushes", C.c_uint64),
                ("pushes", C.c_uint64), ("samples", C.c_uint64)]


class Vdl2GpuError(RuntimeError):
    pass


_libs = {}


def load(testhooks: bool = False):
    """Load libvdl2gpu.so (or, for the tests' handicaps, libvdl2gpu_test.so); raises if the HIP
    extension has not been built."""
    if testhooks in _libs:
        return _libs[testhooks]
    path = LIB_TEST_PATH if testhooks else LIB_PATH
    if not testhooks and os.environ.get("VDL2GPU_LIB"):
        path = os.environ["VDL2GPU_LIB"]      # development: another build of the same library (scripts/dev/abv.sh compares variants on one box)
    if testhooks and os.environ.get("VDL2GPU_LIB_TEST"):
        path = os.environ["VDL2GPU_LIB_TEST"]  # development: another -DVDL2GPU_TESTHOOKS build (bisecting with scripts/soak.py)
    if not os.path.exists(path):
        raise Vdl2GpuError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    if not os.environ.get("VDL2GPU_NO_TORCH"):
        # PyTorch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64; two HSA runtimes in
        # one process cannot both own the GPU.  Importing torch first makes its runtime the one this
        # library binds to (same SONAME), so tensors' device pointers can be pushed directly.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(path)
    L.vdl2gpu_abi_version.restype = C.c_int
    L.vdl2gpu_create.restype = C.c_int
    L.vdl2gpu_create.argtypes = [C.POINTER(ConfigT), C.POINTER(C.c_void_p)]
    L.vdl2gpu_destroy.restype = None
    L.vdl2gpu_destroy.argtypes = [C.c_void_p]
    L.vdl2gpu_push.restype = C.c_int
    L.vdl2gpu_push.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
    L.vdl2gpu_ring_init.restype = C.c_int
    L.vdl2gpu_ring_init.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    L.vdl2gpu_ring_acquire.restype = C.c_void_p
    L.vdl2gpu_ring_acquire.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    L.vdl2gpu_ring_commit.restype = C.c_int
    L.vdl2gpu_ring_commit.argtypes = [C.c_void_p, C.c_size_t]
    L.vdl2gpu_sync.restype = C.c_int
    L.vdl2gpu_sync.argtypes = [C.c_void_p]
    L.vdl2gpu_poll.restype = C.c_int
    L.vdl2gpu_poll.argtypes = [C.c_void_p, C.POINTER(BurstT), C.c_int]
    L.vdl2gpu_poll_ready.restype = C.c_int
    L.vdl2gpu_poll_ready.argtypes = [C.c_void_p, C.POINTER(BurstT), C.c_int]
    L.vdl2gpu_pending.restype = C.c_int
    L.vdl2gpu_pending.argtypes = [C.c_void_p]
    L.vdl2gpu_get_stats.restype = C.c_int
    L.vdl2gpu_get_stats.argtypes = [C.c_void_p, C.POINTER(StatsT)]
    L.vdl2gpu_get_timing.restype = C.c_int
    L.vdl2gpu_get_timing.argtypes = [C.c_void_p, C.POINTER(TimingT), C.c_int]
    L.vdl2gpu_get_host_profile.restype = C.c_int
    L.vdl2gpu_get_host_profile.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    L.vdl2gpu_last_error.restype = C.c_char_p
    L.vdl2gpu_last_error.argtypes = [C.c_void_p]
    L.vdl2gpu_strerror.restype = C.c_char_p
    L.vdl2gpu_strerror.argtypes = [C.c_int]
    L.vdl2gpu_burst_to_msgblk.restype = C.c_int
    L.vdl2gpu_burst_to_msgblk.argtypes = [C.POINTER(BurstT), C.c_void_p, C.c_size_t]
    L.vdl2gpu_decode_blocks.restype = C.c_int
    L.vdl2gpu_decode_blocks.argtypes = [C.c_void_p, C.POINTER(BurstT), C.c_int, C.POINTER(FrameT), C.c_int,
                                        C.POINTER(C.c_int)]
    L.vdl2gpu_poll_frames.restype = C.c_int
    L.vdl2gpu_poll_frames.argtypes = [C.c_void_p, C.POINTER(FrameT), C.c_int]
    L.vdl2gpu_poll_frames_ready.restype = C.c_int
    L.vdl2gpu_poll_frames_ready.argtypes = [C.c_void_p, C.POINTER(FrameT), C.c_int]
    L.reversebits.restype = C.c_uint
    L.reversebits.argtypes = [C.c_uint, C.c_int]
    L.vdl2gpu_lo_table.restype = C.c_int
    L.vdl2gpu_lo_table.argtypes = [C.c_uint, C.c_int, C.c_void_p, C.c_int]
    L.vdl2gpu_plan.restype = C.c_int
    L.vdl2gpu_plan.argtypes = [C.c_uint64, C.c_uint64, C.c_uint, C.c_uint, C.POINTER(C.c_int),
                               C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    L.vdl2gpu_choose_fc_rtl.restype = C.c_int
    L.vdl2gpu_choose_fc_rtl.argtypes = [C.POINTER(C.c_uint), C.c_int, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_int)]
    L.vdl2gpu_choose_fc_air.restype = C.c_int
    L.vdl2gpu_choose_fc_air.argtypes = [C.POINTER(C.c_uint), C.c_int, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.vdl2gpu_debug_dec.restype = C.c_int64
    L.vdl2gpu_debug_dec.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]
    L.vdl2gpu_debug_lo.restype = C.c_int
    L.vdl2gpu_debug_lo.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.vdl2gpu_debug_atan2f.restype = C.c_int
    L.vdl2gpu_debug_atan2f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.vdl2gpu_debug_fail.restype = C.c_int
    L.vdl2gpu_debug_fail.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.vdl2gpu_debug_segs.restype = C.c_int
    L.vdl2gpu_debug_segs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.vdl2gpu_debug_cands.restype = C.c_int
    L.vdl2gpu_debug_cands.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.vdl2gpu_debug_clheads.restype = C.c_int
    L.vdl2gpu_debug_clheads.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.vdl2gpu_debug_heads.restype = C.c_int
    L.vdl2gpu_debug_heads.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.vdl2gpu_debug_counters.restype = C.c_int
    L.vdl2gpu_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    _libs[testhooks] = L
    return L
