"""Device build of vdl2_math.h vs the libm atan2f the reference links (through the oracle lib)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_device_atan2f_bit_exact(built, oracle):
    from vdlm2dec_amd.demod import Receiver, ThreadParam
    L = oracle.lib()
    rng = np.random.default_rng(4)
    n = 200_000
    y = (rng.standard_normal(n) * 10.0 ** rng.uniform(-3, 5, n)).astype(np.float32)
    x = (rng.standard_normal(n) * 10.0 ** rng.uniform(-3, 5, n)).astype(np.float32)
    sp = np.array([0.0, -0.0, 1.0, -1.0, 0.4375, 0.6875, 1.1875, 2.4375, 1e-30, 1e30, 3e-39], np.float32)
    yy, xx = np.meshgrid(sp, sp)
    y = np.concatenate([y, yy.ravel(), -yy.ravel()])
    x = np.concatenate([x, xx.ravel(), xx.ravel()])
    with Receiver(2_000_000, [ThreadParam(0, 136975000, -50000)], fmt="cu8", max_push=1024) as rx:
        g = rx.debug_atan2f(y, x)
    f = L.vo_atan2f
    ref = np.array([f(float(a), float(b)) for a, b in zip(y, x)], np.float32)
    assert np.array_equal(ref.view(np.uint32), g.view(np.uint32))
