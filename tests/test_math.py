"""vdl2_math.h (the fixed-sequence atan2f the kernels use) == the libm atan2f the reference links."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_atan2f_bit_exact_vs_libm(tmp_path):
    exe = str(tmp_path / "atan2_check")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe,
                           os.path.join(ROOT, "tests", "ctests", "atan2_check.c"), "-lm"])
    out = subprocess.run([exe, "30000000", "99"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "0 mismatches" in out.stdout


def test_fma_division_by_window_length_is_exact(tmp_path):
    exe = str(tmp_path / "div_check")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-o", exe,
                           os.path.join(ROOT, "tests", "ctests", "div_check.c"), "-lm"])
    out = subprocess.run([exe, "50000000", "7"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "0 mismatches" in out.stdout


def test_screen_bounds_hold_for_the_reference_fit():
    """The scan kernels skip the exact fit wherever one of three cheap statistics proves its error >= 4.25
    (vdl2gpu_dsp.h): err >= (16 - R)/2, (15 - R2)/2, (14 - R3)/2, with R, R2, R3 the magnitudes of the
    sums of lag-1, lag-2, lag-3 phase-step phasors.  Check the inequalities against the reference's own
    fit (d8psk.c:257-289, in double) on random, near-sync and adversarial windows."""
    import numpy as np
    rng = np.random.default_rng(5)
    SW = np.array([2, 3, 10, 15, 8, 9, 12, 9, 2, 5, 4, 9, 4, 1, -4, -5, 2]) * np.pi / 8

    def ref_fit(ph):                      # the reference's unwrap + line fit on 17 phases
        pr = np.empty(17)
        pv = ph[0] - SW[0]
        pr[0] = pv
        pu = 0.0
        for l in range(1, 17):
            pc = ph[l] - SW[l]
            pd = pc - pv
            pv = pc
            if pd > np.pi:
                pu -= 2 * np.pi
            elif pd < -np.pi:
                pu += 2 * np.pi
            pr[l] = pc + pu
        pr = pr - pr.mean()
        k = np.arange(17) - 8.0
        fr = (pr * k).sum() / 408.0
        return float(((pr - k * fr) ** 2).sum())

    def stats(ph):
        d = np.exp(1j * ((ph[1:] - ph[:-1]) - (SW[1:] - SW[:-1])))      # lag-1 step phasors
        r1 = abs(d.sum())
        r2 = abs((d[:-1] * d[1:]).sum())
        r3 = abs((d[:-2] * d[1:-1] * d[2:]).sum())
        return r1, r2, r3

    worst = 1e9
    for trial in range(6000):
        kind = trial % 4
        if kind == 0:                      # noise
            ph = rng.uniform(-np.pi, np.pi, 17)
        elif kind == 1:                    # a sync word with carrier offset and phase noise of varying strength
            ph = SW + rng.uniform(-np.pi, np.pi) + (np.arange(17) - 8) * rng.uniform(-1.0, 1.0) + rng.normal(0, rng.uniform(0, 0.8), 17)
        elif kind == 2:                    # alternating residuals: the worst case for the difference bounds
            ph = SW + 0.3 + (np.arange(17) - 8) * 0.2 + rng.uniform(0, 0.6) * (-1.0) ** np.arange(17)
        else:                              # half sync word, half noise (a stale ring)
            ph = np.where(np.arange(17) < rng.integers(1, 17), SW + 1.0, rng.uniform(-np.pi, np.pi, 17))
        ph = (ph + np.pi) % (2 * np.pi) - np.pi
        err = ref_fit(ph)
        r1, r2, r3 = stats(ph)
        bound = max((16 - r1) / 2, (15 - r2) / 2, (14 - r3) / 2)
        worst = min(worst, err - bound)
        assert err >= bound - 1e-9, (trial, err, bound, r1, r2, r3)
    assert worst > -1e-9
