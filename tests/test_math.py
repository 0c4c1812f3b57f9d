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
