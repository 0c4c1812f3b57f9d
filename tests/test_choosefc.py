"""Frequency planning: vdl2gpu_choose_fc_rtl / _air restate chooseFc() of rtl.c:123-160 and
air.c:47-70.  PARITY UNPINNED: rtl.c / air.c include the SDR vendor headers and cannot be compiled in the build
container, so these are hand-derived cases from the reference's loops plus an independent (vectorised, ascending)
statement of the same rules."""
import numpy as np
import pytest

from vdlm2dec_amd.demod import choose_fc

STEP = 25000


def rtl_rule(freqs, rate=2_000_000):
    """Independent statement of rtl.c:142-160: all admissible centres at once, the highest one wins."""
    fd = np.sort(np.asarray(freqs, np.int64))
    if fd[-1] - fd[0] > rate - 4 * STEP:
        return 0
    fc = np.arange(fd[0] - 2 * STEP + 1, fd[-1] + 2 * STEP + 1, dtype=np.int64)     # the loop's range, ascending
    d = np.abs(fc[:, None] - fd[None, :])
    ok = (d <= rate // 2 - 2 * STEP).all(1) & (d >= 2 * STEP).all(1)
    for n in range(1, len(fd)):
        ok &= (fc - fd[n - 1]) != (fd[n] - fc)
    return int(fc[ok].max()) if ok.any() else int(fd[0] - 2 * STEP)


@pytest.mark.parametrize("freqs", [
    [136_975_000],
    [136_975_000, 136_725_000, 136_775_000],
    [136_650_000, 136_700_000, 136_800_000, 136_975_000],
    [136_000_000, 137_900_000],                       # 1.9 MHz apart: the widest span rtl.c accepts
    [136_725_000, 136_775_000, 136_825_000, 136_875_000, 136_925_000, 136_975_000, 136_675_000, 136_625_000],
    [136_900_000, 137_000_000],                       # the first candidate is the mirror point of nothing: plain max + 50 kHz
])
def test_rtl_centre_matches_the_rules(built, freqs):
    fc, plan = choose_fc(freqs, 2_000_000, "rtl")
    assert fc == rtl_rule(freqs)
    assert [p.Fr for p in plan] == list(freqs)                  # caller's order kept (rtl.c:225-228)
    assert [p.Fo for p in plan] == [f - fc for f in freqs]      # rtl.c:245-247


def test_rtl_hand_derived(built):
    # one channel: Fc starts at Fr + 50 kHz; |Fc - Fr| = 50 kHz is neither > 950 kHz nor < 50 kHz -> taken at once
    assert choose_fc([136_975_000])[0] == 137_025_000
    # two channels 100 kHz apart: Fc = 137_050_000 passes (distances 50/150 kHz; not the mirror point 136_950_000)
    fc, plan = choose_fc([136_900_000, 137_000_000])
    assert fc == 137_050_000 and [p.Fo for p in plan] == [-150_000, -50_000]
    # span > SDRINRATE - 100 kHz: refused like the reference ("Frequencies too far apart", returns 0)
    assert choose_fc([136_000_000, 137_900_001])[0] == 0
    assert choose_fc([131_725_000, 136_975_000])[0] == 0
    # the mirror rule: channels at +-75 kHz of a candidate forbid exactly that candidate
    fc, _ = choose_fc([136_000_000, 137_850_000])
    assert fc == rtl_rule([136_000_000, 137_850_000]) and abs(fc - 137_850_000) >= 50_000 and abs(fc - 136_000_000) <= 950_000


def test_air_hand_derived(built):
    # 6 MS/s (Airspy Mini): no filter offset, midpoint rounded to the 25 kHz grid; Fo = Fr - (Fc + 1.5 MHz)
    fc, plan, regs = choose_fc([136_725_000, 136_975_000], 6_000_000, "air")
    assert fc == 136_850_000 and regs == (0, 0)
    assert [p.Fo for p in plan] == [136_725_000 - 138_350_000, 136_975_000 - 138_350_000]
    # 5 MS/s (Airspy R2): bw = 250 kHz + 50 kHz; i = 6 (2087988 - 1715133 = 372855 >= 300000, i = 7 gives 234366),
    # j: 2001344 - 1715133 = 286211 <= 300000 at j = 2, so j = 3; off = (2032592 + 1715133)/2 - 1250000 = 623862;
    # Fc = (136850000 + 623862 + 12500) / 25000 * 25000 = 137475000; registers 0xB0|12, 0xE0|9
    fc, plan, regs = choose_fc([136_975_000, 136_725_000, 136_775_000], 5_000_000, "air")
    assert fc == 137_475_000 and regs == (0xBC, 0xE9)
    assert [p.Fo for p in plan] == [f - (137_475_000 + 1_250_000) for f in (136_975_000, 136_725_000, 136_775_000)]
    # a span no filter pair covers: the reference returns 0
    assert choose_fc([136_000_000, 137_600_000], 5_000_000, "air")[0] == 0
