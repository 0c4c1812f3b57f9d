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


# ---------------------------------------------------------------------------------------------------------------
# Brute force: still UNPINNED -- rtl.c / air.c cannot be compiled here -- so the restatement in
# libvdl2gpu.so is held against a second, structurally different statement of the same rules on >= 10^5 channel sets.

def rtl_rule_intervals(freqs, rate=2_000_000):
    """rtl.c:142-160 without a loop over centres: the admissible set is an interval (every channel within
    SDRINRATE/2 - 50 kHz) minus the open 50 kHz neighbourhoods of the channels minus the mirror points of neighbours;
    the reference takes its highest element above min - 50 kHz, else min - 50 kHz."""
    fd = sorted(int(f) for f in freqs)
    if fd[-1] - fd[0] > rate - 4 * STEP:
        return 0
    lim = rate // 2 - 2 * STEP
    lo_excl = fd[0] - 2 * STEP                       # the loop stops BEFORE this value
    c = min(fd[-1] + 2 * STEP, fd[0] + lim)          # highest centre with every channel within `lim`
    low = fd[-1] - lim                               # lowest such centre
    mirrors = {(a + b) // 2 for a, b in zip(fd, fd[1:]) if (a + b) % 2 == 0}
    while c > lo_excl and c >= low:
        moved = False
        for f in fd:
            if abs(c - f) < 2 * STEP:                # inside a channel's neighbourhood: below it
                c = f - 2 * STEP
                moved = True
        if c in mirrors:
            c -= 1
            moved = True
        if not moved:
            return c
    return lo_excl


def _grid(seed, n=40, width=86):
    """40 channels on a 25 kHz grid spread over 2.15 MHz (spans beyond SDRINRATE - 100 kHz occur), both ends taken"""
    rng = np.random.default_rng(seed)
    k = np.sort(np.concatenate([[0, width - 1], rng.choice(np.arange(1, width - 1), n - 2, replace=False)]))
    return [136_000_000 + STEP * int(x) for x in k]


def test_rtl_brute_force_over_a_40_channel_grid(built):
    """All 1-, 2- and 3-subsets of a 40-channel 25 kHz grid and 90 000 random 4..8-subsets (mirror-image pairs are
    everywhere on a regular grid; spans over SDRINRATE - 100 kHz are refused) : >= 10^5 channel sets, library ==
    interval statement; a random 300 of them also against the exhaustive numpy statement above."""
    import itertools
    grid = _grid(1)
    rng = np.random.default_rng(2)
    cases = [list(c) for k in (1, 2, 3) for c in itertools.combinations(grid, k)]
    while len(cases) < 100_800:
        k = int(rng.integers(4, 9))
        cases.append([grid[i] for i in rng.choice(40, k, replace=False)])
    refused = mirrors = walked = 0
    for fr in cases:
        fc, plan = choose_fc(fr, 2_000_000, "rtl")
        want = rtl_rule_intervals(fr)
        assert fc == want, fr
        assert [p.Fo for p in plan] == ([0] * len(fr) if fc == 0 else [f - fc for f in fr])
        refused += fc == 0
        walked += fc != 0 and fc != max(fr) + 2 * STEP
        s = sorted(fr)
        mirrors += any((a + b) % 2 == 0 and min(s) - 2 * STEP < (a + b) // 2 <= max(s) + 2 * STEP for a, b in zip(s, s[1:]))
    assert len(cases) >= 100_000 and refused > 1000 and walked > 10_000 and mirrors > 50_000
    for i in rng.choice(len(cases), 300, replace=False):
        assert rtl_rule(cases[i]) == rtl_rule_intervals(cases[i]), cases[i]


def test_rtl_mirror_points_and_odd_spacings(built):
    """Off-grid sets (chooseFc steps in 1 Hz): mirror points exist only for even sums; neighbourhood edges are
    inclusive at exactly 50 kHz."""
    rng = np.random.default_rng(3)
    for _ in range(3000):
        k = int(rng.integers(1, 9))
        fr = [int(x) for x in 136_000_000 + rng.integers(0, 1_950_000, k)]
        assert choose_fc(fr, 2_000_000, "rtl")[0] == rtl_rule_intervals(fr), fr
    # the candidate max + 50 kHz is the mirror point of the two highest channels when they are 100 kHz ... no: the
    # mirror of (a, b) lies between them; a centre BELOW the top channel by d and above the next by d is forbidden
    assert choose_fc([136_000_000, 136_900_000, 137_100_000], 2_000_000)[0] == rtl_rule_intervals([136_000_000, 136_900_000, 137_100_000])


HF = [1953050, 1980748, 2001344, 2032592, 2060291, 2087988]           # r820t_hf, air.c:44
LF = [525548, 656935, 795424, 898403, 1186034, 1502073, 1715133, 1853622]   # r820t_lf, air.c:45


def air_rule(minf, maxf, rate):
    """air.c:47-70, stated without its loops: i = the widest low-pass edge whose pass band up to hf[5] still covers the
    span + 50 kHz; j = the first high-pass edge above it that leaves MORE than the span."""
    bw = maxf - minf + 2 * STEP
    off, r10, r11 = 0, 0, 0
    if rate == 5_000_000:
        ii = [i for i in range(8) if HF[5] - LF[i] >= bw]
        if not ii:
            return 0, 0, 0
        i = max(ii)
        jj = [j for j in range(6) if HF[j] - LF[i] <= bw]
        j = (max(jj) + 1) if jj else 0
        j = min(j, 5)           # equality at j = 5 would index past the table in the reference; the library clamps
        off = (HF[j] + LF[i]) // 2 - rate // 4
        r10, r11 = 0xB0 | (15 - j), 0xE0 | (15 - i)
    return ((maxf + minf) // 2 + off + STEP // 2) // STEP * STEP, r10, r11


def test_air_registers_boundary_by_boundary(built):
    """Every boundary of the R820T2 filter table (air.c:55-68): spans whose bw = span + 50 kHz sits at, one below and
    one above every hf[j] - lf[i], at 5 MS/s; plus the 40-channel grid's (min, max) pairs at 5 and 6 MS/s."""
    n = 0
    base = 136_000_000
    for i in range(8):
        for j in range(6):
            for d in (-1, 0, 1):
                bw = HF[j] - LF[i] + d
                span = bw - 2 * STEP
                if span < 0:
                    continue
                if j == 5 and d == 0:
                    continue        # bw == hf[5] - lf[i] exactly: the reference reads r820t_hf[6], one past the table
                fr = [base, base + span]
                fc, plan, regs = choose_fc(fr, 5_000_000, "air")
                want = air_rule(base, base + span, 5_000_000)
                assert (fc, regs[0], regs[1]) == want, (i, j, d)
                if fc:
                    assert [p.Fo for p in plan] == [f - (fc + 1_250_000) for f in fr]
                n += 1
    grid = _grid(5)
    for rate in (5_000_000, 6_000_000):
        for a in range(40):
            for b in range(a, 40):
                fr = [grid[b], grid[a]]
                fc, plan, regs = choose_fc(fr, rate, "air")
                assert (fc, regs[0], regs[1]) == air_rule(grid[a], grid[b], rate), (rate, a, b)
                n += 1
    assert n >= 1700
