"""Parity tests proper: the HIP path, called through the C ABI (vdlm2dec_amd.demod.Receiver is a thin
ctypes shim over include/vdl2gpu.h), against
  * the golden vectors produced by the REAL reference (tests/golden), and
  * the oracle on the same seeded inputs,
bit-exact at every level: 84 kS/s samples (P3), msgblk_t records (P2), CRC-clean frames (P1)."""
import glob
import json
import os

import numpy as np
import pytest

import scenarios as S
from vdlm2dec_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "*.json")))


def _rx(rate, fos, fmt, **kw):
    from vdlm2dec_amd.demod import Receiver, plan_channels
    return Receiver(rate, plan_channels(S.FC, fos), fmt=fmt, **kw)


def _gpu_keys(bursts):
    return sorted((b.chn, b.nbrow, b.nlbyte, b.data) for b in bursts)


def _frames(O, bursts):
    out = []
    for b in sorted(bursts, key=lambda b: (b.chn, b.end_dec)):
        out += [(b.chn, f) for f in O.frames_of_block(b.nbrow, b.nlbyte, b.data)]
    return out


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-5] for p in CASES])
@pytest.mark.parametrize("block", [None, 32768, 50001])
def test_golden_vectors_from_the_reference(built, oracle, path, block):
    meta = json.load(open(path))
    raw = np.load(os.path.join(HERE, "golden", meta["iq"] + ".npz"))["raw"]
    if meta["quirk"] and block == 50001:
        pytest.skip("VDL2GPU_F_RTL_QUIRK takes whole 32768-sample blocks (rtl.c:278-281 drops anything else)")
    with _rx(meta["rate"], meta["fo"], meta["fmt"], max_push=1 << 21, rtl_quirk=bool(meta["quirk"])) as rx:
        got = rx.run(raw, block=block)
    want = sorted((c["chn"], b["nbrow"], b["nlbyte"], bytes.fromhex(b["data"])) for c in meta["channels"] for b in c["blocks"])
    assert _gpu_keys(got) == want                                   # msgblk_t level (pre-RS)
    by = {}
    for b in got:
        by.setdefault(b.chn, []).append(b)
    for c in meta["channels"]:
        mine = sorted(by.get(c["chn"], []), key=lambda b: b.end_dec)
        assert [int(np.float32(b.df).view(np.uint32)) for b in mine] == [g["df_bits"] for g in c["blocks"]]
        assert [int(np.float32(b.ppm).view(np.uint32)) for b in mine] == [g["ppm_bits"] for g in c["blocks"]]
        fr = [f.hex() for b in mine for f in oracle.frames_of_block(b.nbrow, b.nlbyte, b.data)]
        assert fr == c["frames"]                                    # CRC-pass frame level


SCEN = {
    "regimes_cu8": (lambda: S.regimes(seed=301), "cu8"),
    "regimes_cs16": (lambda: S.regimes(seed=302, infos=(1, 2, 3, 28, 31, 60, 66, 70, 200, 247, 250, 497, 900)), "cs16"),
    "eight_cs16": (lambda: S.eight_channels(seed=303), "cs16"),
    "eight_cu8": (lambda: S.eight_channels(seed=304), "cu8"),
    "back_to_back": (lambda: S.back_to_back(), "cu8"),
    "odd_headers": (lambda: S.odd_headers(), "cs16"),
    "corrupted": (lambda: S.corrupted(), "cu8"),
    "noisy": (lambda: S.regimes(seed=305, infos=(20, 50, 90, 30, 10, 77), noise=6.0), "cu8"),
    "ten_ms": (lambda: S.single_short(10_000_000, 2_400_000, seed=306, info_len=33, blocks=6), "cs16"),
    "air_real": (lambda: S.single_short(6_000_000, 1_200_000, seed=307, info_len=14, blocks=5), "f32"),
    "cf32": (lambda: S.regimes(seed=308, infos=(7, 35, 80)), "cf32"),
    # BASELINE.json configs[2] at its stated size: 8 channels @ 10 MS/s; and the Airspy shape, 8 channels of real f32 @ 5 MS/s
    "eight_cs16_10ms": (lambda: S.eight_channels(rate=10_000_000, seed=309, dur=0.06, info=(3, 17, 40, 24, 30, 13, 5, 45), fo=S.FO8_10MS), "cs16"),
    "eight_f32_5ms": (lambda: S.eight_channels(rate=5_000_000, seed=311, dur=0.06, info=(3, 17, 40, 24, 30, 13, 5, 45), fo=S.FO8_AIR_5MS), "f32"),
    "eight_cu8_6ms": (lambda: S.eight_channels(rate=6_000_000, seed=312, dur=0.05, info=(3, 9, 5, 12, 7, 4, 6, 8), fo=S.FO8_AIR_5MS), "cu8"),
}


@pytest.mark.parametrize("name", sorted(SCEN))
def test_against_oracle_all_levels(built, oracle, name):
    mk, fmt = SCEN[name]
    spec = mk()
    raw = synth.synth_stream(spec, fmt)
    want = oracle.run_oracle(raw, fmt, spec.rate, spec.fo, S.FC)
    with _rx(spec.rate, spec.fo, fmt, max_push=spec.nsamples, keep_dec=True) as rx:
        got = rx.run(raw)
        assert _gpu_keys(got) == sorted(b.key() for b in want)
        # same instants, same carrier estimate
        assert sorted((b.chn, b.trig_dec, b.end_dec, np.float32(b.df).view(np.uint32).item()) for b in got) == \
            sorted((b.chn, b.trig_dec, b.end_dec, np.float32(b.df).view(np.uint32).item()) for b in want)
        # P3: the whole 84 kS/s stream of every channel, bit for bit
        for c, fo in enumerate(spec.fo):
            ch = oracle.OracleChannel(spec.rate, fo, S.FC + fo, tap_dec=True)
            ch.feed(raw, fmt)
            d, g = ch.dec(), rx.debug_dec(0, c)
            assert len(d) == len(g) and np.array_equal(d.view(np.uint32), g.view(np.uint32)), (name, c)
            ch.close()
        st = rx.stats()
        assert st["bursts"] == len(want) and st["overflowed"] == 0
    assert _frames(oracle, got) == _frames(oracle, want)


def test_rtl_quirk_mode_equals_the_oracle(built, oracle):
    """VDL2GPU_F_RTL_QUIRK = in_callback() as written (rtl.c:285-292): per 32768-sample block sample k lands in slot
    k+1, slot 0 is 0, the last sample is lost.  Against the oracle's cu8_quirk conversion (pinned to the real
    reference's quirk build in test_oracle_vs_ref.py and tests/golden/regimes_cu8_2ms_quirk.json), at every
    level including the 84 kS/s stream; pushes of one and of several blocks."""
    spec = S.regimes(seed=320, infos=(5, 40, 70, 250))
    raw = synth.synth_stream(spec, "cu8")
    want = oracle.run_oracle(raw, "cu8_quirk", spec.rate, spec.fo, S.FC)
    assert len(want) >= 3
    for block in (32768, 3 * 32768, None):
        with _rx(spec.rate, spec.fo, "cu8", max_push=spec.nsamples, rtl_quirk=True) as rx:
            got = rx.run(raw, block=block)
            assert _gpu_keys(got) == sorted(b.key() for b in want)
            assert sorted((b.chn, b.trig_dec, np.float32(b.df).view(np.uint32).item()) for b in got) == \
                sorted((b.chn, b.trig_dec, np.float32(b.df).view(np.uint32).item()) for b in want)
    with _rx(spec.rate, spec.fo, "cu8", max_push=spec.nsamples, rtl_quirk=True) as rx:
        rx.push(raw)
        for c, fo in enumerate(spec.fo):
            ch = oracle.OracleChannel(spec.rate, fo, S.FC + fo, tap_dec=True)
            ch.feed(raw, "cu8_quirk")
            d, g = ch.dec(), rx.debug_dec(0, c)
            assert len(d) == len(g) and np.array_equal(d.view(np.uint32), g.view(np.uint32))
            ch.close()
        from vdlm2dec_amd import lib
        with pytest.raises(lib.Vdl2GpuError):
            rx.push(raw[:2 * 1000])         # not a whole block
    from vdlm2dec_amd import lib
    with pytest.raises(lib.Vdl2GpuError):
        _rx(spec.rate, spec.fo, "cs16", rtl_quirk=True)     # the quirk is in_callback's: cu8 only


@pytest.mark.parametrize("blocks", [[1], [1, 2, 3, 5, 7, 11], [23], [24], [2000], [4096, 1, 4095], [100000, 17]])
def test_ragged_pushes_equal_one_shot(built, oracle, blocks):
    """Any way of cutting the stream into pushes (down to single samples) gives the same bursts:
    decimator carry, sync state and deferred bursts survive every boundary."""
    spec = S.regimes(seed=310, infos=(3, 40, 70))
    raw = synth.synth_stream(spec, "cu8")
    want = sorted(b.key() for b in oracle.run_oracle(raw, "cu8", spec.rate, spec.fo, S.FC))
    n = spec.nsamples
    if blocks == [1]:
        n = 40000           # single-sample pushes: keep it short (covers the first burst's sync)
        want = None
    with _rx(spec.rate, spec.fo, "cu8", max_push=1 << 20, keep_dec=True) as rx:
        pos, i, got = 0, 0, []
        decs = []
        while pos < n:
            k = min(blocks[i % len(blocks)], n - pos)
            rx.push(raw[2 * pos:2 * (pos + k)])
            if blocks == [1] or len(blocks) > 3:
                decs.append(rx.debug_dec(0, 0))
            pos += k
            i += 1
        got = rx.poll()
        if want is not None:
            assert _gpu_keys(got) == want
        if decs:
            ch = oracle.OracleChannel(spec.rate, spec.fo[0], S.FC + spec.fo[0], tap_dec=True)
            ch.feed(raw[:2 * n], "cu8")
            d = ch.dec()
            g = np.concatenate(decs) if decs else np.zeros(0, np.complex64)
            assert len(g) == len(d) and np.array_equal(d.view(np.uint32), g.view(np.uint32))


@pytest.mark.parametrize("fmt", ["cs16", "cu8", "cf32", "f32"])
@pytest.mark.parametrize("nch", [1, 3, 8])
def test_channeliser_push_sizes_around_its_tickets(built, oracle, fmt, nch):
    """k1_fast hands its superperiods (8000 samples) out in tickets of 8 per role and XCD, two to an iteration: pushes of
    3 .. 130 superperiods with every remainder (no ticket for most workgroups, half a pair at the end, one lonely
    superperiod), back to back so that the schedule's phase and the LO phase differ from push to push, with 1, 3 and 8
    channels (a wavefront without any channel still keeps its loads and stores counted).  The 84 kS/s planes must be
    the oracle's bit for bit."""
    if fmt == "f32" and nch > 4:
        pytest.skip("real input: mirror-image offsets interfere; covered with 1 and 3 channels")
    fos = S.FO8_AIR_5MS[:nch] if fmt == "f32" else S.FO8[:nch]
    # (the first three start on a window boundary and are whole superperiods: k1_fast takes them alone, no general launch)
    sizes = [24000, 8000 * 4, 8000 * 17, 25001, 8000 * 5 + 3, 8000 * 11 + 123, 8000 * 18, 8000 * 19 + 7999, 8000 * 66 + 5, 8000 * 130 + 77, 8000 * 9]
    n = sum(sizes)
    rng = np.random.default_rng(4242 + nch)
    if fmt == "cs16":
        raw = rng.integers(-3000, 3000, 2 * n, dtype=np.int16)
    elif fmt == "cu8":
        raw = rng.integers(0, 256, 2 * n, dtype=np.uint8)
    elif fmt == "cf32":
        raw = rng.normal(0, 300, 2 * n).astype(np.float32)
    else:
        raw = rng.normal(0, 300, n).astype(np.float32)
    per = 1 if fmt == "f32" else 2
    with _rx(2_000_000, fos, fmt, max_push=1 << 21, keep_dec=True) as rx:
        decs = {c: [] for c in {0, nch - 1}}
        pos = 0
        for k in sizes:
            rx.push(raw[per * pos:per * (pos + k)])
            for c in decs:
                decs[c].append(rx.debug_dec(0, c))
            pos += k
        rx.poll()
    for c, parts in decs.items():
        ch = oracle.OracleChannel(2_000_000, fos[c], S.FC + fos[c], tap_dec=True)
        ch.feed(raw, fmt)
        d, g = ch.dec(), np.concatenate(parts)
        ch.close()
        assert len(g) == len(d) and np.array_equal(d.view(np.uint32), g.view(np.uint32)), (fmt, nch, c)


@pytest.mark.parametrize("rate,fmt", [(10_000_000, "cs16"), (6_000_000, "cu8"), (5_000_000, "cs16")])
def test_period_parallel_channeliser_whole_and_ragged_pushes(built, oracle, rate, fmt):
    """k1_pp (5 / 6 / 10 MS/s) takes whole periods of the dump schedule (4 * SDRCLK samples = 84 outputs); a push that starts on a
    schedule boundary and is a whole number of periods is ONE launch since round 5 (no general-kernel launch at either end, the
    kernel leaves the stream state itself), any other push runs its first and last period through the general kernel.  Both kinds
    back to back, so that carried windows meet whole pushes: the 84 kS/s planes must be the oracle's bit for bit."""
    per = 4 * (rate // 4000)
    fos = (S.FO8_10MS if rate == 10_000_000 else S.FO8)[:8]
    sizes = [per * 8, per * 5, per * 64, per * 6 + 17, per * 9 - 17, per * 4, per * 130, per * 7 + 1, per * 12 - 1, per * 66]
    n = sum(sizes)
    rng = np.random.default_rng(rate // 1000)
    raw = rng.integers(-3000, 3000, 2 * n, dtype=np.int16) if fmt == "cs16" else rng.integers(0, 256, 2 * n, dtype=np.uint8)
    with _rx(rate, fos, fmt, max_push=1 << 22, keep_dec=True) as rx:
        decs = {c: [] for c in (0, 7)}
        pos = 0
        for k in sizes:
            rx.push(raw[2 * pos:2 * (pos + k)])
            for c in decs:
                decs[c].append(rx.debug_dec(0, c))
            pos += k
        rx.poll()
    for c, parts in decs.items():
        ch = oracle.OracleChannel(rate, fos[c], S.FC + fos[c], tap_dec=True)
        ch.feed(raw, fmt)
        d, g = ch.dec(), np.concatenate(parts)
        ch.close()
        assert len(g) == len(d) and np.array_equal(d.view(np.uint32), g.view(np.uint32)), (rate, fmt, c)


@pytest.mark.parametrize("fmt", ["cs16", "cu8"])
def test_long_pushes_are_cut_into_parts(built, oracle, monkeypatch, fmt):
    """A push longer than ~36 s of air time is cut into equal parts inside the library (the tables hold what a busy
    channel triggers in about that long); here the limit is lowered to 262144 samples: bursts and the 84 kS/s planes
    are those of the oracle, and of the same recording pushed whole."""
    spec = S.regimes(seed=311)
    assert spec.nsamples > 3 * 262144       # four parts
    raw = synth.synth_stream(spec, fmt)
    want = sorted(b.key() for b in oracle.run_oracle(raw, fmt, spec.rate, spec.fo, S.FC))
    with _rx(spec.rate, spec.fo, fmt, max_push=spec.nsamples, keep_dec=True) as rx:
        whole = _gpu_keys(rx.run(raw))
    monkeypatch.setenv("VDL2GPU_SPLIT_SAMPLES", "262144")     # a handicap of libvdl2gpu_test.so; the product library ignores it
    with _rx(spec.rate, spec.fo, fmt, max_push=spec.nsamples, keep_dec=True, testhooks=True) as rx:
        rx.push(raw)
        got = rx.poll()
        assert rx.stats()["samples_in"] == spec.nsamples
    assert _gpu_keys(got) == want == whole and len(want) >= 10


def test_empty_and_invalid_pushes(built):
    from vdlm2dec_amd import lib
    with _rx(2_000_000, [-50000], "cu8", max_push=4096) as rx:
        assert rx.L.vdl2gpu_push(rx.h, None, 0, 0, 0) == 0            # empty push is a no-op
        assert rx.L.vdl2gpu_push(rx.h, None, 10, 0, 0) == lib.load().vdl2gpu_push(rx.h, None, 10, 0, 0) == -1
        buf = np.zeros(2 * 8192, np.uint8)
        with pytest.raises(lib.Vdl2GpuError):
            rx.push(buf)                                              # larger than max_push
        assert rx.poll() == []
        assert rx.stats()["samples_in"] == 0


def test_multi_stream_batch_equals_single_streams(built, oracle):
    """nstreams > 1 (config 4 shape): independent wideband streams decoded side by side."""
    from vdlm2dec_amd.demod import Receiver, plan_channels
    specs = [S.eight_channels(seed=320 + i, dur=0.1) for i in range(3)]
    raws = [synth.synth_stream(sp, "cs16") for sp in specs]
    n = min(len(r) for r in raws)
    raw = np.stack([r[:n] for r in raws])
    plans = [plan_channels(S.FC, sp.fo) for sp in specs]
    with Receiver(2_000_000, plans, fmt="cs16", max_push=n // 2) as rx:
        got = rx.run(raw, block=40000)
    for s, sp in enumerate(specs):
        want = sorted(b.key() for b in oracle.run_oracle(raws[s][:n], "cs16", sp.rate, sp.fo, S.FC))
        mine = sorted((b.chn, b.nbrow, b.nlbyte, b.data) for b in got if b.stream == s)
        assert mine == want and len(want) >= 8


@pytest.mark.parametrize("nch,nstr", [(4, 2), (3, 3), (1, 5)])
def test_multi_stream_with_fewer_than_eight_channels(built, oracle, nch, nstr):
    """Several streams of fewer than 8 channels each, on the parallel path (pushes longer than the serial machine's
    4096 frames): the payload kernel addresses (stream, channel) slots as stream * 8 + channel like every other kernel
    (round 2's grid spanned nbch * nstreams slots and never decoded the channels of streams >= 1)."""
    from vdlm2dec_amd.demod import Receiver, plan_channels
    specs = [synth.random_scenario(2_000_000, S.FO8[:nch], 600_000, seed=500 + 10 * nch + i, bursts_per_s=40.0, info_max=60) for i in range(nstr)]
    raws = [synth.synth_stream(sp, "cs16") for sp in specs]
    raw = np.stack(raws)
    with Receiver(2_000_000, [plan_channels(S.FC, sp.fo) for sp in specs], fmt="cs16", max_push=300_000) as rx:
        got = rx.run(raw, block=300_000)      # 12 600 frames per push
        assert rx.stats()["serial_samples"] < 100_000
    total = 0
    for s, sp in enumerate(specs):
        want = sorted(b.key() for b in oracle.run_oracle(raws[s], "cs16", sp.rate, sp.fo, S.FC))
        mine = sorted((b.chn, b.nbrow, b.nlbyte, b.data) for b in got if b.stream == s)
        assert mine == want, s
        total += len(want)
    assert total >= 6 * nstr


def test_device_resident_input(built, oracle):
    import torch
    spec = S.eight_channels(seed=330)
    raw = synth.synth_stream(spec, "cs16")
    dev = torch.from_numpy(raw).to("cuda:0")
    with _rx(spec.rate, spec.fo, "cs16", max_push=spec.nsamples) as rx:
        rx.push_device(dev.data_ptr(), spec.nsamples)
        got = rx.poll()
    assert _gpu_keys(got) == sorted(b.key() for b in oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC))


def test_long_stream_properties(built, oracle):
    """Full-size property checks (BASELINE config 2 shape, 8 ch @ 2 MS/s, ~8 MS):
    every transmitted frame comes back CRC-clean, chunking does not matter, and a bounded
    oracle sample agrees bit-for-bit."""
    spec = synth.random_scenario(2_000_000, S.FO8, 1 << 23, seed=77, bursts_per_s=5.0, info_max=220)
    raw = synth.synth_stream(spec, "cs16")
    with _rx(spec.rate, spec.fo, "cs16", max_push=1 << 23) as rx:
        a = rx.run(raw)
    with _rx(spec.rate, spec.fo, "cs16", max_push=1 << 20) as rx:
        b = rx.run(raw, block=(1 << 20) - 12345)
    assert _gpu_keys(a) == _gpu_keys(b)
    sent = {(x.chan, synth.avlc_frame(x.info, src=(1 << 24) | (0x400000 + x.chan * 0x111 + len(x.info)))) for x in spec.bursts}
    got = {(c, f[1:-3]) for c, f in _frames(oracle, a)}
    assert len(sent - got) <= max(1, len(sent) // 50), (len(sent), len(sent - got))
    want = oracle.run_oracle(raw[: 2 * (1 << 21)], "cs16", spec.rate, spec.fo[:2], S.FC)
    with _rx(spec.rate, spec.fo[:2], "cs16", max_push=1 << 21) as rx:
        c = rx.run(raw[: 2 * (1 << 21)])
    assert _gpu_keys(c) == sorted(x.key() for x in want)


@pytest.mark.parametrize("mode", ["lazy", "full_scan", "serial"])
def test_scan_modes_agree_with_oracle(built, oracle, mode):
    """The three ways of finding sync triggers -- probe + regions + verify (default), all four
    sub-phases everywhere, and the plain serial machine -- must give identical bursts."""
    from vdlm2dec_amd.demod import Receiver, plan_channels
    spec = synth.random_scenario(2_000_000, S.FO8[:4], 1 << 21, seed=91, bursts_per_s=12.0, info_max=120)
    raw = synth.synth_stream(spec, "cs16")
    want = sorted(b.key() for b in oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC))
    with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cs16", max_push=1 << 20,
                  serial=(mode == "serial"), full_scan=(mode == "full_scan")) as rx:
        got = rx.run(raw, block=700_001)
        st = rx.stats()
    assert _gpu_keys(got) == want and len(want) >= 20
    assert st["serial_redos"] == 0
    if mode == "serial":
        assert st["candidates"] == 0


@pytest.mark.parametrize("scale", [1e18, 1e-17])
def test_screens_pass_what_they_cannot_judge(built, oracle, scale):
    """cf32 input far outside the range in which the scan's screens normalise the filter output:
    every instant must then go to the exact path (worklists overflow, tiles are done in pieces, the
    survivor list is flushed mid-tile) -- and the bursts are still the oracle's."""
    spec = S.regimes(seed=350, infos=(5, 40, 90))
    raw = synth.synth_stream(spec, "cf32").astype(np.float32) * np.float32(scale)
    assert np.isfinite(raw).all() and np.abs(raw).max() > 0
    want = sorted(b.key() for b in oracle.run_oracle(raw, "cf32", spec.rate, spec.fo, S.FC))
    with _rx(spec.rate, spec.fo, "cf32", max_push=1 << 20) as rx:
        got = rx.run(raw, block=300_000)
    assert _gpu_keys(got) == want and len(want) >= 3


def test_silence_between_signals(built, oracle):
    """Stretches of exact zeros (squelched or padded recordings): the filter output is exactly zero
    there, atan2f(0, 0) = 0, and the screens must treat that as the phase it is."""
    spec = S.regimes(seed=351, infos=(12, 64, 30, 75))
    raw = synth.synth_stream(spec, "cs16").copy()
    n = len(raw) // 2
    iq = raw.reshape(n, 2)
    iq[: n // 10] = 0
    iq[n // 2: n // 2 + n // 8] = 0          # may cut a burst short: whatever the oracle makes of it
    iq[-(n // 12):] = 0
    raw = iq.reshape(-1)
    want = sorted(b.key() for b in oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC))
    with _rx(spec.rate, spec.fo, "cs16", max_push=1 << 20) as rx:
        got = rx.run(raw, block=250_000)
        st = rx.stats()
    assert _gpu_keys(got) == want and len(want) >= 1


def test_verify_pass_catches_incomplete_tables(built, oracle, monkeypatch):
    """With the region scan switched off the candidate tables lack most trigger classes, so the
    resolver's chain is wrong after the first burst; K2a-verify must notice and -- with no repair round
    scheduled -- the serial redo must still deliver exactly the oracle's bursts."""
    from vdlm2dec_amd import lib
    from vdlm2dec_amd.demod import Receiver, plan_channels
    monkeypatch.setenv("VDL2GPU_REPAIR_ROUNDS", "0")
    spec = synth.random_scenario(2_000_000, S.FO8[:3], 1 << 21, seed=92, bursts_per_s=10.0, info_max=100)
    raw = synth.synth_stream(spec, "cs16")
    want = sorted(b.key() for b in oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC))
    with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cs16", max_push=1 << 20,
                  flags=lib.F_TEST_NOREGION) as rx:
        got = rx.run(raw, block=600_000)
        st = rx.stats()
    assert st["serial_redos"] > 0
    assert _gpu_keys(got) == want and len(want) >= 15


@pytest.mark.parametrize("rounds", [2, 3])
def test_repair_rounds_settle_what_the_verify_pass_finds(built, oracle, monkeypatch, rounds):
    """Same handicap, with two or three repair rounds from the first push: every event the verify pass finds joins the
    table as a candidate without a cluster and the resolver runs again (replaying those itself); with most classes
    missing the new chain fails its own verify pass, and the last scheduled round scans the channels that still fail
    completely, which leaves nothing to verify -- no channel may be left to the serial redo, and the result is the oracle's."""
    from vdlm2dec_amd import lib
    from vdlm2dec_amd.demod import Receiver, plan_channels
    monkeypatch.setenv("VDL2GPU_REPAIR_ROUNDS", str(rounds))
    spec = synth.random_scenario(2_000_000, S.FO8[:3], 1 << 21, seed=95, bursts_per_s=10.0, info_max=100)
    raw = synth.synth_stream(spec, "cs16")
    want = sorted(b.key() for b in oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC))
    with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cs16", max_push=1 << 20,
                  flags=lib.F_TEST_NOREGION) as rx:
        got = rx.run(raw, block=600_000)
        st = rx.stats()
    assert _gpu_keys(got) == want and len(want) >= 15
    assert st["serial_redos"] == 0


@pytest.mark.parametrize("seed,bps", [(1077, 4.0), (5077, 4.0), (1077, 15.0)])
def test_an_event_in_one_class_only_costs_a_resolver_round_not_a_serial_redo(built, oracle, seed, bps):
    """Recordings of bench.py's generator in which a noise trigger exists in ONE timing class only, far from anything the
    probe's class sees (scripts/dev/unseeded.py: about one per 100 channel-seconds) -- and the chain happens to idle in
    that class there.  The verify pass finds it; the one repair round that is always scheduled re-resolves the channel
    with it (round 3 before this test: a serial redo of the channel's whole push, 8-10 ms for the first push of a handle).
    Defaults throughout: no environment knobs, no test build.  (The second half of the push is another recording: the same
    one twice holds the same event twice, the second behind the repair of the first -- two rounds' work.)"""
    import bench
    from vdlm2dec_amd.demod import Receiver, plan_channels
    spec, raw = bench.make_tile(seed, "cs16", 2_000_000, S.FO8, bps)
    big = np.concatenate([raw, bench.make_tile(2077, "cs16", 2_000_000, S.FO8, bps)[1]])
    want = sorted(b.key() for b in oracle.run_oracle(big, "cs16", spec.rate, spec.fo, S.FC))
    with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cs16", max_push=big.size // 2) as rx:
        rx.push(big)
        got = rx.poll()
        st = rx.stats()
    assert _gpu_keys(got) == want and len(want) >= 60
    assert st["repairs"] >= 1 and st["serial_redos"] == 0, st


@pytest.mark.timeout(300)
@pytest.mark.parametrize("drop", [1, 2, 3])
def test_resolver_builds_the_clusters_it_is_not_given(built, oracle, monkeypatch, drop):
    """K2s withholds every drop-th candidate from K2b (VDL2GPU_PRIM_DROP): the resolver must replay
    those stretches itself and end up where the oracle does -- which cluster is precomputed is a cost
    decision, never a correctness one.  (drop=1: no cluster at all; the second trigger of a push used
    to spin in this path.)"""
    from vdlm2dec_amd.demod import Receiver, plan_channels
    monkeypatch.setenv("VDL2GPU_PRIM_DROP", str(drop))
    spec = synth.random_scenario(2_000_000, S.FO8[:4], 1 << 21, seed=97 + drop, bursts_per_s=12.0, info_max=90)
    raw = synth.synth_stream(spec, "cs16")
    want = sorted(b.key() for b in oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC))
    with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cs16", max_push=1 << 20, testhooks=True) as rx:
        got = rx.run(raw, block=700_000)
    assert _gpu_keys(got) == want and len(want) >= 20


def test_candidate_tables_overflow_falls_back_to_the_serial_machine(built, oracle, monkeypatch):
    """More than 6144 trigger candidates of one channel in one part (950 short bursts in 15 s of air time, the part
    length pinned to the whole push through the test build's VDL2GPU_SPLIT_SAMPLES): the parallel tables are unusable
    for that push and the channel is handled by the serial machine -- slower, and still the oracle's bursts; the next,
    ordinary push goes through the tables again."""
    from vdlm2dec_amd.demod import Receiver, plan_channels
    n = 30_000_000
    spec = synth.random_scenario(2_000_000, S.FO8[:1], n, seed=7, bursts_per_s=200.0, info_max=4)
    raw = synth.synth_stream(spec, "cs16")
    want = sorted(b.key() for b in oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC))
    assert len(want) >= 900         # x 8 classes: more than the tables' 6144 candidates
    monkeypatch.setenv("VDL2GPU_SPLIT_SAMPLES", str(n))
    with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cs16", max_push=n, testhooks=True) as rx:
        got = rx.run(raw, block=n)
        st = rx.stats()
        assert _gpu_keys(got) == want
        assert st["serial_samples"] > 600_000 and st["overflowed"] == 0     # most of the push's 1 260 000 decimated samples
        s0 = st["serial_samples"]
        rx.push(raw[:4_000_000])        # 2 M samples: far below the tables' capacity
        rx.poll()
        assert rx.stats()["serial_samples"] - s0 < 100_000


def test_dense_pushes_never_reach_the_serial_machine(built, oracle):
    """The same dense recording (50 bursts a second on one channel: 8500 trigger candidates in a 10 s push, twice what
    the tables hold) pushed whole again and again through the PRODUCT library: the first pushes are cut into 8.4 s
    parts, the later ones into parts sized by the measured candidate density -- no push touches the serial machine,
    and every burst is the oracle's."""
    from vdlm2dec_amd.demod import Receiver, plan_channels
    n, reps = 20_000_000, 5
    spec = synth.random_scenario(2_000_000, S.FO8[:1], n, seed=7, bursts_per_s=110.0, info_max=4)
    raw = synth.synth_stream(spec, "cs16")
    want = sorted(b.key() for b in oracle.run_oracle(np.tile(raw, reps), "cs16", spec.rate, spec.fo, S.FC))
    with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cs16", max_push=n) as rx:
        got, serial = [], []
        for _ in range(reps):
            rx.push(raw)
            got += rx.poll()
            serial.append(rx.stats()["serial_samples"])
    assert _gpu_keys(got) == want
    assert serial[-1] < 100_000, serial                     # 840 000 decimated samples per push: none of them serial


def test_a_sudden_load_overflows_once_and_the_parts_adapt(built, oracle):
    """Quiet pushes first (the parts grow to the 36 s limit), then the dense recording: its first push overflows the
    tables (serial machine for that push: exact, counted), the following ones are cut by the density it showed and go
    through the tables."""
    from vdlm2dec_amd.demod import Receiver, plan_channels
    n = 30_000_000
    dense = synth.random_scenario(2_000_000, S.FO8[:1], n, seed=7, bursts_per_s=200.0, info_max=4)      # 949 bursts x 8 classes > 6144 candidates
    quiet = synth.random_scenario(2_000_000, S.FO8[:1], n, seed=8, bursts_per_s=1.0, info_max=40)
    rd, rq = synth.synth_stream(dense, "cs16"), synth.synth_stream(quiet, "cs16")
    seq = [rq] * 10 + [rd] * 4          # ten quiet pushes: the density window (four parts) has forgotten the start-up parts
    want = sorted(b.key() for b in oracle.run_oracle(np.concatenate(seq), "cs16", 2_000_000, dense.fo, S.FC))
    with Receiver(2_000_000, plan_channels(S.FC, dense.fo), fmt="cs16", max_push=n) as rx:
        got, serial = [], []
        for r in seq:
            rx.push(r)
            got += rx.poll()
            serial.append(rx.stats()["serial_samples"])
    assert _gpu_keys(got) == want
    assert serial[9] < 100_000                              # quiet: tables
    assert serial[10] - serial[9] > 600_000                 # the first dense push: one part, overflow, serial machine
    assert serial[13] - serial[12] < 100_000                # two pushes later the parts fit


def test_pipelined_polling_delivers_everything_once(built, oracle):
    """vdl2gpu_poll_ready (non-blocking) + a final vdl2gpu_poll: two pushes in flight, every burst
    handed out exactly once, in stream-time order per poll."""
    from vdlm2dec_amd import lib
    from vdlm2dec_amd.demod import Burst
    spec = synth.random_scenario(2_000_000, S.FO8[:4], 1 << 21, seed=93, bursts_per_s=12.0, info_max=80)
    raw = synth.synth_stream(spec, "cu8")
    want = sorted(b.key() for b in oracle.run_oracle(raw, "cu8", spec.rate, spec.fo, S.FC))
    buf = (lib.BurstT * 4096)()
    got = []
    with _rx(spec.rate, spec.fo, "cu8", max_push=1 << 18) as rx:
        blk = 1 << 18
        for s0 in range(0, spec.nsamples, blk):
            rx.push(raw[2 * s0:2 * (s0 + blk)])
            n = rx.poll_ready_raw(buf, 4096)
            got += [(buf[i].chn, buf[i].nbrow, buf[i].nlbyte, bytes(buf[i].data)) for i in range(n)]
        n = rx.poll_raw(buf, 4096)
        got += [(buf[i].chn, buf[i].nbrow, buf[i].nlbyte, bytes(buf[i].data)) for i in range(n)]
    assert sorted(got) == want and len(want) >= 20


def test_many_streams_batch(built, oracle):
    """More than 8 streams (= more than 64 channel slots) in one handle: config-4 shape."""
    from vdlm2dec_amd.demod import Receiver, plan_channels
    specs = [S.eight_channels(seed=400 + i, dur=0.06, info=(3, 9, 5, 12, 7, 4, 6, 8)) for i in range(10)]
    raws = [synth.synth_stream(sp, "cs16") for sp in specs]
    n = min(len(r) for r in raws)
    raw = np.stack([r[:n] for r in raws])
    with Receiver(2_000_000, [plan_channels(S.FC, sp.fo) for sp in specs], fmt="cs16", max_push=n // 2) as rx:
        got = rx.run(raw)
    for s in (0, 7, 8, 9):
        want = sorted(b.key() for b in oracle.run_oracle(raws[s][:n], "cs16", specs[s].rate, specs[s].fo, S.FC))
        mine = sorted((b.chn, b.nbrow, b.nlbyte, b.data) for b in got if b.stream == s)
        assert mine == want and len(want) >= 6, s


def test_three_streams_ragged_pushes_in_the_pipeline(built, oracle):
    """Three streams of eight channels, several pushes in the pipeline, ragged push lengths, bursts collected as they become ready.
    (Round 5 ran this scenario under four opt-in arrangements of the pipeline -- stream groups, reachable-class clusters, the cluster
    kernel in the front stage, CU-masked streams; none of them was faster, and round 6 took them out of the library: docs/HISTORY.md.)"""
    from vdlm2dec_amd.demod import Receiver, plan_channels
    specs = [synth.random_scenario(2_000_000, S.FO8, 1 << 21, seed=700 + i, bursts_per_s=12.0, info_max=120) for i in range(3)]
    raws = [synth.synth_stream(sp, "cs16") for sp in specs]
    raw = np.stack(raws)
    got = []
    with Receiver(2_000_000, [plan_channels(S.FC, sp.fo) for sp in specs], fmt="cs16", max_push=1 << 20) as rx:
        pos = 0
        for k in (600_000, 400_001, 500_000, 597_151):
            rx.push(raw[:, 2 * pos:2 * (pos + k)])
            got += rx.poll_ready()
            pos += k
        got += rx.poll()
    for s in range(3):
        want = sorted(b.key() for b in oracle.run_oracle(raws[s], "cs16", specs[s].rate, specs[s].fo, S.FC))
        mine = sorted((b.chn, b.nbrow, b.nlbyte, b.data) for b in got if b.stream == s)
        assert mine == want and len(want) >= 20, s


def test_more_than_512_channel_slots_take_the_repaired_selection(built, oracle):
    """A handle with more than 64 streams has channel slots the 16-word repair mask does not reach (slot = stream * 8 + channel):
    the payload decode there is one pass behind the commit and must still take the selection a repair round re-resolved, and
    nothing of the resolver's for a channel K2f redid serially (round 4's build decoded the stale first selection for slots >= 512:
    ADVICE r4).  66 streams of one channel each, region scan dropped so that nearly every burst needs a repair; streams 0, 63, 64,
    65 against the oracle."""
    from vdlm2dec_amd import lib
    from vdlm2dec_amd.demod import Receiver, plan_channels
    spec = synth.random_scenario(2_000_000, S.FO8[:1], 1 << 20, seed=93, bursts_per_s=25.0, info_max=60)
    raw1 = synth.synth_stream(spec, "cs16")
    want = sorted(b.key() for b in oracle.run_oracle(raw1, "cs16", spec.rate, spec.fo, S.FC))
    nstr = 66
    raw = np.stack([raw1] * nstr)
    with Receiver(spec.rate, [plan_channels(S.FC, spec.fo)] * nstr, fmt="cs16", max_push=1 << 19, flags=lib.F_TEST_NOREGION) as rx:
        got = rx.run(raw)
        st = rx.stats()
    assert st["repairs"] + st["serial_redos"] > 0 and len(want) >= 8
    for s in (0, 63, 64, 65):
        mine = sorted((b.chn, b.nbrow, b.nlbyte, b.data) for b in got if b.stream == s)
        assert mine == want, s


def test_ingest_ring_equals_push(built, oracle):
    """blocks written in place into the page-locked ring (more blocks than slots, so slots
    are reused while earlier copies and kernels are still in flight, one dropped block in between) decode
    to exactly the oracle's bursts."""
    from vdlm2dec_amd.demod import Receiver, plan_channels
    spec = synth.random_scenario(2_000_000, S.FO8[:4], 3_000_000, seed=77, bursts_per_s=20.0, info_max=120)
    raw = synth.synth_stream(spec, "cu8")
    want = sorted(b.key() for b in oracle.run_oracle(raw, "cu8", spec.rate, spec.fo, S.FC))
    blk = 32768   # the reference's hand-off size, vdlm2.h:35
    with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cu8", max_push=blk) as rx:
        rx.ring_init(blk, nslots=3)
        got = []
        rawb = raw.view(np.uint8).reshape(-1)
        for i, s0 in enumerate(range(0, spec.nsamples, blk)):
            n = min(blk, spec.nsamples - s0)
            slot = rx.ring_acquire()
            slot[0, :2 * n] = rawb[2 * s0:2 * (s0 + n)]
            rx.ring_commit(n)
            if i == 5:   # a short USB read: the producer gives the slot back empty
                rx.ring_acquire()
                rx.ring_commit(0)
            if i % 16 == 15:
                got += rx.poll_ready()
        got += rx.poll()
    assert _gpu_keys(got) == want and len(want) >= 20


def test_ingest_ring_multi_stream(built, oracle):
    """Ring slots hold one run of samples per stream, `stride` bytes apart."""
    from vdlm2dec_amd.demod import Receiver, plan_channels
    specs = [S.eight_channels(seed=520 + i, dur=0.1) for i in range(2)]
    raws = [synth.synth_stream(sp, "cs16") for sp in specs]
    n = min(len(r) for r in raws) // 2      # complex samples
    plans = [plan_channels(S.FC, sp.fo) for sp in specs]
    blk = 50000
    with Receiver(2_000_000, plans, fmt="cs16", max_push=blk) as rx:
        rx.ring_init(blk, nslots=2)
        got = []
        for s0 in range(0, n, blk):
            m = min(blk, n - s0)
            slot = rx.ring_acquire()
            for s, r in enumerate(raws):
                slot[s, :4 * m] = r[2 * s0:2 * (s0 + m)].view(np.uint8)
            rx.ring_commit(m)
            got += rx.poll_ready()
        got += rx.poll()
    for s, sp in enumerate(specs):
        want = sorted(b.key() for b in oracle.run_oracle(raws[s][:2 * n], "cs16", sp.rate, sp.fo, S.FC))
        mine = sorted((b.chn, b.nbrow, b.nlbyte, b.data) for b in got if b.stream == s)
        assert mine == want and len(want) >= 8


@pytest.mark.timeout(600)
@pytest.mark.parametrize("bps", [15.0, 30.0])
def test_busy_channels_stay_on_the_parallel_path(built, oracle, bps):
    """BASELINE configs[1] with 15 and 30 bursts a second and channel offered (at 30 the channels are saturated), through
    bench.py's own leg: every burst of the run -- the first push of the handle included -- equals the oracle's, no push
    takes longer than 5 ms (round 2: 57 ms for the first one, through the serial machine), and less than 1 % of the
    decimated samples go through the serial machine."""
    import bench
    from vdlm2dec_amd import synth as sy
    with _rx(2_000_000, S.FO8, "cs16", max_push=1 << 20) as rx:     # (a cold process loads the kernels' code objects on their
        rx.push(np.zeros(2 << 20, np.int16))                        #  first launch: not what the first push of a HANDLE costs)
        rx.poll()
    r = bench.run_leg("busy", "test", 0, 2_000_000, sy.DEFAULT_FO_8CH, "cs16", 1, 16, bps, steps=4, warmup=2, seed0=77, repeats=1)
    assert r["parity"]["equal"] and r["parity"]["bursts_checked"] > 10_000
    assert r["serial_samples_frac"] < 0.01 and r["overflowed"] == 0
    assert r["first_push_ms"] < 5.0 and r["max_push_ms"] < 5.0, r
    assert r["value"] > 20_000          # MS/s; the measured figure is in the bench line (configs.config2_busy_*)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("nstr,ntiles,bps,steps", [(2, 8, 8.0, 5), (1, 4, 30.0, 7), (1, 16, 2.0, 4)])
def test_three_pushes_in_the_pipeline_equal_the_oracle(built, oracle, nstr, ntiles, bps, steps):
    """Device-resident pushes back to back, bursts taken as they become ready (bench.py's leg: what scripts/soak_pipeline.py
    draws at random): the front stage of one push beside the back stage of the one before and the tail of the one before
    that, on three plane / table sets, three output rings and four slabs -- every burst from the first sample on equals the
    oracle's, whatever the load and the push length, and nothing is left to the serial redo."""
    import bench
    from vdlm2dec_amd import synth as sy
    r = bench.run_leg("pipeline", "test", 0, 2_000_000, sy.DEFAULT_FO_8CH, "cs16", nstr, ntiles, bps, steps=steps, warmup=2, seed0=4321, repeats=1)
    assert r["parity"]["equal"] and r["parity"]["bursts_checked"] > 1000, r["parity"]
    assert r["serial_redos"] == 0 and r["overflowed"] == 0, r


def test_paced_live_ring_equals_the_oracle(built, oracle):
    """BASELINE configs[4] through bench.py's own leg: 32768-sample cu8 blocks (one RTL-SDR USB transfer) written in place
    into the 8-slot page-locked ring every 16.384 ms, bursts collected block by block -- the bursts are the oracle's, and
    a block's bursts are on the host long before the next block is due."""
    import bench
    r = bench.live_leg(0, nblocks=90, paced=True)
    assert r["parity"]["equal"] and r["parity"]["bursts_checked"] >= 30
    lat = r["latency_ms"]
    assert lat["p50"] < 2.0 and lat["p99"] < 16.384 and r["blocks_started_late"] == 0, r
    r2 = bench.live_leg(0, nblocks=90, paced=False, seed=78)      # as fast as the producer can: the same bursts, no pacing to hide behind
    assert r2["parity"]["equal"]


def test_records_beyond_the_slab_and_uncollected_records_survive(built, oracle, monkeypatch):
    """Collecting: the GPU writes a push's records into a slab of page-locked host memory (k_export_records); what
    exceeds the slab comes through the bounce buffer, and what the caller has not taken when the slab is due again (two
    pushes later) moves to the pageable queue.  Slab pinned to 5 records (test build), several pushes without polling:
    every burst once, in order, equal to the oracle's."""
    from vdlm2dec_amd.demod import Receiver, plan_channels
    monkeypatch.setenv("VDL2GPU_SLAB_CAP", "5")
    spec = synth.random_scenario(2_000_000, S.FO8[:4], 4_000_000, seed=901, bursts_per_s=30.0, info_max=60)
    raw = synth.synth_stream(spec, "cs16")
    want = sorted(b.key() for b in oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC))
    assert len(want) >= 60
    with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cs16", max_push=1 << 20, testhooks=True) as rx:
        n = spec.nsamples
        got = []
        for i, s0 in enumerate(range(0, n, 500_000)):       # eight pushes (21 000 frames each: the parallel path), ~10 bursts a push
            rx.push(raw[2 * s0:2 * min(n, s0 + 500_000)])
            if i == 5:
                got += rx.poll_ready()                       # once in the middle, never waiting
        got += rx.poll()
        assert rx.stats()["overflowed"] == 0
    assert _gpu_keys(got) == want
    ends = [(b.end_dec, b.stream, b.chn) for b in got]
    assert ends == sorted(ends)                              # hand-out order: (end, stream, channel)


@pytest.mark.parametrize("seed", [11, 12])
def test_long_and_short_pushes_in_any_order(built, oracle, seed):
    """Pushes long enough for the parallel path (three stages on three streams, their tails on the payload stream) and pushes
    short enough for the serial one (everything on the main stream), in random order, bursts taken as they become ready:
    a short push follows the long one's tail, a long one the short one's commit -- the bursts are the oracle's whatever
    the order (the transitions between the two paths are event waits nothing else exercises)."""
    from vdlm2dec_amd.demod import Receiver, plan_channels
    rng = np.random.default_rng(seed)
    spec = synth.random_scenario(2_000_000, S.FO8[:4], 9_000_000, seed=500 + seed, bursts_per_s=25.0, info_max=120)
    raw = synth.synth_stream(spec, "cs16")
    want = sorted(b.key() for b in oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC))
    assert len(want) >= 200
    got = []
    with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cs16", max_push=1 << 20) as rx:
        pos, n = 0, spec.nsamples
        while pos < n:
            k = int(rng.choice([700_000, 1_000_000, 40_000, 8_192, 131_072, 1 << 20]))     # >= 97 524 samples is the parallel path
            k = min(k, n - pos)
            rx.push(raw[2 * pos:2 * (pos + k)])
            pos += k
            if rng.integers(0, 3):
                got += rx.poll_ready()
        got += rx.poll()
        st = rx.stats()
    assert _gpu_keys(got) == want
    assert st["overflowed"] == 0


def test_a_device_buffer_is_free_once_two_more_pushes_have_been_issued(built, oracle):
    """include/vdl2gpu.h: a VDL2GPU_MEM_DEVICE buffer is read in place, asynchronously, and must stay unchanged "until the
    second push after this one has been issued".  Three buffers in turn, each overwritten with noise the moment the call that
    issues the second push after it has returned -- with three pushes in the pipeline that call is what waits for the
    buffer's channeliser (an event behind every channeliser)."""
    import torch
    from vdlm2dec_amd.demod import Receiver, plan_channels
    spec = synth.random_scenario(2_000_000, S.FO8[:4], 8_000_000, seed=777, bursts_per_s=20.0, info_max=100)
    raw = synth.synth_stream(spec, "cs16")
    want = sorted(b.key() for b in oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC))
    assert len(want) >= 150
    n, blk = spec.nsamples, 1_000_000
    bufs = [torch.empty(2 * blk, dtype=torch.int16, device="cuda") for _ in range(3)]
    noise = torch.randint(-3000, 3000, (2 * blk,), dtype=torch.int16, device="cuda")
    side = torch.cuda.Stream()      # the overwrites must not be ordered behind the library's work by accident
    got = []
    with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cs16", max_push=blk) as rx:
        for i, s0 in enumerate(range(0, n, blk)):
            k = min(blk, n - s0)
            b = bufs[i % 3]
            b[:2 * k].copy_(torch.from_numpy(raw[2 * s0:2 * (s0 + k)]).cuda())
            torch.cuda.synchronize()
            rx.push_device(b.data_ptr(), k, 0)
            if i >= 2:              # the push two back: its buffer is ours again
                with torch.cuda.stream(side):
                    bufs[(i - 2) % 3].copy_(noise)
                side.synchronize()
            got += rx.poll_ready()
        got += rx.poll()
    assert _gpu_keys(got) == want


@pytest.mark.timeout(600)
@pytest.mark.parametrize("hooks", [{"VDL2GPU_TEST_ITEM_GRID": "3"}, {"VDL2GPU_TEST_ITEM_GRID": "3", "VDL2GPU_TEST_ITEM_COMMON": "200"}])
def test_item_list_common_area_and_overflow(built, oracle, monkeypatch, hooks):
    """The scans hand what passes their first screen to k2x_second through per-workgroup areas of an item list.  With three scan
    workgroups per channel and the smallest private areas (test build: VDL2GPU_TEST_ITEM_GRID) nearly every item takes the
    other path -- the common area behind the private ones, one device-scope atomic per wavefront and pass; with a common area of
    200 items (VDL2GPU_TEST_ITEM_COMMON) the list overflows: the channel's tables count as unusable, k2x_second reads nothing
    past the first refused group, and the serial machine decodes the channel.  Either way: the oracle's bursts."""
    from vdlm2dec_amd.demod import Receiver, plan_channels
    for k, v in hooks.items():
        monkeypatch.setenv(k, v)
    spec = synth.random_scenario(2_000_000, S.FO8[:3], 6_000_000, seed=4242, bursts_per_s=10.0, info_max=120)
    raw = synth.synth_stream(spec, "cs16")
    want = sorted(b.key() for b in oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC))
    with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cs16", max_push=3_000_000, testhooks=True) as rx:
        got = rx.run(raw, block=3_000_000)
        st = rx.stats()
    assert _gpu_keys(got) == want and len(want) >= 60
    if "VDL2GPU_TEST_ITEM_COMMON" in hooks:
        assert st["serial_samples"] > 100_000, st       # the overflow was noticed and the serial machine took over
    else:
        assert st["serial_samples"] < 20_000, st        # the common area is an ordinary path: nothing fell back


@pytest.mark.timeout(600)
@pytest.mark.parametrize("rounds", [1, 2])
def test_local_repair_does_not_stand_on_a_verify_pass_whose_list_overflowed(built, oracle, monkeypatch, rounds):
    """Region scan dropped (every burst is an event the verify pass has to find) AND the verify passes' item lists cut down to three
    small private areas and a common area of 200 items (test build): the list refuses items, the pass cannot vouch for the channel
    (fail = 0) although it has listed the events it got to.  The local repair (k2p_patch) verifies only the stretches IT changes: it
    must leave such a channel to a round that resolves it from its input state, or to the serial redo -- round 6's first version
    repaired it locally and called it verified (scripts/soak.py seeds 6064, 6068, 6077 on handles with small item lists)."""
    from vdlm2dec_amd import lib as _lib
    from vdlm2dec_amd.demod import Receiver, plan_channels
    monkeypatch.setenv("VDL2GPU_REPAIR_ROUNDS", str(rounds))
    monkeypatch.setenv("VDL2GPU_TEST_ITEM_VERIFY", "1")
    monkeypatch.setenv("VDL2GPU_TEST_ITEM_GRID", "3")
    monkeypatch.setenv("VDL2GPU_TEST_ITEM_COMMON", "200")
    spec = synth.random_scenario(2_000_000, S.FO8[:3], 7_000_000, seed=3131 + rounds, bursts_per_s=14.0, info_max=400)
    raw = synth.synth_stream(spec, "cs16")
    want = sorted(b.key() for b in oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC))
    with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cs16", max_push=1_000_000, flags=_lib.F_TEST_NOREGION, testhooks=True) as rx:
        got = rx.run(raw, block=1_000_000)
        st = rx.stats()
    assert _gpu_keys(got) == want and len(want) >= 60
    assert st["repairs"] >= 5, st


@pytest.mark.timeout(600)
@pytest.mark.parametrize("rounds", [1, 2, 3])
def test_superseded_repair_selections_leave_no_records_behind(built, oracle, monkeypatch, rounds):
    """Region scan dropped (test build), so the verify pass finds an unlisted event behind nearly every burst and channels go
    through several repair rounds and serial redos per push, over many short pushes (the output rings are reused every third
    push).  A repair round's selection can be superseded by the next round's or by the serial redo: records must be reserved
    for the FINAL selection only -- reserved-and-never-written records once handed out whatever an earlier push had left in the
    ring (scripts/soak.py seeds 1006..1051 of round 4).  Every burst the oracle's, none extra."""
    from vdlm2dec_amd import lib as _lib
    from vdlm2dec_amd.demod import Receiver, plan_channels
    monkeypatch.setenv("VDL2GPU_REPAIR_ROUNDS", str(rounds))
    spec = synth.random_scenario(2_000_000, S.FO8[:3], 7_000_000, seed=1019 + rounds, bursts_per_s=14.0, info_max=120)
    raw = synth.synth_stream(spec, "cs16")
    want = sorted(b.key() for b in oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC))
    with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cs16", max_push=400_000, flags=_lib.F_TEST_NOREGION, testhooks=True) as rx:
        got = rx.run(raw, block=400_000)
        st = rx.stats()
    assert _gpu_keys(got) == want and len(want) >= 80
    assert st["repairs"] >= 5, st


def test_stage_dump_is_a_gantt_chart_of_the_pipeline(built, oracle, tmp_path):
    """VDL2GPU_STAGE_DUMP=1: every push prints where its stages began and ended on the GPU's clock (the pipeline's Gantt chart
    without a profiler, scripts/dev/stage_gantt.py) -- ordered in time within a push, and the bursts are the oracle's all the same."""
    import re
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import scenarios as S\n"
        "from vdlm2dec_amd import synth\n"
        "from vdlm2dec_amd.demod import Receiver, plan_channels\n"
        "spec = synth.random_scenario(2_000_000, S.FO8, 3 << 21, seed=91, bursts_per_s=6.0, info_max=120)\n"
        "raw = synth.synth_stream(spec, 'cs16')\n"
        "with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt='cs16', max_push=1 << 21) as rx:\n"
        "    got = rx.run(raw, block=1 << 21)\n"
        "    rx.sync()\n"
        "print('KEYS', sorted((b.chn, b.nbrow, b.nlbyte, bytes(b.data).hex()) for b in got))\n"
    ) % (os.path.dirname(HERE), HERE)
    env = dict(os.environ, VDL2GPU_STAGE_DUMP="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [dict((int(k), float(v)) for k, v in re.findall(r"e(\d+)=(-?[\d.]+)", ln))
            for ln in r.stderr.splitlines() if ln.startswith("vdl2gpu stage dump push")]
    assert len(rows) >= 3, r.stderr[-2000:]
    for e in rows:
        chain = [e[k] for k in (0, 1, 10, 4, 2, 13, 14, 12, 15, 5, 6, 7)]
        assert all(t >= 0.0 for t in chain), e
        assert chain == sorted(chain), e                 # a push's stages follow each other
    spec = synth.random_scenario(2_000_000, S.FO8, 3 << 21, seed=91, bursts_per_s=6.0, info_max=120)
    raw = synth.synth_stream(spec, "cs16")
    want = sorted(k[:3] + (bytes(k[3]).hex(),) for k in (b.key() for b in oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC)))
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("KEYS ")][0]
    assert eval(line[5:]) == want
