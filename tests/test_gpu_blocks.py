"""Block path on the GPU: vdl2gpu_decode_blocks == the reference's blk_thread
(vdlm2.c:84-161, rs.c, crc.c) as restated by the oracle, frame for frame, byte for byte."""
import json
import glob
import os

import numpy as np
import pytest

import scenarios as S
from vdlm2dec_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _rx():
    from vdlm2dec_amd.demod import Receiver, plan_channels
    return Receiver(2_000_000, plan_channels(S.FC, [-50000]), fmt="cu8", max_push=4096)


def _want(oracle, blocks):
    out = []
    for i, (nbrow, nlbyte, data) in enumerate(blocks):
        out += [(i, f) for f in oracle.frames_of_block(nbrow, nlbyte, data)]
    return out


def test_golden_blocks_give_the_reference_frames(built, oracle):
    """msgblk_t records of the real reference (tests/golden) -> the frames its own out() received."""
    blocks, frames = [], []
    for path in sorted(glob.glob(os.path.join(HERE, "golden", "*.json"))):
        meta = json.load(open(path))
        for c in meta["channels"]:
            for b in c["blocks"]:
                blocks.append((b["nbrow"], b["nlbyte"], bytes.fromhex(b["data"])))
            frames += c["frames"]
    with _rx() as rx:
        got = rx.decode_blocks(blocks)
    assert [f.hex() for _, f in got] == frames and len(frames) >= 20
    assert got == _want(oracle, blocks)


def _tx_block(rng, info_len, nerr_rows=()):
    """One transmitted burst as the demodulator would hand it over, optionally with byte errors."""
    info = bytes(rng.integers(0, 256, info_len, dtype=np.uint8).tolist())
    nbrow, nlbyte, data = synth.received_rows(synth.hdlc_payload(synth.avlc_frame(info)))
    data = bytearray(data)
    for r, nerr in nerr_rows:
        if r < nbrow:
            span = 249 if r < nbrow - 1 else max(nerr, nlbyte)
            for pos in rng.choice(span, size=nerr, replace=False):
                data[r * 255 + int(pos)] ^= int(rng.integers(1, 256))
    return nbrow, nlbyte, bytes(data)


def test_corrected_uncorrectable_and_random_rows(built, oracle):
    """Everything rs() can meet: clean rows, 1..3 byte errors (corrected), 4+ (miscorrected or given
    up half way -- the partially applied corrections must agree too), every FEC-shortening regime,
    and rows of pure noise."""
    rng = np.random.default_rng(77)
    blocks = []
    for n in (1, 2, 3, 10, 28, 31, 60, 66, 70, 120, 247, 250, 400, 497, 900, 1500, 1900):
        blocks.append(_tx_block(rng, n))
        for nerr in (1, 2, 3, 4, 5, 7):
            blocks.append(_tx_block(rng, n, [(0, nerr)]))
            blocks.append(_tx_block(rng, n, [(r, nerr) for r in range(8)]))
    for _ in range(200):
        nbrow = int(rng.integers(1, 9))
        blocks.append((nbrow, int(rng.integers(0, 250)), bytes(rng.integers(0, 256, 8 * 255, dtype=np.uint8).tolist())))
    for fill in (0x00, 0xff, 0x7e, 0x3f, 0xfc):       # degenerate streams: all stuffing / all flags
        blocks.append((8, 249, bytes([fill]) * (8 * 255)))
    want = _want(oracle, blocks)
    with _rx() as rx:
        got = rx.decode_blocks(blocks)
    assert got == want
    assert len(want) >= 80


def test_decoded_stream_to_frames_end_to_end(built, oracle):
    """IQ -> bursts (demodulator kernels) -> frames (block kernel), all on the GPU, against
    IQ -> oracle demodulator -> oracle block path."""
    from vdlm2dec_amd.demod import Receiver, plan_channels
    spec = S.eight_channels(seed=340)
    raw = synth.synth_stream(spec, "cs16")
    ob = sorted(oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC), key=lambda b: (b.chn, b.end_dec))
    want = [(b.chn, f) for b in ob for f in oracle.frames_of_block(b.nbrow, b.nlbyte, b.data)]
    with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cs16", max_push=1 << 21) as rx:
        bursts = sorted(rx.run(raw), key=lambda b: (b.chn, b.end_dec))
        got = [(bursts[i].chn, f) for i, f in rx.decode_blocks(bursts)]
    assert got == want and len(want) >= 8


def test_block_path_in_the_pipeline(built, oracle):
    """frames=True: the block kernel runs on every push's records in device memory; frames arrive
    through vdl2gpu_poll_frames, the bursts are still there for vdl2gpu_poll."""
    from vdlm2dec_amd.demod import Receiver, plan_channels
    spec = S.eight_channels(seed=341)
    raw = synth.synth_stream(spec, "cs16")
    ob = sorted(oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC), key=lambda b: (b.end_dec, b.chn))
    want = [(0, b.chn, f) for b in ob for f in oracle.frames_of_block(b.nbrow, b.nlbyte, b.data)]
    with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cs16", max_push=1 << 19, frames=True) as rx:
        bursts, frames = [], []
        per = 4
        for s0 in range(0, spec.nsamples, 1 << 19):
            rx.push(raw[2 * s0:2 * (s0 + (1 << 19))])
            frames += rx.poll_frames()
            bursts += rx.poll()
    assert sorted(frames) == sorted(want) and len(want) >= 8
    assert sorted((b.chn, b.nbrow, b.nlbyte, b.data) for b in bursts) == sorted(b.key() for b in ob)


def test_frames_of_channels_redone_serially(built, oracle):
    """The payload gather runs ahead of the verify pass; when that pass fails a channel, the serial redo
    makes the channel's records again and the ones made ahead are void.  The block kernel must skip
    them on the device (the records themselves are filtered on the host): every frame exactly once."""
    from vdlm2dec_amd import lib
    from vdlm2dec_amd.demod import Receiver, plan_channels
    spec = synth.random_scenario(2_000_000, S.FO8[:3], 1 << 21, seed=92, bursts_per_s=10.0, info_max=100)
    raw = synth.synth_stream(spec, "cs16")
    ob = oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC)
    want = sorted((0, b.chn, f) for b in ob for f in oracle.frames_of_block(b.nbrow, b.nlbyte, b.data))
    with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cs16", max_push=1 << 20, frames=True,
                  flags=lib.F_TEST_NOREGION) as rx:
        bursts, frames = [], []
        for s0 in range(0, spec.nsamples, 600_000):
            rx.push(raw[2 * s0:2 * (s0 + 600_000)])
            frames += rx.poll_frames()
            bursts += rx.poll()
        st = rx.stats()
    assert st["serial_redos"] > 0
    assert sorted(frames) == want and len(want) >= 10
    assert sorted((b.chn, b.nbrow, b.nlbyte, b.data) for b in bursts) == sorted(b.key() for b in ob)


def test_frames_of_channels_repaired_behind_the_payload_decode(built, oracle, monkeypatch):
    """The same handicap with repair rounds scheduled from the first push: the payload decode has run ahead on the first
    pass's selection when a round re-resolves a channel; those records (and the frames K4 would make of them) are void,
    the repaired selection is decoded behind the rounds -- bursts and frames must be the oracle's, none twice."""
    from vdlm2dec_amd import lib
    from vdlm2dec_amd.demod import Receiver, plan_channels
    monkeypatch.setenv("VDL2GPU_REPAIR_ROUNDS", "3")
    monkeypatch.setenv("VDL2GPU_DEBUG_COUNTERS", "1")
    spec = synth.random_scenario(2_000_000, S.FO8[:3], 1 << 21, seed=96, bursts_per_s=10.0, info_max=100)
    raw = synth.synth_stream(spec, "cs16")
    ob = oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC)
    want = sorted((0, b.chn, f) for b in ob for f in oracle.frames_of_block(b.nbrow, b.nlbyte, b.data))
    with Receiver(spec.rate, plan_channels(S.FC, spec.fo), fmt="cs16", max_push=1 << 20, frames=True,
                  flags=lib.F_TEST_NOREGION) as rx:
        bursts, frames = [], []
        for s0 in range(0, spec.nsamples, 600_000):
            rx.push(raw[2 * s0:2 * (s0 + 600_000)])
            frames += rx.poll_frames()
            bursts += rx.poll()
        repaired = rx.debug_counters(32, reset=False)[24]      # channel-pushes that went through a repair round
    assert repaired > 0
    assert sorted(frames) == want and len(want) >= 10
    assert sorted((b.chn, b.nbrow, b.nlbyte, b.data) for b in bursts) == sorted(b.key() for b in ob)
