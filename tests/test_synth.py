"""The synthetic transmitter: every burst it emits is decoded CRC-clean by the oracle
(which is itself pinned to the reference), across FEC regimes, rates and formats."""
import numpy as np
import pytest

import scenarios as S
from vdlm2dec_amd import synth


def _frames(O, spec, fmt):
    raw = synth.synth_stream(spec, fmt)
    blocks = O.run_oracle(raw, fmt, spec.rate, spec.fo, S.FC)
    return blocks, [f for b in blocks for f in O.frames_of_block(b.nbrow, b.nlbyte, b.data)]


def test_all_regimes_decode(oracle):
    spec = S.regimes(seed=200)
    blocks, frames = _frames(oracle, spec, "cu8")
    sent = {b.payload() for b in spec.bursts}
    assert len(frames) >= len(spec.bursts)
    # every transmitted AVLC frame comes back verbatim between the flags
    got = {f[1:-3] for f in frames}
    for b in spec.bursts:
        assert synth.avlc_frame(b.info, src=(1 << 24) | (0x400000 + b.chan * 0x111 + len(b.info))) in got
    assert sent


@pytest.mark.parametrize("fmt", ["cs16", "cf32"])
def test_formats(oracle, fmt):
    spec = S.regimes(seed=201, infos=(9, 45, 100))
    _, frames = _frames(oracle, spec, fmt)
    assert len(frames) == 3


def test_chunked_feed_equals_one_shot(oracle):
    spec = S.eight_channels(seed=202)
    raw = synth.synth_stream(spec, "cs16")
    a = oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC)
    b = oracle.run_oracle(raw, "cs16", spec.rate, spec.fo, S.FC, chunk=12345)
    assert [x.key() for x in a] == [x.key() for x in b]
    assert len(a) >= 12


def test_rs_repairs_up_to_three_byte_errors(oracle):
    spec = S.corrupted()
    blocks, frames = _frames(oracle, spec, "cu8")
    assert len(blocks) == 5
    assert len(frames) == 4          # the 4-error row is beyond t=3
