"""Known-answer tests for the integer pieces of the path (no floats involved)."""
import ctypes as C

import numpy as np
import pytest

from vdlm2dec_amd import synth


def test_reversebits(oracle):
    L = oracle.lib()
    assert L.vo_reversebits(0b1, 17) == 1 << 16
    assert L.vo_reversebits(0b1011, 4) == 0b1101
    assert L.vo_reversebits(0x12345, 17) == int(format(0x12345, "017b")[::-1], 2)
    for n in (1, 6, 7, 17):
        for v in (0, 1, (1 << n) - 1, 0x15555 & ((1 << n) - 1)):
            assert L.vo_reversebits(L.vo_reversebits(v, n), n) == v


def test_pn_sequence_matches_transmitter(oracle):
    L = oracle.lib()
    buf = np.zeros(20000, np.uint8)
    L.vo_pn_bits(buf.ctypes.data_as(C.c_void_p), buf.size)
    assert np.array_equal(buf, synth.pn_sequence(buf.size))
    # the first bits of the 0x4D4B LFSR (x^15 + x + 1 feedback taps 0 and 14)
    s, exp = 0x4D4B, []
    for _ in range(64):
        b = (s ^ (s >> 14)) & 1
        s = (s << 1) | b
        exp.append(b)
    assert buf[:64].tolist() == exp


@pytest.mark.parametrize("length", [96, 97, 1991, 1992, 1993, 4000, 15935, 131071, 0, 8])
def test_header_code_roundtrip(oracle, length):
    """(25,20) header: encode with the parity-check columns, decode with the restated Viterbi,
    also with one soft bit pushed the wrong way (the code corrects single errors)."""
    L = oracle.lib()
    hb = synth.header_bits(length)
    soft = np.array([0.999998 if b else 0.000002 for b in hb], np.float32)
    bits = C.c_uint32()
    assert L.vo_header_decode(soft.ctypes.data_as(C.POINTER(C.c_float)), C.byref(bits)) == length
    for flip in (5, 12, 19, 22):
        s2 = soft.copy()
        s2[flip] = 0.35 if hb[flip] else 0.65
        assert L.vo_header_decode(s2.ctypes.data_as(C.POINTER(C.c_float)), None) == length


def test_fec_layout_regimes():
    # (len bits) -> (nbrow, nlbyte, rows with FEC, FEC bytes of the last row)   d8psk.c:94-95,153-161
    assert synth.fec_layout(8 * 13) == (1, 13, 1, 2)
    assert synth.fec_layout(8 * 30) == (1, 30, 1, 2)
    assert synth.fec_layout(8 * 31) == (1, 31, 1, 4)
    assert synth.fec_layout(8 * 67) == (1, 67, 1, 4)
    assert synth.fec_layout(8 * 68) == (1, 68, 1, 6)
    assert synth.fec_layout(8 * 249 + 8) == (2, 1, 1, 6)      # last row <= 2 bytes: dropped from FEC
    assert synth.fec_layout(8 * 249 + 16) == (2, 2, 1, 6)
    assert synth.fec_layout(8 * 249 + 24) == (2, 3, 2, 2)
    assert synth.fec_layout(8 * 249) == (2, 0, 1, 6)          # len % 1992 == 0 quirk


def test_rs_encode_decode(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(1)
    for nerr, nera in [(0, 0), (1, 0), (3, 0), (2, 2), (1, 4), (0, 4)]:
        row = rng.integers(0, 256, 249, dtype=np.uint8)
        cw = np.concatenate([row, np.array(synth.rs_parity(row.tolist()), np.uint8)])
        good = cw.copy()
        eras = {0: [], 2: [253, 254], 4: [251, 252, 253, 254]}[nera]
        for e in eras:
            cw[e] = 0
        for pos in rng.choice(249, nerr, replace=False):
            cw[pos] ^= 0x5A
        ep = (C.c_int * 6)(*(eras + [0] * (6 - len(eras))))
        r = L.vo_rs_decode(cw.ctypes.data_as(C.c_void_p), ep, nera)
        assert r >= 0
        assert np.array_equal(cw[:249], good[:249])
    # 4 errors without erasures exceed t=3
    row = rng.integers(0, 256, 249, dtype=np.uint8)
    cw = np.concatenate([row, np.array(synth.rs_parity(row.tolist()), np.uint8)])
    for pos in (3, 50, 100, 200):
        cw[pos] ^= 0x11
    assert L.vo_rs_decode(cw.ctypes.data_as(C.c_void_p), (C.c_int * 6)(), 0) == -1


def test_fcs_and_hdlc_roundtrip(oracle):
    info = bytes(range(1, 60)) + b"\x7e\x7d\xff\xff\xff\x1f"
    frame = synth.avlc_frame(info)
    pl = synth.hdlc_payload(frame)
    data = np.zeros((8, 255), np.uint8)
    data[0, :len(pl)] = np.frombuffer(pl, np.uint8)
    data[0, 249:] = synth.rs_parity(data[0, :249].tolist())
    fr = oracle.frames_of_block(1, len(pl), data.tobytes())
    assert len(fr) == 1
    crc = synth.fcs16(frame)
    assert fr[0] == b"\x7e" + frame + bytes([crc & 0xFF, crc >> 8]) + b"\x7e"


def test_decimation_schedule_closed_form(built):
    """vdl2gpu_plan() against a brute-force run of the reference's clock loop (d8psk.c:369-381)."""
    from vdlm2dec_amd.demod import plan
    rng = np.random.default_rng(3)
    for sdrclk, L in ((500, 80), (1250, 200), (1500, 240), (2500, 400)):
        clk = nf = no = 0
        total = 0
        for _ in range(40):
            n = int(rng.integers(1, 5000))
            c0, no0, nf0, nout = plan(total, n, sdrclk, L)
            assert (c0, no0, nf0) == (clk, no, nf)
            outs = 0
            for _i in range(n):
                nf += 1
                no = (no + 1) % L
                clk += 21
                if clk >= sdrclk:
                    clk %= sdrclk
                    nf = 0
                    outs += 1
            assert outs == nout
            total += n
