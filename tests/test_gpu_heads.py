"""Header soft bits (what viterbi_add() is given, d8psk.c:81-83; viterbi.c:46) of the HIP path against the real
reference: tests/golden/*.json hold the sha256 of every such float of the reference run (make_golden.py taps
viterbi_add at link time); tests/test_golden.py pins the oracle's to it; here the GPU's (vdl2gpu_debug_heads: every
trigger any kernel of the push handled -- K2b's clusters of all eight timing classes, the resolver's serial
stretches) must contain, for every trigger of the real chain, an entry with exactly those 25 floats, and the digest
over them must be the reference's."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

import scenarios as S
from vdlm2dec_amd import lib

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "*.json")))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-5] for p in CASES])
def test_header_soft_bits_equal_the_reference(built, oracle, path):
    from vdlm2dec_amd.demod import Receiver, plan_channels
    meta = json.load(open(path))
    raw = np.load(os.path.join(HERE, "golden", meta["iq"] + ".npz"))["raw"]
    fmt = meta["fmt"] + ("_quirk" if meta["quirk"] else "")
    with Receiver(meta["rate"], plan_channels(meta["fc"], meta["fo"]), fmt=meta["fmt"], max_push=meta["nsamples"],
                  rtl_quirk=bool(meta["quirk"]), flags=lib.F_DEBUG_HEADS) as rx:
        rx.push(raw)            # one push: the tap holds the last push's triggers
        bursts = rx.poll()
        heads = rx.debug_heads()
        st = rx.stats()
    assert len(heads) >= st["triggers"] > 0
    checked = 0
    for chn in meta["channels"]:
        ch = oracle.OracleChannel(meta["rate"], chn["fo"], meta["fc"] + chn["fo"], chn=chn["chn"])
        ch.feed(raw, fmt)
        trigs = ch.triggers()
        ch.close()
        mine = heads[heads["sc"] == chn["chn"]]
        picked = []
        for t in trigs:
            if len(t["head"]) < 25:
                continue        # the recording ended inside the header: the GPU defers that trigger
            want = t["head"].view(np.uint32)
            cand = mine[mine["nstar"] == t["dec_index"]]
            hit = [e for e in cand if np.array_equal(e["soft"].view(np.uint32), want)
                   and e["perr"].view(np.uint32) == np.float32(t["perr"]).view(np.uint32)
                   and e["err"].view(np.uint32) == np.float32(t["err"]).view(np.uint32)
                   and e["clk0"] == t["clk"]]
            assert hit, (chn["chn"], t["dec_index"], len(cand))
            picked.append(hit[0]["soft"].astype("<f4").tobytes())
            checked += 1
        if all(len(t["head"]) == 25 for t in trigs):
            assert len(picked) * 25 == chn["n_head"]
            assert hashlib.sha256(b"".join(picked)).hexdigest() == chn["head_sha256"]     # the real reference's digest
    assert checked >= len(bursts) > 0
