"""The transmitter as a tool: command line, ground truth, and the Es/N0 sweep's scoring."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synth_cli_writes_a_recording_the_oracle_decodes(tmp_path, oracle):
    iq, truth = str(tmp_path / "x.cs16"), str(tmp_path / "x.json")
    r = subprocess.run([sys.executable, "-m", "vdlm2dec_amd.synth", iq, "--fmt", "cs16", "--seconds", "0.25", "--fo", "-50000", "250000",
                        "--seed", "5", "--noise", "1.2", "--truth", truth], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    t = json.load(open(truth))
    raw = np.fromfile(iq, np.int16)
    assert raw.size == 2 * t["nsamples"] and t["nsamples"] % 32768 == 0 and len(t["bursts"]) >= 2
    got = oracle.run_oracle(raw, "cs16", t["rate"], t["fo"], 136_975_000)
    frames = [f for b in got for f in oracle.frames_of_block(b.nbrow, b.nlbyte, b.data)]
    assert len(got) == len(t["bursts"]) == len(frames)
    assert sorted((b.nbrow, b.nlbyte) for b in got) == sorted((b["nbrow"], b["nlbyte"]) for b in t["bursts"])


def test_ber_curve_oracle_only(tmp_path):
    out = str(tmp_path / "c.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ber_curve.py"), "--oracle-only", "--esn0", "16", "32", "--bursts", "16",
                        "--out", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    p = json.load(open(out))["points"]
    assert p[1]["oracle"]["frames_ok"] >= 14                    # 32 dB: (nearly) every frame; 16 dB: hardly any
    assert p[0]["oracle"]["frames_ok"] < p[1]["oracle"]["frames_ok"]
