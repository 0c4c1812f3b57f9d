"""dev aid: the drop-in executable on a golden recording, block by block against the golden file: which block differs, and where"""
import json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name = sys.argv[1] if len(sys.argv) > 1 else "regimes_cu8_2ms"
meta = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
raw = np.load(os.path.join(ROOT, "tests", "golden", meta["iq"] + ".npz"))["raw"]
td = tempfile.mkdtemp()
iq = os.path.join(td, "iq.raw"); raw.tofile(iq); out = os.path.join(td, "out.txt")
fos = ",".join(str(f) for f in meta["fo"]); frs = ",".join(str(meta["fc"] + f) for f in meta["fo"])
env = dict(os.environ); env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_rtl_gpu"), iq, meta["fmt"], str(meta["rate"]), fos, frs, out, "0", ""], check=True, env=env)
multi = len(meta["fo"]) > 1
blocks = {}
for line in open(out):
    p = line.split()
    if p[0] == "B":
        chn = int(p[6][1:]) if multi else 0
        blocks.setdefault(chn, []).append((int(p[1]), int(p[2]), int(p[4], 16), p[-1]))
for c in meta["channels"]:
    want = [(b["nbrow"], b["nlbyte"], b["df_bits"], b["data"]) for b in c["blocks"]]
    got = blocks.get(c["chn"], [])
    print("chn", c["chn"], "got", len(got), "want", len(want))
    for i, (g, w) in enumerate(zip(got, want)):
        if g != w:
            print("  block", i, "got", g[:3], "want", w[:3], "data equal", g[3] == w[3])
            if g[3] != w[3]:
                d = [k // 2 for k in range(0, min(len(g[3]), len(w[3])), 2) if g[3][k:k + 2] != w[3][k:k + 2]]
                print("   differing byte offsets", d[:20], "count", len(d), "lens", len(g[3]), len(w[3]))
                print("   got ", [g[3][2 * k:2 * k + 2] for k in d[:20]])
                print("   want", [w[3][2 * k:2 * k + 2] for k in d[:20]])
