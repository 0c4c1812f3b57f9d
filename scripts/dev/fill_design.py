"""dev aid: the numbers of DESIGN.md / README.md placeholders (@@NAME@@) from profiles/r06_*: prints a dict, or substitutes with --apply"""
import json, csv, sys, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = lambda f: os.path.join(ROOT, "profiles", f)
full = json.load(open(P("r06_bench_line_full.json")))
dflt = json.load(open(P("r06_bench_line.json")))
v = {}
v["DRV"] = "%.1f" % (full["value"] / 1e3)
v["DRVMS"] = "%.4f" % full["ms_per_step"]
v["STEADY"] = "%.3f (driver-style) / %.3f (default command)" % (full["steady_state"]["ms_per_step"], dflt["steady_state"]["ms_per_step"])
v["DEF"] = "%.1f" % (dflt["value"] / 1e3)
v["DEFMS"] = "%.4f" % dflt["ms_per_step"]
ks = list(csv.DictReader(open(P("r06_bench_kernel_stats.csv"))))
npush = next(int(r["Calls"]) for r in ks if r["Name"].startswith("k2a_probe"))
names = {"void k1_fast<1>": "K1", "k2a_probe": "probe", "k2r_regions": "regions", "k2a_region": "region scan", "k2s_sort": "sort", "k3_carry": "carry", "k_push_init": "init",
         "k2b_clusters": "clusters", "k2c_resolve": "resolver", "k2a_verify": "verify", "k2d_payload": "payload", "k2p_patch": "patch", "k2f_commit": "commit",
         "k_export_records": "export", "k3_rebase": "rebase"}
parts = []
for r in ks:
    n = r["Name"].split("(")[0]
    if n in names:
        c = int(r["Calls"]) / npush
        parts.append("%s %.1f%s" % (names[n], float(r["AverageNs"]) / 1e3, "" if abs(c - 1) < 0.2 else " × %d" % round(c)))
v["KSTATS"] = "; ".join(parts) + " µs"
ws = dflt["roofline"].get("whole_step") or {}
if ws:
    v["WHOLE"] = "%.1f M wave instructions = %.3f ms of the SIMDs' time (%.2f of the step); %.2f GB = %.1f × algorithmic" % (
        ws["valu_wave_insts"] / 1e6, ws["valu_floor_ms"], ws["valu_floor_ms"] / dflt["ms_per_step"], ws["hbm_traffic_bytes"] / 1e9, ws["hbm_traffic_over_algorithmic"])
    iu = ws.get("issue_utilisation") or {}
    v["ISSUE"] = "; ".join("%s %.0f / %.0f µs = %.0f %%" % (k.replace("k2a_", "").replace("k2b_", "").replace("k2c_", "").replace("k2d_", ""), x["valu_us"], x["kernel_us"], 100 * x["frac"])
                           for k, x in sorted(iu.items(), key=lambda kv: -kv[1]["valu_us"])[:8])
else:
    v["WHOLE"] = v["ISSUE"] = "(the committed PMC passes belong to other sources: not quoted)"
rf = dflt["roofline"]
v["K1ALONEMS"] = "%.4f" % rf["alone"]["avg_launch_ms"]
v["K1ALONE"] = "%.3f" % rf["alone"]["frac"]
v["K1LIVEMS"] = "%.4f" % rf["avg_launch_ms"]
v["K1LIVE"] = "%.3f" % rf["frac"]
v["TRAFFIC"] = "%.0f" % ((rf.get("traffic") or 0) / 1e6)
hm = full["host_ms_per_step"]
v["HOST"] = "in `vdl2gpu_push` %.3f ms = enqueue %.3f + wait_for_ring %.3f; in `vdl2gpu_poll_ready` %.3f" % (hm["in_push"], hm.get("enqueue", 0), hm.get("wait_for_ring", 0), hm["in_poll_ready"])
cf = full["configs"]
def leg(k):
    L = cf[k]
    return "%.1f" % (L["value"] / 1e3), "%.1f–%.1f" % (L["repeats"]["min"] / 1e3, L["repeats"]["max"] / 1e3)
v["C3"], v["C3R"] = leg("config3_8ch_10MSps")
v["C4"], v["C4R"] = leg("config4_share_8x8ch")
v["B15"], v["B15R"] = leg("config2_busy_15")
v["B30"], v["B30R"] = leg("config2_busy_30")
lv = cf["config5_live_ring"]["latency_ms"]
v["LIVE"] = "p50 %.2f ms, p99 %.2f, max %.2f" % (lv["p50"], lv["p99"], lv["max"])
dr = cf.get("dropin_replay") or {}
cb = full.get("cpu_baseline") or {}
v["DROPIN"] = "%.0f MS/s; CPU reference %.0f MS/s (%s threads)" % (dr.get("value") or 0, cb.get("value") or 0, cb.get("cores"))
try:
    so = [l.strip() for l in open(P("r06_soak.txt")) if l.startswith("soak")]
    v["SOAK"] = " + ".join(so)
except OSError:
    v["SOAK"] = "?"
if "--apply" in sys.argv:
    for f in ("DESIGN.md", "README.md"):
        s = open(os.path.join(ROOT, f)).read()
        for k, x in v.items():
            s = s.replace("@@" + k + "@@", x)
        open(os.path.join(ROOT, f), "w").write(s)
        print(f, "left:", re.findall(r"@@\w+@@", s))
else:
    for k, x in v.items():
        print(k, "=", x)
