"""Print the headline numbers and the per-launch table of a bench.py JSON line.  python scripts/show_bench.py file.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("frames/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 5), "latency ms", d.get("latency_ms_single_stream"), "one call ms", d.get("latency_ms_one_call"))
print("whole_path", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["whole_path"].items() if k != "per_frame"})
for key in ("roofline", "roofline_mfma_all", "roofline_fps"):
    r = d.get(key) or {}
    print(key, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in ("kernel", "us", "avg_launch_us", "achieved", "frac", "mfma_busy", "traffic", "us_per_round", "share_of_call", "launches")})
for r in d.get("roofline_hbm") or []:
    print("hbm/valu", r["entry"], round(r["us"], 1), "us frac", round(r.get("frac", 0), 3), r.get("mfma_frac"))
print("eager one stream us:", round(d["launches"]["eager_one_stream_us"], 1))
for r in d["launches"]["table"]:
    print(f"{r['entry'][:36]:36s} {r['us']:8.1f} {str(r.get('bound')):8s} {str(r.get('frac')):6s} | {r['what'][:100]}")
