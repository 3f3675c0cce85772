"""the figures of a bench line at a glance:  python tools/show_bench.py FILE.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
print("value", d["value"], "reference-semantic", d["value_reference_semantic"], "ms/step", d["ms_per_step"])
r = d["roofline"]
print("roofline:", r["kernel"][:60], "frac", round(r["frac"], 4), "avg launch us", round(r["avg_launch_us"], 2), "rocprof", r.get("rocprofv3"))
print("roofline_year", round(d["roofline_year"]["frac"], 4), "year s", d["roofline_year"]["seconds_per_year"], "gmres in HBM", d.get("gmres_solve_in_hbm", {}).get("jvps_per_s"))
lp = d.get("launch_per_phase_path")
if lp:
    print("launch-per-phase path:", round(lp["jvps_per_s"], 3), "JVPs/s, year", lp["year_seconds"], "roofline", round(lp["roofline"]["frac"], 4),
          round(lp["roofline"]["avg_launch_us"], 2), "us")
if "ladder" in d:
    print("ladder", [(row["grid"], round(row["jvps_per_s"], 2)) for row in d["ladder"]])
for key in ("config4_mix", "shard_e3"):
    if key in d:
        print(key, json.dumps(d[key])[:300])
print("jvp", {k: d["jvp"][k] for k in ("frozen_years_rejected", "frozen_years_resumed")})
