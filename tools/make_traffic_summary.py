"""profiles/rNN_pmc_traffic_one_launch_N.json from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/summarize_pmc.py
output) over tools/probe_traffic.py and that probe's own line (algorithmic bytes of the last frozen year)

    python tools/make_traffic_summary.py pmc_FETCH_SIZE.json pmc_WRITE_SIZE.json pmc_FETCH_SIZE.log out.json [n]
"""
import json
import sys

fetch, write, log, out = sys.argv[1:5]
n = int(sys.argv[5]) if len(sys.argv) > 5 else 416
fj, wj = json.load(open(fetch)), json.load(open(write))
line = next(ln for ln in open(log) if ln.startswith("{"))
probe = json.loads(line)
kern = next(k for k in fj if k.startswith("void k_frozen_persistent") or k.startswith("k_frozen_persistent"))
f_kb = fj[kern]["FETCH_SIZE"]["mean"]
w_kb = wj[kern]["WRITE_SIZE"]["mean"]
alg = probe["last_frozen_year"]["algorithmic_bytes"]
upper = 1024.0 * (2.0 * f_kb + w_kb)
res = {
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 tools/probe_traffic.py %d (one free-running forward year by "
               "launches + two years frozen on its steps, each ONE launch of k_frozen_persistent on the schedule cache)" % n,
    "kernel": kern.replace("void ", ""),
    "dispatches": fj[kern]["FETCH_SIZE"]["dispatches"],
    "phases_of_the_year": probe["frozen_year"]["nsweeps"],
    "steps_of_the_year": probe["frozen_year"]["nsteps"],
    "FETCH_SIZE_mean_KB": f_kb, "WRITE_SIZE_mean_KB": w_kb,
    "note": "gfx950 FETCH_SIZE tallies 128-B requests at 64 B: doubled (calibrated on k_pc_gemv in round 2); Infinity-Cache hits are counted: "
            "fabric traffic, not pure HBM.  Per launch = per YEAR.  Round 4: the static coefficients and W of a column live in LDS for "
            "the year, so the traffic falls BELOW the algorithmic bytes of the launches the year replaces.",
    "traffic_bytes_per_launch_upper": upper,
    "traffic_bytes_per_launch_lower": 1024.0 * (f_kb + w_kb),
    "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": upper / alg,
    "grid": [n, n],
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
