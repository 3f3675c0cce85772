"""probe: forward year with one wave per column (k_newton_fused, option team 0) against one workgroup per
column (k_newton_team, team 1) under host control: wall time, counters, and whether the results are bit-identical

    python tools/probe_team.py [n ...] [--kind iage|phos]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nk_ooc_amd.engine import ModuleEngine, iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
for n in [int(v) for v in (args or ["26", "104", "416"])]:
    grid = Grid2d.default(n, n)
    if "--tc1" in sys.argv:   # one tracer: half the columns
        eng = ModuleEngine(grid, tc=1, surf_rate=(24.0 / 86400.0 * 10.0 / grid.depth.delta[0],), const_src=1.0 / (365.0 * 86400.0))
    else:
        eng = iage_engine(grid)
    tc = eng.tc if hasattr(eng, "tc") else (1 if "--tc1" in sys.argv else 2)
    eng.set_option("device_ctl", 0)
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * tc).copy())
    out = {}
    for rep in range(2):
        for team, xcd in ((0, 0), (0, 1), (1, 0), (1, 1), (2, 0)):
            eng.set_option("team", team)
            eng.set_option("xcd_map", xcd)
            fx, st, sched = eng.comp_fcn(x, record=True)
            _, stf = eng.comp_fcn_frozen(x, sched)
            out[(team, xcd)] = (st["seconds"], st["nsteps"], st["nnewton"], st["nsweeps"], eng.download(fx), stf["seconds"])
    ref = out[(0, 0)]
    for key, val in out.items():
        print(f"n={n}: team {key[0]} xcd_map {key[1]}: {val[0]:.4f} s ({ref[0] / val[0]:.2f}x), frozen year {val[5]:.4f} s ({ref[5] / val[5]:.2f}x), steps {val[1]}, Newton {val[2]}, "
              f"sweeps {val[3]}, identical to (0, 0): {np.array_equal(val[4], ref[4])}", flush=True)
    eng.close()
