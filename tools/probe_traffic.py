"""probe (torch-free) for rocprofv3 --pmc passes: the years of a Krylov solve at n x n -- one free-running forward year
(F(x)) and two years frozen on its steps (the perturbed years of finite-difference products); prints the algorithmic
bytes per launch of the dominant kernel over exactly these launches, for the comparison with FETCH_SIZE / WRITE_SIZE"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
grid = Grid2d.default(n, n)
eng = iage_engine(grid)
eng.set_option("stream_years", 0)       # (by launches)
col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
eng.profile_reset(0)
fx, st, sched = eng.comp_fcn(x, record=True)
rng = np.random.default_rng(0)
eng.set_option("frozen_cache_after", 0)      # (a cache of this size is otherwise built for the fourth year of a schedule)
eng.set_option("frozen_err_check", 0)       # (white-noise perturbations: not states the recorded steps control)
for _ in range(2):
    xp = eng.upload(eng.download(x) * (1.0 + 1.0e-4 * rng.standard_normal(eng.shape)))
    before = eng.profile_totals()
    _, stf = eng.comp_fcn_frozen(xp, sched)
    after = eng.profile_totals()
shapes = eng.profile_shapes()
totals = eng.profile_totals()
print(json.dumps({"free_year": st, "frozen_year": stf, "launch_shapes_without_factorisation": shapes,
                  "all_launches": totals,
                  "last_frozen_year": {"algorithmic_bytes": after["bytes"] - before["bytes"],
                                       "launches_booked": after["launches"] - before["launches"],
                                       "one_launch_years_so_far": eng.counter("frozen_persistent_years")},
                  "algorithmic_bytes_per_launch_without_factorisation": sum(shapes["bytes"]) / max(sum(shapes["counts"]), 1)}))
