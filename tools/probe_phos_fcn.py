"""probe: one phosphorus forward year at n x n (timing + counters)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ooc_amd.engine import phosphorus_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
grid = Grid2d.default(n, n)
eng = phosphorus_engine(grid)
prof = [np.interp(grid.depth.mid, zs, vs) for zs, vs in (([1.3e2, 2.6e2], [5.5e-3, 4.1e0]),
                                                           ([9.5e1, 1.4e2], [7.1e-2, 1.5e-4]),
                                                           ([1.7e2, 2.5e2], [1.8e-2, 7.9e-4]))]
y0 = np.stack([np.broadcast_to(p[:, None], (n, n)) for p in prof]).copy()
x = eng.upload(y0)
for rep in range(2):
    t0 = time.time()
    fx, st, _ = eng.comp_fcn(x)
    print(f"n={n} wall={time.time()-t0:.3f}s " + " ".join(f"{k}={v}" for k, v in st.items() if k != "seconds"), flush=True)
# the perturbed year of a product: frozen on the steps of the year above
sched = eng.last_schedule()
for rep in range(2):
    rng = np.random.default_rng(rep)
    xp = eng.upload(y0 * (1.0 + 1.0e-5 * rng.standard_normal(y0.shape)))
    t0 = time.time()
    fxp, stf = eng.comp_fcn_frozen(xp, sched)
    print(f"n={n} frozen year wall={time.time()-t0:.3f}s " + " ".join(f"{k}={v}" for k, v in stf.items() if k != "seconds"),
          f"rejected={eng.frozen_fallbacks()}", flush=True)
fx2, _ = eng.comp_fcn_frozen(x, sched)
print("frozen year of the recorded state bit-identical:", bool(np.array_equal(eng.download(fx2), eng.download(fx))), flush=True)
