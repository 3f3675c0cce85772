"""probe: forward year of iage at n x n (default controller mode) for a range of inner tolerances, now that a solve
whose bound meets the tolerance after ONE sweep runs its Newton iteration as a single launch: time, iteration and
launch counts, and the distance from a year integrated 1000 times tighter in units of the CI tolerance"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
grid = Grid2d.default(n, n)
col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
y0 = np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy()
tight = iage_engine(grid, rtol=1.0e-9, atol=1.0e-9, lin_tol=1.0e-10)
tight.set_option("stream_years", 0)       # (by launches)
tight.set_option("jac_fresh", 0)
x0 = tight.upload(y0)
x0 = tight.axpby(1.0, x0, 1.0, tight.comp_fcn(x0)[0])
xh = tight.download(x0)
ref = tight.download(tight.comp_fcn(x0)[0])
tight.close()
eng = iage_engine(grid)
eng.set_option("stream_years", 0)       # (by launches)
x = eng.upload(xh)
for min_sweeps, tol in ((2, 3e-2), (1, 1e-2), (1, 3e-2), (1, 6e-2), (1, 1e-1), (1, 2e-1), (1, 3e-1)):
    eng.set_option("min_sweeps", min_sweeps)
    eng.set_option("lin_tol", tol)
    best = None
    for _ in range(2):
        fx, st, _ = eng.comp_fcn(x)
        best = st if best is None or st["seconds"] < best["seconds"] else best
    got = eng.download(fx)
    margin = float(np.max(np.abs(got - ref) / (1.0e-6 + 1.0e-3 * np.abs(ref))))
    print(f"n={n} min_sweeps={min_sweeps} lin_tol={tol:g}: {best['seconds']:.4f} s, steps {best['nsteps']}, Newton "
          f"{best['nnewton']} ({best['nnewton'] / best['nsteps']:.2f}/step), fused launches {best['nsweeps']} "
          f"({best['nsweeps'] / best['nnewton']:.2f}/iteration), launches {best['nlaunch']}, margin {margin:.3f}", flush=True)
