"""probe: free-running forward year by the step growth allowed after a Newton failure (option "growth_cap"), in the
engines' default mode: time, counters, distance to a year integrated 1000 times tighter in units of the CI tolerance"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

for n in [int(v) for v in (sys.argv[1:] or ["52", "416"])]:
    grid = Grid2d.default(n, n)
    eng = iage_engine(grid)
    eng.set_option("stream_years", 0)       # (by launches)
    col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
    x = eng.axpby(1.0, x, 1.0, eng.comp_fcn(x)[0])
    tight = iage_engine(grid, rtol=1.0e-9, atol=1.0e-9, lin_tol=1.0e-10)
    tight.set_option("jac_fresh", 0)
    ref = tight.download(tight.comp_fcn(tight.upload(eng.download(x)))[0])
    tight.close()
    for cap in (0.0, 1.0, 1.5, 2.0, 3.0, 5.0):
        eng.set_option("growth_cap", cap)
        fx, st, _ = eng.comp_fcn(x)
        res = eng.download(fx)
        margin = float(np.max(np.abs(res - ref) / (1.0e-6 + 1.0e-3 * np.abs(ref))))
        print(f"n={n} growth_cap={cap}: {st['seconds']:.4f} s, steps {st['nsteps']}, rejected {st['nrejected']}, Newton {st['nnewton']} "
              f"({st['nnewton'] / st['nsteps']:.2f}/step), launches {st['nlaunch']}, |F - F_tight| / tol = {margin:.3f}", flush=True)
    eng.close()
