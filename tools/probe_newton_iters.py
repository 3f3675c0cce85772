"""probe: Newton iterations per Radau step of the iage forward year as a function of the inner tolerance
(what bounds the convergence rate of the simplified Newton iteration: the inexact line relaxation, or the
Jacobian frozen at the start of the step while the vertical mixing changes over it)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ooc_amd.engine import iage_engine, phosphorus_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
module = sys.argv[2] if len(sys.argv) > 2 else "iage"
grid = Grid2d.default(n, n)
if module == "iage":
    col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    y0 = np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy()
    make = iage_engine
else:
    prof = [np.interp(grid.depth.mid, zs, vs) for zs, vs in (([1.3e2, 2.6e2], [5.5e-3, 4.1e0]),
                                                               ([9.5e1, 1.4e2], [7.1e-2, 1.5e-4]),
                                                               ([1.7e2, 2.5e2], [1.8e-2, 7.9e-4]))]
    y0 = np.stack([np.broadcast_to(p[:, None], (n, n)) for p in prof]).copy()
    make = phosphorus_engine
ref = None
# NK2D_GROWTH_CAPS=0,1.0,1.5: scan of the growth cap after a Newton failure at the default inner tolerance
caps = [float(v) for v in os.environ.get("NK2D_GROWTH_CAPS", "").split(",") if v]
for lin_tol in ((3e-2,) if caps else (1e-8, 1e-3, 1e-2, 3e-2, 1e-1, 3e-1)):
    for fresh in (range(1, len(caps) + 1) if caps else (0, 1)):
        eng = make(grid, lin_tol=lin_tol)
        eng.set_option("jac_fresh", 1 if fresh else 0)
        eng.set_option("growth_cap", caps[fresh - 1] if caps else 0.0)
        if caps:
            print("growth_cap", caps[fresh - 1], end=": ")
        x = eng.upload(y0)
        x = eng.axpby(1.0, x, 1.0, eng.comp_fcn(x)[0])      # the state after one fixed-point year, as in bench.py
        t0 = time.time()
        fx, st, _ = eng.comp_fcn(x)
        wall = time.time() - t0
        res = eng.download(fx)
        if ref is None:
            ref = res
        err = np.max(np.abs(res - ref)) / np.max(np.abs(ref))
        print(f"{module} lin_tol={lin_tol:g} jac_fresh={fresh}: {wall:.3f} s, steps {st['nsteps']}, rejected {st['nrejected']}, "
              f"newton {st['nnewton']} ({st['nnewton'] / st['nsteps']:.2f} per step), fused launches {st['nsweeps']} "
              f"({st['nsweeps'] / st['nnewton']:.2f} per iteration), nfev {st['nfev']}, njev {st['njev']}, nlu {st['nlu']}, "
              f"max |F - F(lin_tol 1e-8)| / max |F| = {err:.2e}", flush=True)
        eng.close()
