"""probe: does GMRES get what it thinks it gets?  One Krylov solve of k iterations (nk2d_gmres_solve) with the products on
frozen years and with free-running products; for each the residual norm the solver REPORTS after every iteration and
the TRUE preconditioned residual |M^-1 (F + J d_j)| of the same iterate, J d_j from products of years integrated 10^4
times tighter (frozen on the tight base year's steps, i.e. the derivative of the accurate map)

    python tools/probe_gmres_quality.py [n] [k]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 52
k = int(sys.argv[2]) if len(sys.argv) > 2 else 6
grid = Grid2d.default(n, n)
eng = iage_engine(grid)
tight = iage_engine(grid, rtol=1.0e-10, atol=1.0e-10, lin_tol=1.0e-10)
tight.set_option("jac_fresh", 0)
weight = np.outer(grid.depth.delta, grid.ypos.delta)
mask = np.ones((n, n), dtype=np.int32)
for e in (eng, tight):
    e.set_region(mask, weight)
col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
x = eng.axpby(1.0, x, 1.0, eng.comp_fcn(x)[0])
xh = eng.download(x)
fx, _, sched = eng.comp_fcn(x, record=True)
xt = tight.upload(xh)
fxt, _, sched_t = tight.comp_fcn(xt, record=True)
eng.precond_setup()
tight.precond_setup()


def true_resid(inc_host):
    """|M^-1 (F + J inc)| / |M^-1 F| with the accurate map's derivative"""
    nrm = np.sqrt(tight.dot(tight.upload(inc_host), tight.upload(inc_host))[0])
    v = tight.upload(inc_host / nrm)
    w, sigma, _ = tight.jvp(xt, fxt, v, sched=sched_t)
    r = tight.axpby(1.0, fxt, nrm, w)
    pr, p0 = tight.precond_apply(r), tight.precond_apply(fxt)
    return float(np.sqrt(tight.dot(pr, pr)[0] / tight.dot(p0, p0)[0]))


for name, s in (("frozen products", sched), ("free-running products", None)):
    print(f"n={n}, {name}:", flush=True)
    for j in range(1, k + 1):
        inc, info = eng.gmres_solve(x, fx, 0.0, 0, j, sched=s)
        reported = float(info["resid_norm"][j - 1][0] / info["beta"][0])
        print(f"    after {j} iterations: reported |r_j| / |r_0| = {reported:.3e}, true = {true_resid(eng.download(inc)):.3e}", flush=True)
