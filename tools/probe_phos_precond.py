"""probe: phases of the phosphorus preconditioner set-up and apply at n x n"""
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
po4 = np.broadcast_to(np.interp(grid.depth.mid, [1.3e2, 2.6e2], [5.5e-3, 4.1e0])[:, None], (n, n)).copy()
ylin = np.zeros(eng.shape)
ylin[0] = po4
eng.set_lin_state(eng.upload(ylin))
t0 = time.time()
eng.shift_factor(0.5 * 365 * 86400.0, 365 * 86400.0, [0.02])
eng.sync()
print("factor 1 shift %.3f s" % (time.time() - t0))
v = eng.upload(np.random.default_rng(0).standard_normal(eng.shape))
eng.sync()
for rep in range(3):
    t0 = time.time()
    out = eng.shift_solve(0, v)
    eng.sync()
    print("solve %.4f s" % (time.time() - t0))
t0 = time.time()
host = eng.download(out)
print("download %.4f s" % (time.time() - t0))
t0 = time.time()
eng.upload(host)
print("upload %.4f s" % (time.time() - t0))
t0 = time.time()
pc = eng.precond_setup_state(po4)
eng.sync()
print("precond_setup_state %.3f s, %d solves" % (time.time() - t0, pc.eig_solves), pc.clock)
t0 = time.time()
for rep in range(5):
    out = eng.precond_apply(v)
eng.sync()
print("precond_apply %.4f s" % ((time.time() - t0) / 5))
