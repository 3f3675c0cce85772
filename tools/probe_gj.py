"""probe: block Gauss-Jordan factorisation time (nk2d_shift_factor) with one and two systems at n x n"""
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
for shifts in ([0.02], [0.02, 0.01], [0.02], [0.02, 0.01]):
    t0 = time.time()
    eng.shift_factor(0.5 * 365 * 86400.0, 365 * 86400.0, shifts)
    eng.sync()
    print("factor %d shift(s) %.3f s" % (len(shifts), time.time() - t0), flush=True)
