"""probe (torch-free) for rocprofv3 --pmc passes: one free-running forward year at n x n"""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ooc_amd.engine import iage_engine
from nk_ooc_amd.grid import Grid2d
n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
grid = Grid2d.default(n, n)
eng = iage_engine(grid)
col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
fx, st, _ = eng.comp_fcn(x)
print(st)
