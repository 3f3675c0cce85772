"""probe: a handful of line-relaxation solves at n x n (for rocprofv3 --pmc runs)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from nk_ooc_amd.engine import iage_engine
from nk_ooc_amd.grid import Grid2d
n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = iage_engine(Grid2d.default(n, n))
rng = np.random.default_rng(0)
b = eng.upload(rng.standard_normal((2, n, n)))
bi = eng.upload(rng.standard_normal((2, n, n)))
year = 365 * 86400.0
for r in range(reps):
    x, _, m = eng.shifted_solve(0.3 * year, 5e-5 * year, 3.637834252744496, b)
    x, xi, m2 = eng.shifted_solve(0.3 * year, 5e-5 * year, 2.6810828736277523 - 3.050430199247411j, b, bi)
eng.sync()
print("sweeps real", m, "complex", m2)
