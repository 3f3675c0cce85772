"""A/B of a library build (NK2D_LIB_PATH) on the frozen year: time of a replayed year, launches, and the year itself
(printed as norms so that two builds can be compared).   python tools/probe_now.py [n] [module]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
grid = Grid2d.default(n, n)
eng = iage_engine(grid)
col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
y0 = np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy()
x = eng.upload(y0)
t0 = time.perf_counter()
fx, st, sched = eng.comp_fcn(x, record=True)
eng.sync()
t_free = time.perf_counter() - t0
f0 = eng.download(fx)
rng = np.random.default_rng(1)
xp = eng.upload(y0 * (1.0 + 1.0e-4 * rng.standard_normal(y0.shape)))
best = 1e30
for rep in range(4):
    t0 = time.perf_counter()
    fp, stf = eng.comp_fcn_frozen(xp, sched)
    eng.sync()
    best = min(best, time.perf_counter() - t0)
f1 = eng.download(fp)
print(f"lib {os.environ.get('NK2D_LIB_PATH', 'default')}: {n}x{n} free-running year {t_free:.3f} s ({st['nsteps']} steps, {st['nnewton']} Newton), "
      f"frozen year {best * 1e3:.1f} ms, launches {stf['nlaunch']}, |F(x)| {np.linalg.norm(f0):.15e}, "
      f"|F(xp)-F(x)| {np.linalg.norm(f1 - f0):.15e}, sum F(xp) {f1.sum():.15e}", flush=True)
np.save(os.path.join("gpurun_out", "now_%s_%d.npy" % ("ab" if os.environ.get("NK2D_LIB_PATH") else "base", n)), np.stack([f0, f1]))
