"""phosphorus: free-running and frozen forward year by launches and as command streams"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nk_ooc_amd.engine import phosphorus_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
grid = Grid2d.default(n, n)
eng = phosphorus_engine(grid)
prof = [np.interp(grid.depth.mid, d, v) for d, v in (([1.3e2, 2.6e2], [5.5e-3, 4.1]), ([9.5e1, 1.4e2], [7.1e-2, 1.5e-4]),
                                                     ([1.7e2, 2.5e2], [1.8e-2, 7.9e-4]))]
x0 = np.stack([np.broadcast_to(p[:, None], (n, n)) for p in prof]).copy()
x = eng.upload(x0)
zz = np.linspace(0.0, 1.0, n)
xp = eng.upload(x0 * (1.0 + 1.0e-5 * np.outer(np.sin(3.0 * zz), np.cos(2.0 * zz))[None]))
res = {}
sched = None
for mode in (0, 3):
    eng.set_option("stream_years", mode)
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        fx, st, sc = eng.comp_fcn(x, record=True)
        wall = time.perf_counter() - t0
        if best is None or wall < best[0]:
            best = (wall, st)
    if sched is None:
        sched = sc
    bestf = None
    for rep in range(3):
        t0 = time.perf_counter()
        fz, stf = eng.comp_fcn_frozen(xp, sched)
        wall = time.perf_counter() - t0
        if bestf is None or wall < bestf[0]:
            bestf = (wall, stf)
    res[mode] = (eng.download(fx), eng.download(fz), sc)
    print(f"phosphorus {n}^2 stream_years={mode}: free-running year {best[0]:.4f} s ({best[1]['nsteps']} steps, {best[1]['nnewton']} Newton "
          f"iterations, {best[1]['nlaunch']} launches); frozen year {bestf[0]:.4f} s ({bestf[1]['nnewton']} Newton iterations, "
          f"{bestf[1]['nlaunch']} launches)", flush=True)
pr = [eng.counter(f"stream_prof_{i}") for i in range(12)]
print(f"    stream years per workgroup: waiting for commands {pr[0] / 1e3:.1f} ms, executing {pr[1] / 1e3:.1f} ms, waiting for neighbours "
      f"{pr[2] / 1e3:.1f} ms, {pr[3]} commands; SETUP {pr[8]} x {pr[4] / max(pr[8], 1):.1f} us, NEWTON {pr[9]} x {pr[5] / max(pr[9], 1):.1f} us, "
      f"ERR {pr[10]} x {pr[6] / max(pr[10], 1):.1f} us, BOUNDARY {pr[11]} x {pr[7] / max(pr[11], 1):.1f} us", flush=True)
print("bit-identical:", np.array_equal(res[0][0], res[3][0]), np.array_equal(res[0][1], res[3][1]), np.array_equal(res[0][2], res[3][2]))
eng.close()
