"""one-launch frozen year at 416 levels and a growing number of ypos columns: microseconds per phase against the number of
workgroups a compute unit has to hold (a workgroup = one ypos column = two waves; 256 compute units)"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

nz = 416
for ny in [int(a) for a in sys.argv[1:]] or [64, 128, 256, 288, 320, 416]:
    eng = iage_engine(Grid2d.default(nz, ny))
    eng.set_option("frozen_alloc_async", 0)
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x0 = np.stack([np.broadcast_to(col[:, None], (nz, ny))] * 2).copy()
    x = eng.upload(x0)
    xp = eng.upload(x0 * (1.0 + 1.0e-4 * np.outer(np.sin(3.0 * np.linspace(0.0, 1.0, nz)), np.cos(2.0 * np.linspace(0.0, 1.0, ny)))[None]))
    fx, st, sched = eng.comp_fcn(x, record=True)
    runs = [eng.comp_fcn_frozen(xp, sched)[1] for _ in range(5)]
    best = min(runs, key=lambda s: s["seconds"])
    print(f"{nz} x {ny}: one-launch frozen year {1e3 * best['seconds']:.2f} ms, {best['nsweeps']} phases, "
          f"{1e6 * best['seconds'] / best['nsweeps']:.2f} us per phase, one-launch years {eng.counter('frozen_persistent_years')}, "
          f"workgroups per compute unit up to {(ny + 255) // 256}", flush=True)
    eng.close()
