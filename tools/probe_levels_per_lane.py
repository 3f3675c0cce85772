import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from nk_ooc_amd.engine import iage_engine
from nk_ooc_amd.grid import Grid2d
for nz, ny in ((320, 48), (384, 48), (512, 48), (250, 48)):
    eng = iage_engine(Grid2d.default(nz, ny))
    eng.set_option("stream_years", 0)       # (by launches)
    eng.set_option("frozen_alloc_async", 0)
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x0 = np.stack([np.broadcast_to(col[:, None], (nz, ny))] * 2).copy()
    x = eng.upload(x0)
    fx, st, sched = eng.comp_fcn(x, record=True)
    zz = np.linspace(0.0, 1.0, nz)
    xp = eng.upload(x0 * (1.0 + 1.0e-4 * np.outer(np.sin(3.0 * zz), np.cos(2.0 * np.linspace(0, 1, ny)))[None]))
    eng.set_option("frozen_persistent", 0)
    want = [eng.download(eng.comp_fcn_frozen(v, sched)[0]) for v in (x, xp)]
    _, st_l = eng.comp_fcn_frozen(xp, sched)
    eng.set_option("frozen_persistent", 1)
    got = [eng.download(eng.comp_fcn_frozen(v, sched)[0]) for v in (x, xp)]
    _, st_p = eng.comp_fcn_frozen(xp, sched)
    print(f"{nz}x{ny} (E={(nz + 63) // 64}): identical {np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])}, "
          f"one-launch years {eng.counter('frozen_persistent_years')}, year {1e3*st_l['seconds']:.1f} -> {1e3*st_p['seconds']:.1f} ms, err checked {st_p['nerr_checked']}", flush=True)
    eng.close()
