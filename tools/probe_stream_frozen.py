"""frozen year of iage: launch by launch, ONE launch on the schedule cache (k_frozen_persistent), as a command stream"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [416]
for n in sizes:
    eng = iage_engine(Grid2d.default(n, n))
    eng.set_option("stream_years", 0)       # (by launches)
    eng.set_option("frozen_alloc_async", 0)
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x0 = np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy()
    x = eng.upload(x0)
    zz = np.linspace(0.0, 1.0, n)
    xp = eng.upload(x0 * (1.0 + 1.0e-4 * np.outer(np.sin(3.0 * zz), np.cos(2.0 * zz))[None]))
    fx, st, sched = eng.comp_fcn(x, record=True)
    res = {}
    for name, opts in (("launches", {"frozen_persistent": 0, "stream_years": 0}),
                       ("one launch on the cache", {"frozen_persistent": 1, "stream_years": 0}),
                       ("command stream", {"frozen_persistent": 0, "stream_years": 2})):
        for k, v in opts.items():
            eng.set_option(k, v)
        best = None
        for rep in range(4):
            t0 = time.perf_counter()
            f, s = eng.comp_fcn_frozen(xp, sched)
            wall = time.perf_counter() - t0
            if best is None or wall < best[0]:
                best = (wall, s)
        res[name] = eng.download(f)
        print(f"{n}^2 frozen year, {name}: {1e3 * best[0]:.1f} ms ({best[1]['nlaunch']} launches, {best[1]['nnewton']} Newton iterations, "
              f"{best[1]['nerr_checked']} estimates checked)  same bits {np.array_equal(res[name], res['launches'])}", flush=True)
    pr = [eng.counter(f"stream_prof_{i}") for i in range(12)]
    print(f"    stream years per workgroup: waiting for commands {pr[0] / 1e3:.1f} ms, executing {pr[1] / 1e3:.1f} ms, waiting for "
          f"neighbours {pr[2] / 1e3:.1f} ms, {pr[3]} commands; NEWTON {pr[9]} x {pr[5] / max(pr[9], 1):.2f} us, BOUNDARY {pr[11]} x "
          f"{pr[7] / max(pr[11], 1):.2f} us, SETUP {pr[8]}, ERR {pr[10]}", flush=True)
    eng.close()
