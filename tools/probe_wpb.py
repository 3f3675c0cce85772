"""the one-launch frozen year of a large grid (a wave per column, neighbour hand-over) by columns per workgroup, option "frozen_wpb"
    python tools/probe_wpb.py [n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
eng = iage_engine(Grid2d.default(n, n))
eng.set_option("frozen_cache_after", 0)
eng.set_option("frozen_team", 0)
col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
x = eng.axpby(1.0, x, 1.0, eng.comp_fcn(x)[0])
fx, st, sched = eng.comp_fcn(x, record=True)
want = eng.download(fx)
for wpb in (4, 2, 1, 4, 2, 1):
    eng.set_option("frozen_wpb", wpb)
    best = 1e9
    for _ in range(3):
        out, stf = eng.comp_fcn_frozen(x, sched)
        best = min(best, stf["seconds"])
    print(f"{n}x{n} columns per workgroup {wpb}: year {1e3 * best:.2f} ms, identical {np.array_equal(eng.download(out), want)}, "
          f"one-launch years {eng.counter('frozen_persistent_years')}", flush=True)
