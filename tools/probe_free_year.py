"""what does a free-running (host-controlled) year launch, and what of it is dropped?   python tools/probe_free_year.py [n]"""
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
eng.set_option("stream_years", 0)       # (by launches)
col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
y0 = np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy()
x = eng.upload(y0)
eng.comp_fcn(x)
names = ("spec_launches_dropped", "spec_front_launches_dropped", "err_estimates_queued", "err_estimates_dropped")
before = {k: eng.counter(k) for k in names}
t0 = time.perf_counter()
fx, st, _ = eng.comp_fcn(x)
eng.sync()
el = time.perf_counter() - t0
after = {k: eng.counter(k) - before[k] for k in names}
print(f"{n}x{n} free-running year {el:.3f} s: steps {st['nsteps']} rejected {st['nrejected']} Newton {st['nnewton']} sweeps {st['nsweeps']} "
      f"launches {st['nlaunch']} nlu {st['nlu']} njev {st['njev']}; {after}", flush=True)
