"""what does the schedule cache of the one-launch frozen year cost at a large grid?  first build (allocation included), build for a
second schedule (buffers kept), and the years on either path.   python tools/probe_cache_build.py [n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
eng = iage_engine(Grid2d.default(n, n))
eng.set_option("frozen_persistent_max_e", 8)
eng.set_option("frozen_cache_gb", 128.0)
eng.set_option("frozen_cache_after", 0)
col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
x = eng.axpby(1.0, x, 1.0, eng.comp_fcn(x)[0])


def timed(vec, sched):
    eng.sync()
    t0 = time.perf_counter()
    out, st = eng.comp_fcn_frozen(vec, sched)
    eng.sync()
    return 1e3 * (time.perf_counter() - t0), st


fx, st, sched = eng.comp_fcn(x, record=True)
extra = sys.argv[2:] if len(sys.argv) > 2 else []
if "precond" in extra:      # the preconditioner's 10 GB first, as a solver has them
    t0 = time.perf_counter()
    eng.precond_setup()
    eng.sync()
    print(f"precond_setup {1e3 * (time.perf_counter() - t0):.0f} ms", flush=True)
if "launchfirst" in extra:  # a launch-per-phase frozen year first
    eng.set_option("frozen_persistent", 0)
    print(f"launch-per-phase year first: {timed(x, sched)[0]:.1f} ms", flush=True)
    eng.set_option("frozen_persistent", 1)
if "hist" in extra:         # a year with history samples first (the 169 MB host array of a Newton iteration)
    t0 = time.perf_counter()
    eng.comp_fcn_hist(x, np.linspace(0.0, 365.0 * 86400.0, 61))
    print(f"comp_fcn_hist {1e3 * (time.perf_counter() - t0):.0f} ms", flush=True)
if "churn" in extra or "churn_dev" in extra:        # what a solver leaves behind: hundreds of vectors allocated, some kept
    keep = [eng.upload(np.zeros(eng.shape)) for _ in range(200)]
    del keep[::2]
    print("device churn done", len(keep), flush=True)
if "churn" in extra or "churn_host" in extra:       # ... and host arrays
    host = [np.ones((61,) + eng.shape) for _ in range(3)]
    print("host churn done", sum(h.nbytes for h in host) / 1e6, "MB host", flush=True)
t_first, _ = timed(x, sched)
t_again, st_a = timed(x, sched)
x2 = eng.axpby(1.0, x, 0.5, fx)
fx2, st2, sched2 = eng.comp_fcn(x2, record=True)
t_second, _ = timed(x2, sched2)
t_again2, _ = timed(x2, sched2)
eng.set_option("frozen_persistent", 0)
t_launch, st_l = timed(x2, sched2)
print(f"{n}x{n}: first one-launch year of the first schedule {t_first:.1f} ms (allocation + cache build + year), the same year again "
      f"{t_again:.1f} ms; first year of a second schedule ({len(sched2)} steps) {t_second:.1f} ms -> cache build {t_second - t_again2:.1f} ms, "
      f"again {t_again2:.1f} ms; launch-per-phase year {t_launch:.1f} ms; cache builds {eng.counter('frozen_cache_builds')}", flush=True)
