"""probe: phosphorus forward year with the Jacobian's mixing plane at the step start (default) or at the second stage
time (option "jac_stage_state": the state part stays at the step start): counters, time, distance between the
results and to a year integrated 1000 times tighter, in units of the CI tolerance; frozen year on each schedule"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ooc_amd.engine import phosphorus_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
grid = Grid2d.default(n, n)
prof = [np.interp(grid.depth.mid, zs, vs) for zs, vs in (([1.3e2, 2.6e2], [5.5e-3, 4.1e0]),
                                                           ([9.5e1, 1.4e2], [7.1e-2, 1.5e-4]),
                                                           ([1.7e2, 2.5e2], [1.8e-2, 7.9e-4]))]
y0 = np.stack([np.broadcast_to(p[:, None], (n, n)) for p in prof]).copy()
tight = phosphorus_engine(grid, rtol=1.0e-9, atol=1.0e-9, lin_tol=1.0e-10)
tight.set_option("jac_fresh", 0)
ref = tight.download(tight.comp_fcn(tight.upload(y0))[0])
tight.close()
eng = phosphorus_engine(grid)
x = eng.upload(y0)
base = None
for flag in (0, 1, 0, 1):
    eng.set_option("jac_stage_state", flag)
    fx, st, sched = eng.comp_fcn(x, record=True)
    res = eng.download(fx)
    if base is None:
        base = res
    m_ref = float(np.max(np.abs(res - ref) / (1.0e-6 + 1.0e-3 * np.abs(ref))))
    m_base = float(np.max(np.abs(res - base) / (1.0e-6 + 1.0e-3 * np.abs(base))))
    fx2, stf = eng.comp_fcn_frozen(x, sched)
    same = bool(np.array_equal(eng.download(fx2), res))
    print(f"n={n} jac_stage_state={flag}: {st['seconds']:.3f} s, steps {st['nsteps']}, rejected {st['nrejected']}, Newton {st['nnewton']} "
          f"({st['nnewton'] / st['nsteps']:.2f}/step), launches {st['nlaunch']}; |F - F_tight| / tol {m_ref:.3f}, vs default {m_base:.3f}; "
          f"frozen year {stf['seconds']:.3f} s ({stf['nlaunch']} launches), bit-identical {same}", flush=True)
