"""probe: Newton iterations per Radau step when the Jacobian of a step attempt is taken at one of its stage times
(option "jac_stage") instead of the step start: the vertical mixing changes over a step, and the simplified Newton
iteration converges at the rate of the mismatch between the frozen Jacobian and the stage Jacobians"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

for n in [int(v) for v in (sys.argv[1:] or ["52", "416"])]:
    eng = iage_engine(Grid2d.default(n, n))
    eng.set_option("stream_years", 0)       # (by launches)
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
    x = eng.axpby(1.0, x, 1.0, eng.comp_fcn(x)[0])
    ref = None
    for stage in (-1, 0, 1, 2, -1, 1):
        eng.set_option("jac_stage", stage)
        fx, st, _ = eng.comp_fcn(x)
        res = eng.download(fx)
        if ref is None:
            ref = res
        margin = float(np.max(np.abs(res - ref) / (1.0e-6 + 1.0e-3 * np.abs(ref))))
        print(f"n={n} jac_stage={stage}: {st['seconds']:.4f} s, steps {st['nsteps']}, rejected {st['nrejected']}, Newton {st['nnewton']} "
              f"({st['nnewton'] / st['nsteps']:.2f}/step), launches {st['nlaunch']}, |dF|/tol vs jac_stage -1: {margin:.3f}", flush=True)
    eng.close()
