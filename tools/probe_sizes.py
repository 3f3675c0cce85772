"""probe: forward year of iage at the BASELINE grid sizes on the GPU, and the oracle (NumPy + SciPy SuperLU
restatement of the reference's comp_fcn, one host thread) at the sizes it finishes in about a minute"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

cpu_sizes = [int(v) for v in os.environ.get("NK2D_CPU_SIZES", "26,52").split(",") if v]
for n in (26, 52, 104, 208, 416):
    grid = Grid2d.default(n, n)
    eng = iage_engine(grid)
    col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    y0 = np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy()
    x = eng.upload(y0)
    x = eng.axpby(1.0, x, 1.0, eng.comp_fcn(x)[0])
    t0 = time.time()
    fx, st, _ = eng.comp_fcn(x)
    gpu = time.time() - t0
    line = f"n={n}: GPU {gpu:.3f} s per forward year ({st['nsteps']} steps, {st['nnewton']} Newton iterations, {st['nlaunch']} launches)"
    if n in cpu_sizes:
        from helpers import oracle_iage
        from oracle import radau

        _, tm = oracle_iage(n, n)
        xh = eng.download(x).reshape(-1)
        t0 = time.time()
        want, solver = radau.comp_fcn(tm, xh, return_solver=True)
        cpu = time.time() - t0
        def deviation(res):
            """largest difference from the oracle in units of the reference CI tolerance (atol 1e-6, rtol 1e-3)"""
            return float(np.max(np.abs(res - want) / (1.0e-6 + 1.0e-3 * np.abs(want))))

        line += f"; CPU oracle {cpu:.1f} s ({solver.stats.nfev} nfev) -> {cpu / gpu:.0f} x; deviation / CI tolerance {deviation(eng.download(fx).reshape(-1)):.2f}"
        eng.set_option("jac_fresh", 0)
        fx0, st0, _ = eng.comp_fcn(x)
        line += f" (SciPy's Jacobian reuse: {deviation(eng.download(fx0).reshape(-1)):.2f}, {st0['nfev']} nfev)"
    print(line, flush=True)
    eng.close()
