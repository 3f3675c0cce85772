"""preconditioner set-up: a panel step of the Gauss-Jordan inversions as one launch (option pc_fused 1, k_pc_gj_step) against
two (0: k_pc_gj_rows + k_pc_gj_update_mfma), same context, alternating; the applies bit for bit"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [416]:
    eng = iage_engine(Grid2d.default(n, n))
    v = eng.upload(np.random.default_rng(0).standard_normal((2, n, n)))
    res = {}
    for fused in (0, 2, 0, 2):
        eng.set_option("pc_fused", fused)
        t0 = time.perf_counter()
        eng.precond_setup()
        eng.sync()
        setup = time.perf_counter() - t0
        res[fused] = eng.download(eng.precond_apply(v))
        print(f"{n} x {n}: pc_fused={fused}: set-up {setup:.3f} s", flush=True)
    print("applies bit for bit:", np.array_equal(res[0], res[2]), flush=True)
    eng.close()
