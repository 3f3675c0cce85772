"""probe: preconditioner set-up and apply with the round-1 kernels (option pc_valu 1: VALU rank-32 update, 8-byte
mat-vec loads) and the round-2 ones (fp64 MFMA update, 16-byte loads with the row in flight), same context"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
eng = iage_engine(Grid2d.default(n, n))
v = eng.upload(np.random.default_rng(0).standard_normal((2, n, n)))
res = {}
for valu, chain in ((1, 0), (0, 0), (1, 0), (0, 0)):
    eng.set_option("pc_valu", valu)
    t0 = time.perf_counter()
    eng.precond_setup()
    eng.sync()
    setup = time.perf_counter() - t0
    out = eng.precond_apply(v)
    eng.sync()
    eng.timer_begin()
    for _ in range(5):
        eng.precond_apply(v, out=out)
    ms = eng.timer_end() / 5
    res[(valu, 0)] = eng.download(out)
    nbytes = 2.0 * n * 2 * (3 * n) ** 2 * 8.0
    print(f"pc_valu={valu}: setup {setup:.3f} s, apply {ms:.3f} ms = {nbytes / ms / 1e6:.0f} GB/s "
          f"({nbytes / ms / 1e6 / 8000:.3f} of the HBM peak)", flush=True)
for key in ((0, 0),):
    print("apply results,", key, "vs round-1 kernels: max rel diff",
          float(np.max(np.abs(res[key] - res[(1, 0)])) / np.max(np.abs(res[(1, 0)]))))
