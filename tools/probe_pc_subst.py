"""probe: the block substitution of a preconditioner apply as 2 nb - 1 launches (option pc_one_launch 0) and as one
resident launch (k_pc_subst, the default): same bits, time per apply.  iage (one tracer, three intervals) and phosphorus
(nk2d_shift_solve, three tracers in one system)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nk_ooc_amd.engine import iage_engine, phosphorus_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [26, 104, 416]
for n in sizes:
    eng = iage_engine(Grid2d.default(n, n))
    v = eng.upload(np.random.default_rng(0).standard_normal(eng.shape))
    eng.precond_setup()
    res, ms = {}, {}
    for one in (0, 1, 0, 1):
        eng.set_option("pc_one_launch", one)
        out = eng.precond_apply(v)
        eng.sync()
        eng.timer_begin()
        for _ in range(10):
            eng.precond_apply(v, out=out)
        ms[one] = eng.timer_end() / 10
        res[one] = eng.download(out)
    print(f"iage {n}x{n}: launches {ms[0]:.3f} ms, one launch {ms[1]:.3f} ms per apply ({ms[0] / ms[1]:.2f}x); "
          f"bit-identical {bool(np.array_equal(res[0], res[1]))}; one-launch applies {eng.counter('pc_one_launch_applies')}",
          flush=True)
    del eng
for n in sizes:
    if n > 416:
        continue
    eng = phosphorus_engine(Grid2d.default(n, n))
    rng = np.random.default_rng(1)
    x = eng.upload(1.0 + 0.1 * rng.standard_normal(eng.shape))
    v = eng.upload(rng.standard_normal(eng.shape))
    eng.set_lin_state(x)
    eng.shift_factor(0.0, 1.0e5, [1.0])
    res, ms = {}, {}
    for one in (0, 1, 0, 1):
        eng.set_option("pc_one_launch", one)
        out = eng.shift_solve(0, v)
        eng.sync()
        eng.timer_begin()
        for _ in range(10):
            eng.shift_solve(0, v, out=out)
        ms[one] = eng.timer_end() / 10
        res[one] = eng.download(out)
    print(f"phosphorus {n}x{n}: launches {ms[0]:.3f} ms, one launch {ms[1]:.3f} ms per solve ({ms[0] / ms[1]:.2f}x); "
          f"bit-identical {bool(np.array_equal(res[0], res[1]))}", flush=True)
    del eng
