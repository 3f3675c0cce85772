"""one-launch frozen year on the schedule cache: what lives in LDS (option "frozen_coef_lds", bits: 1 coefficients, 2 W, 4 step block,
8 pivots) and which columns share a workgroup (option "frozen_by_column")"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [416]:
    eng = iage_engine(Grid2d.default(n, n))
    eng.set_option("frozen_alloc_async", 0)
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x0 = np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy()
    x = eng.upload(x0)
    zz = np.linspace(0.0, 1.0, n)
    xp = eng.upload(x0 * (1.0 + 1.0e-4 * np.outer(np.sin(3.0 * zz), np.cos(2.0 * zz))[None]))
    fx, st, sched = eng.comp_fcn(x, record=True)
    ref = None
    for by_col, bits in ((0, 0), (0, 3), (1, 3), (1, 7), (1, 15), (0, 3), (1, 15)):
        eng.set_option("frozen_by_column", by_col)
        eng.set_option("frozen_coef_lds", bits)
        best = min(eng.comp_fcn_frozen(xp, sched)[1]["seconds"] for _ in range(5))
        got = eng.download(eng.comp_fcn_frozen(xp, sched)[0])
        if ref is None:
            ref = got
        print(f"{n}^2 one-launch frozen year, by_column {by_col} lds bits {bits:2d}: {1e3 * best:.2f} ms  same bits {np.array_equal(got, ref)}  "
              f"one-launch years {eng.counter('frozen_persistent_years')}", flush=True)
    eng.close()
