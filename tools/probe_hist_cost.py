"""what the 61 history samples of a forward year cost: free-running year of iage with and without them (command streams)"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

YEAR = 365.0 * 86400.0
for n in [int(a) for a in sys.argv[1:]] or [416]:
    eng = iage_engine(Grid2d.default(n, n))
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
    t_eval = np.linspace(0.0, YEAR, 61)
    eng.comp_fcn(x)
    plain, hist = [], []
    for _ in range(3):
        t0 = time.perf_counter(); fx, st, _s = eng.comp_fcn(x); plain.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); fh, sth, h = eng.comp_fcn_hist(x, t_eval); hist.append(time.perf_counter() - t0)
    same = np.array_equal(eng.download(fx), eng.download(fh))
    print(f"{n}^2: free-running year {min(plain):.4f} s ({st['nlaunch']} launches), with 61 history samples {min(hist):.4f} s "
          f"({sth['nlaunch']} launches): {1e3 * (min(hist) - min(plain)) / 61:.3f} ms per sample; same F(x): {same}; "
          f"samples checksum {float(np.sum(h)):.12e}", flush=True)
    eng.close()
