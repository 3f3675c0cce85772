"""stream year at 416^2 for several values of option "spec_bias" (what the host queues behind an iteration not yet judged)"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
eng = iage_engine(Grid2d.default(n, n))
eng.set_option("stream_years", 0)       # (by launches)
col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
ref = None
for stream in (0, 1):
    eng.set_option("stream_years", stream)
    for bias in (1.0, 3.0, 10.0, 30.0, 100.0, 1000.0):
        eng.set_option("spec_bias", bias)
        names = ("spec_launches_dropped", "spec_front_launches_dropped", "err_estimates_queued", "err_estimates_dropped")
        best = None
        for rep in range(3):
            c0 = [eng.counter(k) for k in names]
            t0 = time.perf_counter()
            fx, st, sched = eng.comp_fcn(x, record=True)
            wall = time.perf_counter() - t0
            c1 = [eng.counter(k) for k in names]
            if best is None or wall < best[0]:
                best = (wall, [b - a for a, b in zip(c0, c1)])
        got = eng.download(fx)
        if ref is None:
            ref = got
        print(f"{n}^2 stream {stream} spec_bias {bias:7.1f}: {best[0]:.4f} s  dropped whole {best[1][0]} front {best[1][1]} "
              f"err queued {best[1][2]} void {best[1][3]}  same bits {np.array_equal(got, ref)}", flush=True)
eng.close()
