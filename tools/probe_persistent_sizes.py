"""probe: forward year of iage under host control and in the persistent kernel at the ladder sizes (default mode)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

for n in [int(v) for v in (sys.argv[1:] or ["26", "52", "104", "208", "416"])]:
    eng = iage_engine(Grid2d.default(n, n))
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
    out = {}
    for ctl in (0, 3, 0, 3):
        eng.set_option("device_ctl", ctl)
        fx, st, _ = eng.comp_fcn(x)
        out[ctl] = (st["seconds"], st["nsteps"], st["nnewton"], st["nsweeps"], eng.download(fx))
    same = np.array_equal(out[0][4], out[3][4])
    print(f"n={n}: host {out[0][0]:.4f} s, persistent {out[3][0]:.4f} s ({out[0][0] / out[3][0]:.2f}x), "
          f"steps {out[0][1]} / {out[3][1]}, Newton {out[0][2]} / {out[3][2]}, identical results: {same}", flush=True)
    eng.close()
