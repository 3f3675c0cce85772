"""probe: time one forward year on the GPU at several grid sizes (development aid)"""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ooc_amd.engine import iage_engine
from nk_ooc_amd.grid import Grid2d

import os
sizes = [int(a) for a in sys.argv[1:]] or [26, 104, 416]
FRESH = float(os.environ.get("NK2D_JAC_FRESH", "0"))
for n in sizes:
    grid = Grid2d.default(n, n)
    eng = iage_engine(grid)
    eng.set_option("jac_fresh", FRESH)
    eng.set_option("factor_fp32", float(os.environ.get("NK2D_F32", "0")))
    col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    y0 = np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy()
    x = eng.upload(y0)
    t0 = time.time()
    fx, st, sched = eng.comp_fcn(x, record=True)
    dt = time.time() - t0
    f = eng.download(fx)
    print(f"n={n} wall={dt:.3f}s norm={np.sqrt(np.mean(f**2)):.6e} " + " ".join(f"{k}={v}" for k, v in st.items() if k != 'seconds'), flush=True)
    h = sched[:, 2] / (365 * 86400.0)
    print(f"   h/yr min={h.min():.2e} med={np.median(h):.2e} max={h.max():.2e} sweeps/solve={st['nsweeps']/max(st['nsolve'],1)*2:.1f} us/launch={dt/st['nlaunch']*1e6:.2f}", flush=True)
    if os.environ.get("NK2D_NO_REPLAY"):
        continue
    t0 = time.time()
    fx2, st2, _ = eng.comp_fcn(x, replay=sched)
    print(f"   replay wall={time.time()-t0:.3f}s launches={st2['nlaunch']} diff={np.max(np.abs(eng.download(fx2)-f)):.2e}", flush=True)
