"""probe: accepted-step schedules recorded by the host-controlled and by the persistent integrator, same state"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 104
eng = iage_engine(Grid2d.default(n, n))
col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
fx_h, st_h, sched_h = eng.comp_fcn(x, record=True)
eng.set_option("device_ctl", 3)
fx_p, st_p, sched_p = eng.comp_fcn(x, record=True)
print("results identical:", np.array_equal(eng.download(fx_h), eng.download(fx_p)), len(sched_h), len(sched_p))
for key in ("nfev", "njev", "nlu", "nsteps", "nrejected", "nnewton", "nsolve", "nsweeps"):
    print(key, st_h[key], st_p[key])
m = min(len(sched_h), len(sched_p))
diff = np.argwhere(sched_h[:m] != sched_p[:m])
print("differing entries:", len(diff))
for row, colm in diff[:12]:
    print(row, colm, repr(sched_h[row]), repr(sched_p[row]))
eng.set_option("device_ctl", 0)
for name, sched, ref in (("host", sched_h, fx_h), ("persistent", sched_p, fx_p)):
    fx_r, _, _ = eng.comp_fcn(x, replay=sched)
    a, b = eng.download(fx_r), eng.download(ref)
    print(name, "replay vs its free run:", float(np.max(np.abs(a - b)) / np.max(np.abs(b))))
