"""probe: forward year of a forced module with file-style forcing (module kind 2) at n x n, against
the decay variant (kind 0) on the same grid"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ooc_amd.engine import ModuleEngine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
grid = Grid2d.default(n, n)
rng = np.random.default_rng(0)
times = (15.0 + 30.0 * np.arange(12)) * 86400.0
phase = 2.0 * np.pi * times / (365.0 * 86400.0)
restore = 1.0 + 0.3 * np.sin(phase)[:, None] * np.linspace(0.5, 1.5, n)[None, :]
sms = 2.0e-8 * (np.cos(phase)[:, None, None] * np.exp(-np.arange(n) / (0.3 * n))[None, :, None]
                + 0.5 * rng.standard_normal((12, n, n)))
rate = 10.0 / grid.depth.delta[0] * (24.0 / 86400.0)
y0 = 0.6 + 0.2 * rng.standard_normal((1, n, n))
cases = {
    "decay (kind 0)": dict(decay_rate=(1.0e-8,)),
    "file restoring + file source (kind 2)": dict(surf_rate=(rate,), module_kind=2, restore_series=(times, restore),
                                                  sms_series=(times, sms)),
    "const restoring + file source + sink threshold (kind 2)": dict(
        surf_rate=(rate,), surf_target=(1.5,), module_kind=2, sms_series=(times, sms), sink_thres=0.5),
}
for name, kw in cases.items():
    eng = ModuleEngine(grid, tc=1, **kw)
    x = eng.upload(y0)
    for rep in range(2):
        t0 = time.time()
        fx, st, _ = eng.comp_fcn(x)
        wall = time.time() - t0
    print(f"{name}: n={n} wall={wall:.3f}s nsteps={st['nsteps']} nnewton={st['nnewton']} nlaunch={st['nlaunch']}", flush=True)
    eng.close()
