import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from nk_ooc_amd.engine import forced_engine
from nk_ooc_amd.grid import Grid2d
from oracle import radau
from oracle.grid import default_axes
from oracle.model import Forced, Py2dModel
g = np.load("tests/golden/forced_decay_22x9.npz")
nz, ny = 22, 9
depth, ypos = default_axes(nz, ny)
tm = Forced(Py2dModel(depth, ypos), "none", 0.0, "decay", 1.0e-8)
want, solver = radau.comp_fcn(tm, g["y0"], return_solver=True)
sched = np.array(solver.schedule)
print("steps", len(sched), "n_iter hist", np.bincount(sched[:, 3].astype(int)))
info = {"forced_surf_restore_opt": "none", "forced_sms_opt": "decay", "forced_sms_decay_rate": "1e-8"}
for lt in [1e-4, 1e-7, 1e-10, 1e-13, 1e-15]:
    eng = forced_engine(Grid2d.default(nz, ny), info, lin_tol=lt)
    fx, st, _ = eng.comp_fcn(eng.upload(g["y0"]), replay=sched)
    got = eng.download(fx).reshape(-1)
    print(lt, "replay err", np.max(np.abs(got - want)) / np.max(np.abs(want)), "sweeps", st["nsweeps"])
# partial replays: error growth
eng = forced_engine(Grid2d.default(nz, ny), info, lin_tol=1e-13)
for nst in [100, 110, 120, 130, 140, 150, 160, 163, 164, 165, 166]:
    sol = radau.RadauOracle(tm.comp_tend, tm.comp_jacobian, 0.0, g["y0"], 365 * 86400.0, max_step=0.01 * 365 * 86400.0)
    yo = sol.run_replay(solver.schedule[:nst])
    # GPU: replay first nst steps: comp_fcn computes y(t_end) - x with final dense eval at last t
    fx, st, _ = eng.comp_fcn(eng.upload(g["y0"]), replay=sched[:nst])
    got = eng.download(fx).reshape(-1) + g["y0"]
    r = solver.schedule[nst - 1]
    print(nst, "partial err", np.max(np.abs(got - yo)) / np.max(np.abs(yo)), "t/T=%.4f h/T=%.2e n_iter=%d t_jac/T=%.4f h_lu/h=%.3f" % (r[0] / (365 * 86400.0), r[2] / (365 * 86400.0), r[3], r[4] / (365 * 86400.0), r[5] / r[2]))
