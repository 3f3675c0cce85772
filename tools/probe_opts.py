"""probe: effect of lin_tol / sweep_wpb on the forward-year time and on replay parity"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from nk_ooc_amd.engine import iage_engine
from nk_ooc_amd.grid import Grid2d

def run(n, lin_tol, wpb, sched=None, ref=None):
    grid = Grid2d.default(n, n)
    eng = iage_engine(grid, lin_tol=lin_tol)
    eng.set_option("sweep_wpb", wpb)
    col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    y0 = np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy()
    x = eng.upload(y0)
    t0 = time.time(); fx, st, sc = eng.comp_fcn(x, record=True); dt = time.time() - t0
    out = f"n={n} lin_tol={lin_tol:g} wpb={wpb} free={dt:.3f}s sweeps={st['nsweeps']} steps={st['nsteps']}"
    if sched is not None:
        t0 = time.time(); fx2, st2, _ = eng.comp_fcn(x, replay=sched); dt2 = time.time() - t0
        f2 = eng.download(fx2)
        out += f" replay={dt2:.3f}s"
        if ref is not None:
            out += f" replay_err={np.max(np.abs(f2-ref))/np.max(np.abs(ref)):.2e}"
    print(out, flush=True)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
grid = Grid2d.default(n, n)
eng0 = iage_engine(grid, lin_tol=1e-15)
col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
y0 = np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy()
x0 = eng0.upload(y0)
_, _, sched = eng0.comp_fcn(x0, record=True)
ref = eng0.download(eng0.comp_fcn(x0, replay=sched)[0])
for lt in [1e-13, 1e-6, 1e-4, 1e-3, 1e-2]:
    for wpb in ([4, 2, 1] if lt == 1e-13 else [4]):
        run(n, lt, wpb, sched, ref)
