"""probe: the reference's CI case ci_py_driver_2d_iage_column_regions (20 x 3, independent columns) on the GPU, one Krylov
solve, files against the reference's committed baselines in units of its CI tolerances, for the Jacobian at the step
start (jac_stage -1) and at the second stage time (1)

    NK2D_JAC_STAGE=-1 python tools/probe_ci_case.py ; NK2D_JAC_STAGE=1 python tools/probe_ci_case.py
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_krylov as tk  # noqa: E402


class _Tmp:
    def __init__(self, path):
        self.path = path

    def __str__(self):
        return self.path


def ratio(got, want, rtol, atol):
    return float(np.max(np.abs(got - want) / (atol + rtol * np.abs(want))))


from nk_ooc_amd.krylov_solver import KrylovSolver  # noqa: E402

work = tempfile.mkdtemp()
d = os.path.join(tk.BASE, "ci_py_driver_2d_iage_column_regions")
cfg, ModelState = tk._setup_run(_Tmp(work), 20, 3, {"max_abs_vvel": "0.0", "horiz_mix_coeff": "0.0"})
iterate = ModelState(os.path.join(d, "init_iterate.nc"))
fcn = iterate.comp_fcn(os.path.join(work, "fcn_00.nc"), None)
solverinfo = dict(cfg["solverinfo"])
solverinfo["Krylov_workdir"] = os.path.join(work, "krylov_00")
solver = KrylovSolver(iterate, solverinfo, resume=False, rewind=False, hist_fname=None)
solver.solve(os.path.join(work, "increment_00.nc"), fcn)
kdir = solverinfo["Krylov_workdir"]
print("NK2D_JAC_STAGE =", os.environ.get("NK2D_JAC_STAGE", "(default)"), " forward year:", ModelState.last_stats[0])
for name, where, rtol, atol in (("precond_fcn_00.nc", kdir, 2.0e-3, 2.0e-9),
                                ("basis_00.nc", kdir, 1.0e-7, 5.0e-5), ("perturb_fcn_w_raw_00.nc", kdir, 1.0e-7, 5.0e-6),
                                ("krylov_res_00.nc", kdir, 1.9e-2, 2.0e-9), ("increment_00.nc", work, 1.9e-2, 2.0e-9)):
    got, want = tk._read_state(os.path.join(where, name)), tk._read_state(os.path.join(d, name))
    r = np.abs(got - want) / (atol + rtol * np.abs(want))
    k = int(np.argmax(r))
    print(f"  {name}: max |got - want| / (atol + rtol |want|) = {r.max():.3f} at entry {k} (got {got[k]:.6e}, want {want[k]:.6e}); "
          f"entries over 1: {(r > 1).sum()} of {r.size}")
