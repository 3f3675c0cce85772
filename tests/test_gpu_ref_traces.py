"""G7 / G8 of SURVEY.md section 8(c) on the GPU: the Krylov solve and the Newton run of iage at 26 x 26 against
traces of the reference's OWN solver classes (tests/golden/krylov_trace_26x26.npz, newton_trace_26x26.npz,
made by tests/golden/gen_ref_traces.py running nk_ooc.nk_driver in the build container).

What can agree how closely: the reference's forward year is reproducible to ~1e-6 only, a
finite-difference JVP divides that by sigma = 1e-4 |x| -- per-iteration Krylov quantities carry a few 1e-4 of
noise relative to |v| = 1, which is why the reference's CI compares them at rtol 2e-3 ... 1.9e-2
(scripts/ci_py_driver_2d_iage_column_regions.sh:58-91) -- and the reference's preconditioner formula itself
moves by 4e-3 under rounding-level perturbations at this size (tests/test_oracle_precond.py).  Every deviation
measured here is written to gpurun_out/r02_ref_trace_deviations.json."""
import json
import os

import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "r02_ref_trace_deviations.json")


def record(key, value):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    data = json.load(open(OUT)) if os.path.exists(OUT) else {}
    data[key] = value
    with open(OUT, "w") as fptr:
        json.dump(data, fptr, indent=1, sort_keys=True)


def _setup(tmp_path, n, **solverinfo):
    from nk_ooc_amd.model_config import ModelConfig
    from nk_ooc_amd.model_state import ModelState
    from nk_ooc_amd.setup_solver import gen_grid_vars_file, make_config

    cfg = make_config(str(tmp_path), n, n, extra_solverinfo=solverinfo)
    gen_grid_vars_file(cfg["modelinfo"])
    ModelState.reset_class()
    ModelState.write_files = True
    ModelState.model_config_obj = ModelConfig(cfg["modelinfo"])
    return cfg, ModelState


def _state_from(ModelState, vals):
    state = ModelState("zeros")
    tms = state.tracer_modules[0]
    tms.eng.upload(vals, out=tms.vec)
    return state


def test_krylov_solve_against_the_references_own(tmp_path, golden_dir):
    from nk_ooc_amd.krylov_solver import KrylovSolver

    g = np.load(f"{golden_dir}/krylov_trace_26x26.npz")
    n, iters = int(g["n"]), int(g["k0_iterations"])
    cfg, ModelState = _setup(tmp_path, n, krylov_rel_tol="2.0e-4", krylov_max_iter=str(iters))
    iterate = _state_from(ModelState, g["iterate"][0])
    fcn = iterate.comp_fcn(os.path.join(str(tmp_path), "fcn_00.nc"), None)
    got_fcn = fcn.tracer_modules[0].get_tracer_vals_all()
    assert np.allclose(got_fcn, g["fcn"][0], rtol=1.0e-3, atol=1.0e-6)          # the CI tolerance of fcn files
    # the solve starts from the REFERENCE's fcn, so that iteration 0 compares like with like
    fcn = _state_from(ModelState, g["fcn"][0])
    solverinfo = dict(cfg["solverinfo"], krylov_workdir=os.path.join(str(tmp_path), "krylov_00"))
    solver = KrylovSolver(iterate, solverinfo, False, False, None)
    inc = solver.solve(os.path.join(str(tmp_path), "increment_00.nc"), fcn)
    state = solver._solver_state
    beta, h_mat = state.get_value_saved_state("beta"), state.get_value_saved_state("h_mat")
    assert solver.get_iteration() == iters                     # same stopping decisions as the reference took
    dev = {"beta_rel": rel_err(beta, g["k0_beta"]),
           "h_mat_abs": float(np.max(np.abs(h_mat - g["k0_h_mat"]))),
           "increment_rel": rel_err(inc.tracer_modules[0].get_tracer_vals_all(), g["increment"][0])}
    kdir = solverinfo["krylov_workdir"]
    from nk_ooc_amd import ncio

    def read(name):
        data, _ = ncio.read_file(os.path.join(kdir, name), ["iage", "iage_slow_rest"])
        return np.stack([data["iage"], data["iage_slow_rest"]])

    dev["precond_fcn_rel"] = rel_err(read("precond_fcn_00.nc"), g["k0_precond_fcn"])
    for j in range(iters):
        dev[f"basis_{j}_abs"] = float(np.max(np.abs(read(f"basis_{j:02}.nc") - g["k0_basis"][j])))
        dev[f"w_raw_{j}_rel"] = rel_err(read(f"w_raw_{j:02}.nc"), g["k0_w_raw"][j])
        dev[f"krylov_res_{j}_rel"] = rel_err(read(f"krylov_res_{j:02}.nc"), g["k0_krylov_res"][j])
    from scipy.io import netcdf_file

    with netcdf_file(os.path.join(kdir, "Krylov_stats.nc"), "r", mmap=False) as fptr:
        resid = np.array(fptr.variables["precond_resid_norm_iage"].data)
    dev["resid_norm"] = {"got": resid[:, 0].tolist(), "reference": g["k0_precond_resid_norm"][:, 0].tolist()}
    record("krylov_26x26", dev)
    # Measured (gpurun_out/r02_ref_trace_deviations.json): M^-1 fcn 1e-2, beta 3.5e-4, Hessenberg 1.7e-2, iterates
    # x_j 9.5e-3 / 1.6e-3 / 1.7e-3, increment 1.7e-3, residual history within 9 %.  The reference's preconditioner
    # FORMULA is what limits this (its output moves by 4e-3 under 1e-16 perturbations of its own matrix entries at
    # 26 x 26, tests/test_oracle_precond.py; the library solves the same operator in a stable form): the first Arnoldi
    # vector inherits that 1 %, the later ones -- individually -- rotate within an almost identical Krylov space
    # (their deviations are recorded, not asserted), while everything the solver RETURNS agrees at the CI tolerances.
    assert dev["precond_fcn_rel"] < 2.0e-2 and dev["beta_rel"] < 2.0e-3
    assert dev["h_mat_abs"] < 3.0e-2
    assert dev["basis_0_abs"] < 2.0e-2 * np.max(np.abs(g["k0_basis"][0]))
    for j in range(iters):
        assert dev[f"krylov_res_{j}_rel"] < 1.9e-2, j          # the CI's rtol for krylov_res / increment files
    assert dev["increment_rel"] < 1.9e-2
    # residual history: same decay (the last value sits at the FD noise floor, beta * 1e-4)
    assert np.allclose(resid[:2, 0], g["k0_precond_resid_norm"][:2, 0], rtol=0.15)
    assert resid[2, 0] < 3.0 * g["k0_precond_resid_norm"][2, 0] + 1.0e-4 * beta[0, 0]
    ModelState.reset_class()


def test_newton_run_against_the_references_own(tmp_path, golden_dir):
    """the north star's "same converged Newton iterate as the reference CPU path on identical input / cfg files":
    the driver mirror from the reference's init_iterate with the reference's default cfg"""
    from nk_ooc_amd import ncio, nk_driver
    from nk_ooc_amd.model_state import ModelState
    from nk_ooc_amd.setup_solver import gen_grid_vars_file, make_config

    g = np.load(f"{golden_dir}/newton_trace_26x26.npz")
    n = int(g["n"])
    workdir = str(tmp_path)
    cfg = make_config(workdir, n, n)
    gen_grid_vars_file(cfg["modelinfo"])
    from nk_ooc_amd.model_config import ModelConfig

    ModelState.reset_class()
    ModelState.write_files = True
    ModelState.model_config_obj = ModelConfig(cfg["modelinfo"])
    init = ModelState("zeros")
    init.tracer_modules[0].eng.upload(g["init_iterate"], out=init.tracer_modules[0].vec)
    os.makedirs(os.path.dirname(cfg["solverinfo"]["init_iterate_fname"]), exist_ok=True)
    init.dump(cfg["solverinfo"]["init_iterate_fname"], "test")
    solver = nk_driver.run(cfg)
    assert solver.converged().all()
    n_newton = solver.get_iteration()
    assert n_newton == int(g["newton_iterations"]) == 2

    def read(name):
        data, _ = ncio.read_file(os.path.join(workdir, name), ["iage", "iage_slow_rest"])
        return np.stack([data["iage"], data["iage_slow_rest"]])

    dev = {}
    for it in range(n_newton + 1):
        dev[f"iterate_{it:02}_rel"] = rel_err(read(f"iterate_{it:02}.nc"), g["iterate"][it])
        fcn_ref = g["fcn"][it]
        dev[f"fcn_{it:02}_abs_over_tol"] = float(np.max(
            np.abs(read(f"fcn_{it:02}.nc") - fcn_ref) / (1.0e-6 + 1.0e-3 * np.abs(fcn_ref))))
    record("newton_26x26", dev)
    assert dev["iterate_00_rel"] == 0.0
    for it in range(1, n_newton + 1):
        assert dev[f"iterate_{it:02}_rel"] < 1.9e-2, it            # the CI's rtol for iterate_01
    # the converged iterates agree far better than the CI tolerance: both are fixed points of the same map
    assert dev[f"iterate_{n_newton:02}_rel"] < 1.0e-3
    # step logs: the same sequence of checkpointed actions, Krylov iteration counts included
    got = json.load(open(os.path.join(workdir, "Newton_state.json")))
    got_log = [s.replace(workdir, "$workdir") for s in got["step_log"]]
    assert got_log == json.loads(str(g["newton_step_log"]))
    for k in range(int(g["krylov_solves"])):
        kstate = json.load(open(os.path.join(workdir, f"krylov_{k:02}", "Krylov_state.json")))
        assert kstate["iteration"] == int(g[f"k{k}_iterations"]), k
        klog = [s.replace(workdir, "$workdir") for s in kstate["step_log"]]
        assert klog == json.loads(str(g[f"k{k}_step_log"])), k
    ModelState.reset_class()
