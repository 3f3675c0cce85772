"""G7 / G8 of SURVEY.md section 8(c) on the GPU: the Krylov solve and the Newton run of iage against traces of the
reference's OWN solver classes at 26 x 26, 30 x 30 (the grid of scripts/ci_py_driver_2d_iage.sh:13-14) and 52 x 52
(tests/golden/{krylov,newton}_trace_NxN.npz, made by tests/golden/gen_ref_traces.py running nk_ooc.nk_driver in the build
container) -- in the mode the benchmark runs (products on frozen years, the default) AND with the reference's own
product (two free-running years, NK2D_JVP_FROZEN=0).

What can agree how closely: the reference's forward year is reproducible to ~1e-6 only and its
finite-difference product divides that by sigma = 1e-4 |x| -- per-iteration Krylov quantities of the REFERENCE carry a few
1e-4 ... 1e-3 of noise relative to |v| = 1, which is why its CI compares them at rtol 2e-3 ... 1.9e-2
(scripts/ci_py_driver_2d_iage_column_regions.sh:58-91).  On top of that the reference's preconditioner FORMULA,
(I - A0 A1 A2)^-1 v - v with the product formed explicitly, is a different linear map from the operator it stands for
once rounding has eaten the identity (tests/test_oracle_precond.py: 4e-3 at 26 x 26, 0.7 at 52 x 52 under 1e-16
perturbations of its own matrix entries): the library applies the same operator in a backward-stable form.  Round 3
ISOLATES that: every case also runs with `precond = "reference_formula"` -- a test-only hybrid in which the CPU oracle
(bit for bit the reference's formula, tests/test_ref_traces.py) applies the preconditioner and the GPU does everything
else.  With it the Arnoldi vectors track the reference's to its product noise; with the library's own preconditioner the
returned quantities agree at the CI tolerances and the solver converges in at most the reference's iterations.

Every deviation measured here is written to gpurun_out/r03_ref_trace_deviations.json (committed copy under profiles/)."""
import json
import os

import numpy as np
import pytest

from helpers import oracle_iage, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "r03_ref_trace_deviations.json")
SIZES = [26, 30, 52]
MODES = ["frozen", "free_running"]
PRECONDS = ["library", "reference_formula"]


def record(key, value):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    data = json.load(open(OUT)) if os.path.exists(OUT) else {}
    data[key] = value
    with open(OUT, "w") as fptr:
        json.dump(data, fptr, indent=1, sort_keys=True)


def _wrel(n):
    """relative deviation in the solver's own norm (region-weighted mean of squares, model_config.py:292-315)"""
    model, _ = oracle_iage(n, n)
    weight = np.outer(model.depth.delta, model.ypos.delta)
    weight = weight / weight.sum()

    def wrel(a, b):
        num = np.sqrt(np.sum(weight * (np.asarray(a) - np.asarray(b)) ** 2))
        den = np.sqrt(np.sum(weight * np.asarray(b) ** 2))
        return float(num / (den if den > 0 else 1.0))

    return wrel


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


def _use_reference_formula(monkeypatch, ModelState, n):
    """test-only hybrid: M^-1 v by the CPU oracle's restatement of the reference's formula (iage.py:66-93; bit for bit the
    reference's on the reference's inputs), everything else on the device"""
    _, tm = oracle_iage(n, n)

    def apply_precond_jacobian(self, precond_fname, res_fname, solver_state):
        step = f"apply_precond_jacobian complete for {res_fname}"
        if solver_state is not None and solver_state.step_logged(step):
            return type(self)(res_fname)
        tms = self.tracer_modules[0]
        out = tm.apply_precond(tms.get_tracer_vals_all().reshape(-1))
        res = self._new([tms._like(tms.eng.upload(out))])
        if solver_state is not None:
            solver_state.log_step(step)
        return res.dump(res_fname, "test hybrid: reference preconditioner formula on the CPU")

    monkeypatch.setattr(ModelState, "apply_precond_jacobian", apply_precond_jacobian)


@pytest.mark.parametrize("precond", PRECONDS)
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("n", SIZES)
def test_krylov_solve_against_the_references_own(tmp_path, golden_dir, monkeypatch, n, mode, precond):
    from nk_ooc_amd import ncio
    from nk_ooc_amd.krylov_solver import KrylovSolver

    monkeypatch.setenv("NK2D_JVP_FROZEN", "1" if mode == "frozen" else "0")
    g = np.load(f"{golden_dir}/krylov_trace_{n}x{n}.npz")
    assert int(g["n"]) == n
    iters = int(g["k0_iterations"])
    cfg, ModelState = _setup(tmp_path, n, krylov_rel_tol="2.0e-4", krylov_max_iter=str(iters))
    if precond == "reference_formula":
        _use_reference_formula(monkeypatch, ModelState, n)
    iterate = _state_from(ModelState, g["iterate"][0])
    # F(x) of the device (the products on frozen years repeat ITS accepted steps), inside the CI tolerance of fcn files
    fcn = iterate.comp_fcn(os.path.join(str(tmp_path), "fcn_00.nc"), None)
    got_fcn = fcn.tracer_modules[0].get_tracer_vals_all()
    assert np.allclose(got_fcn, g["fcn"][0], rtol=1.0e-3, atol=1.0e-6)
    solverinfo = dict(cfg["solverinfo"], krylov_workdir=os.path.join(str(tmp_path), "krylov_00"))
    solver = KrylovSolver(iterate, solverinfo, False, False, None)
    inc = solver.solve(os.path.join(str(tmp_path), "increment_00.nc"), fcn)
    state = solver._solver_state
    beta, h_mat = state.get_value_saved_state("beta"), state.get_value_saved_state("h_mat")
    got_iters = solver.get_iteration()
    kdir = solverinfo["krylov_workdir"]

    def read(name):
        data, _ = ncio.read_file(os.path.join(kdir, name), ["iage", "iage_slow_rest"])
        return np.stack([data["iage"], data["iage_slow_rest"]])

    from scipy.io import netcdf_file

    with netcdf_file(os.path.join(kdir, "Krylov_stats.nc"), "r", mmap=False) as fptr:
        resid = np.array(fptr.variables["precond_resid_norm_iage"].data)[:, 0]
    ref_resid = g["k0_precond_resid_norm"][:, 0]
    k = min(got_iters, iters)
    wrel = _wrel(n)
    got_inc = inc.tracer_modules[0].get_tracer_vals_all()
    dev = {"iterations": {"got": got_iters, "reference": iters},
           "beta_rel": rel_err(beta, g["k0_beta"]),
           "h_mat_abs": float(np.max(np.abs(h_mat[:, : k + 1, :k] - g["k0_h_mat"][:, : k + 1, :k]))),
           "increment_rel": rel_err(got_inc, g["increment"][0]), "increment_wrel": wrel(got_inc, g["increment"][0]),
           "precond_fcn_rel": rel_err(read("precond_fcn_00.nc"), g["k0_precond_fcn"]),
           "resid_norm_over_beta": {"got": (resid / beta[0, 0]).tolist(), "reference": (ref_resid / g["k0_beta"][0, 0]).tolist()},
           "frozen_years_rejected": iterate.tracer_modules[0].eng.frozen_fallbacks()}
    for j in range(k):
        for quantity in ("basis", "w_raw", "w", "krylov_res"):
            got_q, ref_q = read(f"{quantity}_{j:02}.nc"), g[f"k0_{quantity}"][j]
            dev[f"{quantity}_{j}_rel"] = rel_err(got_q, ref_q)
            dev[f"{quantity}_{j}_wrel"] = wrel(got_q, ref_q)
    record(f"krylov_{n}x{n}_{mode}_{precond}", dev)
    assert dev["frozen_years_rejected"] == 0
    if precond == "reference_formula":
        # The same preconditioner map on both sides: the same stopping decisions at every size, the residual history to
        # 5 % (measured 3 %), the Hessenberg to 1e-2 (measured 5e-3), the preconditioned product of the FIRST direction --
        # the same vector on both sides to 1e-8 -- and every iterate x_j to 1e-2 in the solver's norm (measured 4e-3 and
        # 5e-3: the reference's own product noise).  The later Arnoldi vectors are each the normalised remainder of a
        # product after its dominant components are projected out, which amplifies that noise: 2 ... 10 % in the solver's
        # norm, on the reference's account as much as on this side's (frozen products halve it) -- recorded, not asserted;
        # the RAW products w_raw_j carry it in the fast modes the preconditioner damps (10 % in the max norm).
        assert got_iters == iters
        assert dev["precond_fcn_rel"] < 1.0e-6 and dev["beta_rel"] < 1.0e-6
        assert np.allclose(resid, ref_resid, rtol=0.05)
        assert dev["h_mat_abs"] < 1.0e-2
        assert dev["w_0_wrel"] < 1.0e-2
        for j in range(iters):
            assert dev[f"krylov_res_{j}_wrel"] < 1.0e-2, j      # (the CI's rtol for krylov_res / increment files: 1.9e-2)
        assert dev["increment_wrel"] < 1.0e-2
    else:
        # The library's preconditioner -- the same operator in a backward-stable form -- is at least as good a
        # preconditioner: never more iterations than the reference took to the same tolerance, and the increment solves the
        # same linear system to that tolerance
        assert got_iters <= iters
        assert resid[got_iters - 1] < 2.0e-4 * beta[0, 0] or got_iters == iters
        assert dev["increment_wrel"] < 1.0e-2                    # measured 7e-4 (26 x 26) ... 3.6e-3 (52 x 52)
        if n == 26:
            # where the reference's formula is still within 1 % of the operator everything the solver RETURNS agrees at the
            # CI tolerances (round 2: M^-1 fcn 1e-2, beta 3.5e-4, Hessenberg 1.7e-2, iterates 1e-2 / 4e-3 / 4e-3)
            assert got_iters == iters
            assert dev["precond_fcn_rel"] < 2.0e-2 and dev["beta_rel"] < 2.0e-3
            assert dev["h_mat_abs"] < 3.0e-2
            for j in range(iters):
                assert dev[f"krylov_res_{j}_rel"] < 1.9e-2, j
            assert np.allclose(resid[:2], ref_resid[:2], rtol=0.15)
    ModelState.reset_class()


@pytest.mark.parametrize("precond", PRECONDS)
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("n", SIZES)
def test_newton_run_against_the_references_own(tmp_path, golden_dir, monkeypatch, n, mode, precond):
    """the north star's "same converged Newton iterate as the reference CPU path on identical input / cfg files":
    the driver mirror from the reference's init_iterate with the reference's default cfg"""
    from nk_ooc_amd import ncio, nk_driver
    from nk_ooc_amd.model_config import ModelConfig
    from nk_ooc_amd.model_state import ModelState
    from nk_ooc_amd.setup_solver import gen_grid_vars_file, make_config

    monkeypatch.setenv("NK2D_JVP_FROZEN", "1" if mode == "frozen" else "0")
    g = np.load(f"{golden_dir}/newton_trace_{n}x{n}.npz")
    assert int(g["n"]) == n and str(g["ended"]) == "converged"
    workdir = str(tmp_path)
    cfg = make_config(workdir, n, n)
    gen_grid_vars_file(cfg["modelinfo"])
    ModelState.reset_class()
    ModelState.write_files = True
    ModelState.model_config_obj = ModelConfig(cfg["modelinfo"])
    if precond == "reference_formula":
        _use_reference_formula(monkeypatch, ModelState, n)
    init = ModelState("zeros")
    init.tracer_modules[0].eng.upload(g["init_iterate"], out=init.tracer_modules[0].vec)
    os.makedirs(os.path.dirname(cfg["solverinfo"]["init_iterate_fname"]), exist_ok=True)
    init.dump(cfg["solverinfo"]["init_iterate_fname"], "test")
    solver = nk_driver.run(cfg)
    assert solver.converged().all()
    n_newton = solver.get_iteration()
    ref_newton = int(g["newton_iterations"])
    ref_krylov = [int(g[f"k{k}_iterations"]) for k in range(int(g["krylov_solves"]))]

    def read(name):
        data, _ = ncio.read_file(os.path.join(workdir, name), ["iage", "iage_slow_rest"])
        return np.stack([data["iage"], data["iage_slow_rest"]])

    got_krylov = []
    for k in range(n_newton):
        kstate = json.load(open(os.path.join(workdir, f"krylov_{k:02}", "Krylov_state.json")))
        got_krylov.append(kstate["iteration"])
    dev = {"newton_iterations": {"got": n_newton, "reference": ref_newton},
           "krylov_iterations": {"got": got_krylov, "reference": ref_krylov},
           "frozen_years_rejected": ModelState._engines["iage"].frozen_fallbacks()}
    for it in range(min(n_newton, ref_newton) + 1):
        dev[f"iterate_{it:02}_rel"] = rel_err(read(f"iterate_{it:02}.nc"), g["iterate"][it])
        fcn_ref = g["fcn"][it]
        dev[f"fcn_{it:02}_abs_over_tol"] = float(np.max(
            np.abs(read(f"fcn_{it:02}.nc") - fcn_ref) / (1.0e-6 + 1.0e-3 * np.abs(fcn_ref))))
    dev["converged_iterate_rel"] = rel_err(read(f"iterate_{n_newton:02}.nc"), g["iterate"][ref_newton])
    dev["converged_iterate_wrel"] = _wrel(n)(read(f"iterate_{n_newton:02}.nc"), g["iterate"][ref_newton])
    record(f"newton_{n}x{n}_{mode}_{precond}", dev)
    assert dev["frozen_years_rejected"] == 0
    assert dev["iterate_00_rel"] == 0.0
    assert n_newton == ref_newton
    # the converged iterates: both are fixed points of the same map to newton_rel_tol = 1e-5; stated rtol 5e-5 in the max
    # norm (measured 2.3e-6 ... 5.9e-6 over the twelve cases, profiles/r03_ref_trace_deviations.json)
    assert dev["converged_iterate_rel"] < 5.0e-5
    for it in range(1, n_newton + 1):
        assert dev[f"iterate_{it:02}_rel"] < 1.9e-2, it            # the CI's rtol for iterate_01
    got = json.load(open(os.path.join(workdir, "Newton_state.json")))
    got_log = [s.replace(workdir, "$workdir") for s in got["step_log"]]
    same_counts = precond == "reference_formula" or n <= 30
    if same_counts:
        # step logs: the same sequence of checkpointed actions, Krylov iteration counts included, string for string
        assert got_log == json.loads(str(g["newton_step_log"]))
        assert got_krylov == ref_krylov
        for k in range(int(g["krylov_solves"])):
            kstate = json.load(open(os.path.join(workdir, f"krylov_{k:02}", "Krylov_state.json")))
            klog = [s.replace(workdir, "$workdir") for s in kstate["step_log"]]
            assert klog == json.loads(str(g[f"k{k}_step_log"])), k
    else:
        # 52 x 52 with the library's preconditioner: the Newton level takes the reference's steps (same log), each Krylov
        # solve at most the reference's iterations (its preconditioner formula is rounding noise in the second tracer)
        assert got_log == json.loads(str(g["newton_step_log"]))
        assert all(a <= b for a, b in zip(got_krylov, ref_krylov))
    ModelState.reset_class()
