"""GPU parity of the preconditioner, the region-weighted algebra and the Krylov solver
(through the ModelState / KrylovSolver mirrors -> C ABI -> HIP)."""
import os

import numpy as np
import pytest

from helpers import oracle_iage, rel_err
from oracle import krylov
from oracle.model import apply_precond_stable

pytestmark = pytest.mark.gpu

BASE = os.path.join(os.path.dirname(__file__), "golden", "ref_baselines")


def make_engine(nz, ny, vv=0.1, kh=1000.0, **kw):
    from nk_ooc_amd.engine import iage_engine
    from nk_ooc_amd.grid import Grid2d

    return iage_engine(Grid2d.default(nz, ny, vv, kh), **kw)


def isclose_all(got, want, rtol=1.0e-7, atol=2.0e-9):
    return bool(np.all(np.isclose(got, want, rtol=rtol, atol=atol)))


@pytest.mark.parametrize("nz,ny,vv,kh", [(26, 26, 0.1, 1000.0), (20, 3, 0.0, 0.0), (70, 40, 0.1, 1000.0)])
def test_precond_apply(nz, ny, vv, kh):
    eng = make_engine(nz, ny, vv, kh)
    _, tm = oracle_iage(nz, ny, vv, kh)
    rng = np.random.default_rng(3)
    v = rng.standard_normal(2 * nz * ny)
    got = eng.download(eng.precond_apply(eng.upload(v))).reshape(-1)
    want = apply_precond_stable(tm, v)
    assert rel_err(got, want) < 1e-9, rel_err(got, want)


@pytest.mark.parametrize("nz,ny,vv,kh", [(26, 26, 0.1, 1000.0), (20, 3, 0.0, 0.0), (70, 40, 0.1, 1000.0), (130, 9, 0.1, 1000.0)])
def test_precond_single_precision_storage(nz, ny, vv, kh):
    """option "pc_fp32": the Schur inverses are kept in single precision (half the HBM: tc ny (3 nz)^2 x 4 bytes, 83 GB
    instead of 166 GB at 832 x 832) and an apply is refined once against the exact block tridiagonal operator -- the same
    1e-9 against the oracle's stable form as with double precision storage; without the refinement the single precision
    shows"""
    _, tm = oracle_iage(nz, ny, vv, kh)
    rng = np.random.default_rng(3)
    v = rng.standard_normal(2 * nz * ny)
    want = apply_precond_stable(tm, v)
    eng = make_engine(nz, ny, vv, kh)
    eng.set_option("pc_fp32", 1)
    got = eng.download(eng.precond_apply(eng.upload(v))).reshape(-1)
    assert rel_err(got, want) < 1e-9, rel_err(got, want)
    eng.set_option("pc_refine", 0)
    raw = eng.download(eng.precond_apply(eng.upload(v))).reshape(-1)
    assert 1e-9 < rel_err(raw, want) < 1e-3, rel_err(raw, want)
    eng.set_option("pc_refine", 2)
    twice = eng.download(eng.precond_apply(eng.upload(v))).reshape(-1)
    assert rel_err(twice, want) < 1e-9
    eng.close()


@pytest.mark.parametrize("nz,ny", [(26, 26), (70, 40), (130, 9), (416, 3)])
def test_precond_one_launch_panel_steps_are_the_two_launches(nz, ny):
    """option "pc_fused": a panel step of the Gauss-Jordan inversions as ONE launch (k_pc_gj_step: the inverse of the pivot
    block comes from the step before, whose first workgroup inverted it beside the streaming update) against the two launches
    of rounds 1 - 3 (k_pc_gj_rows + k_pc_gj_update_mfma) -- the same operations in the same order, so the applies agree bit
    for bit; block sizes with a partial last panel included (3 nz = 78, 210, 390; 1248 = 39 full panels, where the one
    launch is the default), and against the oracle's stable form as before"""
    _, tm = oracle_iage(nz, ny)
    rng = np.random.default_rng(8)
    v = rng.standard_normal(2 * nz * ny)
    got = {}
    for fused in (2, 0):
        eng = make_engine(nz, ny)
        eng.set_option("pc_fused", fused)
        got[fused] = eng.download(eng.precond_apply(eng.upload(v))).reshape(-1)
        eng.close()
    assert np.array_equal(got[0], got[2])
    if nz * ny <= 3000:
        assert rel_err(got[2], apply_precond_stable(tm, v)) < 1e-9


def test_precond_golden_within_reference_noise(golden_dir):
    """against the reference formula's output: agreement is bounded by the reference's own
    roundoff sensitivity (tests/test_oracle_precond.py), i.e. the CI tolerance 2e-3"""
    g = np.load(f"{golden_dir}/precond_26x26.npz")
    eng = make_engine(26, 26)
    got = eng.download(eng.precond_apply(eng.upload(g["v"]))).reshape(-1)
    P = 26 * 26
    for tr in range(2):
        s = slice(tr * P, (tr + 1) * P)
        assert np.linalg.norm(got[s] - g["res"][s]) / np.linalg.norm(g["res"][s]) < 2.0e-3


@pytest.mark.parametrize("nz,ny,column_regions", [(26, 26, False), (20, 7, True), (130, 9, True)])
def test_region_algebra(nz, ny, column_regions):
    eng = make_engine(nz, ny)
    model, tm = oracle_iage(nz, ny)
    weight = np.outer(model.depth.delta, model.ypos.delta)
    mask = np.ones((nz, ny), dtype=np.int32)
    if column_regions:
        mask[:] = np.arange(1, ny + 1)[None, :]
        mask[nz // 2:, 0] = 0  # some cells outside every region
    eng.set_region(mask, weight)
    reg = krylov.Regions(mask, weight)
    mod = krylov.OracleModule(tm, reg)
    rng = np.random.default_rng(4)
    a, b = rng.standard_normal((2, 2 * nz * ny))
    ad, bd = eng.upload(a), eng.upload(b)
    assert rel_err(eng.dot(ad, bd), mod.dot(a, b)) < 1e-13
    coef = rng.standard_normal(reg.nreg)
    coef2 = rng.standard_normal(reg.nreg)
    assert np.array_equal(eng.download(eng.scale(ad, coef)).reshape(-1), mod.scale(a, coef))
    want = mod.scale(a, coef) + mod.scale(b, coef2)
    assert np.array_equal(eng.download(eng.axpby(coef, ad, coef2, bd)).reshape(-1), want)
    assert np.array_equal(eng.download(eng.diff_scale(ad, bd, coef)).reshape(-1), mod.scale(a - b, coef))
    assert np.array_equal(eng.download(eng.apply_region_mask(ad.copy())).reshape(-1), mod.mask_out(a))
    # lin_comb and modified Gram-Schmidt over a small basis
    vecs = rng.standard_normal((4, 2 * nz * ny))
    cf = rng.standard_normal((1, 4, reg.nreg))
    want = krylov.lin_comb([mod], cf, [[v] for v in vecs])[0]
    dv = [eng.upload(v) for v in vecs]
    assert np.array_equal(eng.download(eng.lin_comb(dv, cf[0])).reshape(-1), want)
    h_want, w_want = krylov.mod_gram_schmidt([mod], [a], [[v] for v in vecs])
    wd = eng.upload(a)
    h_got = eng.mgs(wd, dv)
    assert rel_err(h_got, h_want[0]) < 1e-12
    assert rel_err(eng.download(wd).reshape(-1), w_want[0]) < 1e-12


def _read_state(fname):
    from nk_ooc_amd import ncio

    data, _ = ncio.read_file(fname, ["iage", "iage_slow_rest"])
    return np.stack([data["iage"], data["iage_slow_rest"]]).reshape(-1)


def _setup_run(tmp_path, nz, ny, extra_modelinfo=None, extra_solverinfo=None):
    from nk_ooc_amd.model_config import ModelConfig
    from nk_ooc_amd.model_state import ModelState
    from nk_ooc_amd.setup_solver import gen_grid_vars_file, make_config

    cfg = make_config(str(tmp_path), nz, ny, extra_modelinfo=extra_modelinfo,
                      extra_solverinfo=extra_solverinfo)
    gen_grid_vars_file(cfg["modelinfo"])
    ModelState.reset_class()
    ModelState.model_config_obj = ModelConfig(cfg["modelinfo"])
    return cfg, ModelState


def test_krylov_column_regions_vs_reference_baselines(tmp_path):
    """the ci_py_driver_2d_iage_column_regions case end to end on the GPU, compared with the
    reference's committed files at the tolerances of its CI script"""
    import json

    from nk_ooc_amd.krylov_solver import KrylovSolver

    d = os.path.join(BASE, "ci_py_driver_2d_iage_column_regions")
    cfg, ModelState = _setup_run(tmp_path, 20, 3, {"max_abs_vvel": "0.0", "horiz_mix_coeff": "0.0"})
    iterate = ModelState(os.path.join(d, "init_iterate.nc"))
    fcn = iterate.comp_fcn(os.path.join(str(tmp_path), "fcn_00.nc"), None)
    solverinfo = dict(cfg["solverinfo"])
    solverinfo["Krylov_workdir"] = os.path.join(str(tmp_path), "krylov_00")
    solver = KrylovSolver(iterate, solverinfo, resume=False, rewind=False, hist_fname=None)
    inc = solver.solve(os.path.join(str(tmp_path), "increment_00.nc"), fcn)
    kdir = solverinfo["Krylov_workdir"]
    assert isclose_all(_read_state(os.path.join(kdir, "precond_fcn_00.nc")),
                       _read_state(os.path.join(d, "precond_fcn_00.nc")), rtol=2.0e-3)
    assert isclose_all(_read_state(os.path.join(kdir, "basis_00.nc")),
                       _read_state(os.path.join(d, "basis_00.nc")), atol=5.0e-5)
    assert isclose_all(_read_state(os.path.join(kdir, "perturb_fcn_w_raw_00.nc")),
                       _read_state(os.path.join(d, "perturb_fcn_w_raw_00.nc")), atol=5.0e-6)
    assert isclose_all(_read_state(os.path.join(kdir, "krylov_res_00.nc")),
                       _read_state(os.path.join(d, "krylov_res_00.nc")), rtol=1.9e-2)
    # the same four comparisons through the comparer the reference's CI script calls (also checks
    # dimensions, variable names and attributes), scripts/ci_py_driver_2d_iage_column_regions.sh
    from nk_ooc_amd import baseline_cmp

    for name, rtol, atol in (("precond_fcn_00.nc", 2.0e-3, 2.0e-9), ("basis_00.nc", 1.0e-7, 5.0e-5),
                             ("perturb_fcn_w_raw_00.nc", 1.0e-7, 5.0e-6), ("krylov_res_00.nc", 1.9e-2, 2.0e-9)):
        assert baseline_cmp.compare(name, kdir, d, rtol=rtol, atol=atol), name
    # checkpoint trail
    state = json.load(open(os.path.join(kdir, "Krylov_state.json")))
    assert state["iteration"] in (1, 2)
    assert "KrylovSolver._solve0" in state["step_log"]
    assert f"00:comp_fcn complete for {kdir}/perturb_fcn_w_raw_00.nc" in state["step_log"]
    assert f"00:comp_jacobian_fcn_state_prod complete for {kdir}/w_raw_00.nc" in state["step_log"]
    assert np.asarray(state["beta"]["__ndarray__"]).shape == (1, 3)
    hm = np.asarray(state["h_mat"]["__ndarray__"])
    assert hm.shape == (1, state["iteration"] + 1, state["iteration"], 3)
    for name in ("precond_00.nc", "w_raw_00.nc", "w_00.nc", "Krylov_stats.nc"):
        assert os.path.exists(os.path.join(kdir, name)), name
    # resume: a second solver over the same directory must not recompute anything
    solver2 = KrylovSolver(iterate, solverinfo, resume=True, rewind=False, hist_fname=None)
    assert solver2.get_iteration() == state["iteration"]
    assert np.array_equal(inc.tracer_modules[0].get_tracer_vals_all().reshape(-1),
                          _read_state(os.path.join(str(tmp_path), "increment_00.nc")))


def test_krylov_26x26_vs_oracle(tmp_path):
    """26x26, single region: GPU Krylov quantities against the oracle's Krylov loop run with
    the same (stable) preconditioner.  Per-iteration quantities are compared; the FD-JVP
    carries integrator noise (two free-running forward years differ by ~1e-6, divided by
    sigma ~ 1e-4 |x|), hence the tolerances."""
    from nk_ooc_amd.krylov_solver import KrylovSolver

    cfg, ModelState = _setup_run(tmp_path, 26, 26, extra_solverinfo={"krylov_max_iter": "2", "krylov_rel_tol": "1.0e-8"})
    model, tm = oracle_iage(26, 26)
    weight = np.outer(model.depth.delta, model.ypos.delta)
    mod = krylov.OracleModule(tm, krylov.Regions(np.ones((26, 26), dtype=np.int32), weight), precond="stable")
    x_host = (np.stack([np.broadcast_to(np.interp(model.depth.mid, [55.0, 200.0], [0.0, 2.0])[:, None], (26, 26))] * 2)
              + 0.1).reshape(-1)
    fcn_host = mod.comp_fcn(x_host)
    _, trace = krylov.krylov_solve([mod], [x_host], [fcn_host], rel_tol=1e-8, max_iter=2)

    ModelState.write_files = False
    try:
        iterate = ModelState("zeros")
        iterate.tracer_modules[0].eng.upload(x_host, out=iterate.tracer_modules[0].vec)
        fcn = iterate.comp_fcn(os.path.join(str(tmp_path), "fcn_00.nc"), None)
        got_fcn = fcn.tracer_modules[0].get_tracer_vals_all().reshape(-1)
        assert np.allclose(got_fcn, fcn_host, rtol=1e-3, atol=1e-6)
        solverinfo = dict(cfg["solverinfo"])
        solverinfo["Krylov_workdir"] = os.path.join(str(tmp_path), "krylov_00")
        solver = KrylovSolver(iterate, solverinfo, resume=False, rewind=False, hist_fname=None)
        solver.solve(os.path.join(str(tmp_path), "increment_00.nc"), fcn)
        st = solver._solver_state
        beta = st.get_value_saved_state("beta")
        h_mat = st.get_value_saved_state("h_mat")
    finally:
        ModelState.write_files = True
    assert st.get_iteration() == 2
    assert rel_err(beta, trace["beta"]) < 1e-4
    assert h_mat.shape == trace["h_mat"][-1].shape == (1, 3, 2, 1)
    assert rel_err(h_mat, trace["h_mat"][-1]) < 2e-2


def test_krylov_resume_from_checkpoint_is_bitwise(tmp_path):
    """the out-of-core contract: a solve interrupted after one iteration and resumed from
    Krylov_state.json + the NetCDF trail (new process state: nothing resident) ends exactly where
    an uninterrupted solve ends"""
    from nk_ooc_amd.krylov_solver import KrylovSolver

    cfg, ModelState = _setup_run(tmp_path, 20, 5, extra_solverinfo={"krylov_rel_tol": "0.0"})
    ModelState.write_files = True
    iterate = ModelState("gen_init_iterate")
    fcn = iterate.comp_fcn(os.path.join(str(tmp_path), "fcn_00.nc"), None)

    def solve(workdir, max_iter, resume):
        info = dict(cfg["solverinfo"], krylov_workdir=workdir, krylov_max_iter=str(max_iter))
        solver = KrylovSolver(iterate, info, resume, False, None)
        inc = solver.solve(os.path.join(workdir, "increment.nc"), fcn)
        return solver, inc.tracer_modules[0].get_tracer_vals_all()

    full_dir = os.path.join(str(tmp_path), "full")
    _, full = solve(full_dir, 2, False)
    part_dir = os.path.join(str(tmp_path), "part")
    first, _ = solve(part_dir, 1, False)
    assert first.get_iteration() == 1
    # the solver object of the interrupted run did not dump basis_01 (it stopped); the
    # reference writes it only when continuing -- emulate the reinvocation: resume at the point
    # where iteration 0 is complete but the stop test said "continue"
    import json
    state_fname = os.path.join(part_dir, "Krylov_state.json")
    state = json.load(open(state_fname))
    assert state["iteration"] == 1
    ModelState._resident.clear()  # a new process has nothing in HBM
    # iteration 1 needs basis_01: recompute it the way the interrupted loop would have
    # (w_00 orthonormalised against basis_00), from the files only
    w = ModelState(os.path.join(part_dir, "w_00.nc"))
    h = np.asarray(state["h_mat"]["__ndarray__"])
    w.mgs_against([ModelState(os.path.join(part_dir, "basis_00.nc"))])
    w /= h[:, -1, -1, :]
    w.dump(os.path.join(part_dir, "basis_01.nc"), "test")
    ModelState._resident.clear()
    resumed, res = solve(part_dir, 2, True)
    assert resumed.get_iteration() == 2
    assert np.array_equal(res, full)
    h_full = json.load(open(os.path.join(full_dir, "Krylov_state.json")))["h_mat"]
    h_res = json.load(open(state_fname))["h_mat"]
    assert h_full == h_res


def test_zero_iterate_and_masked_cells(tmp_path):
    """sigma = 1 where the iterate's norm is 0 (model_state_base.py:509-512, the ci_zero_iage path),
    and cells with region_mask == 0 stay 0 in comp_fcn results (apply_region_mask)"""
    from nk_ooc_amd import ncio

    cfg, ModelState = _setup_run(tmp_path, 20, 5)
    # punch a hole into the region mask
    gv = cfg["modelinfo"]["grid_vars_fname"]
    data, _ = ncio.read_file(gv)
    eng = ModelState("zeros").tracer_modules[0].eng
    mask = np.array(data["region_mask"])
    mask[3:6, 1] = 0
    weight = np.where(mask == 0, 0.0, data["grid_weight"])
    eng.set_region(mask, weight)
    ModelState.write_files = False
    try:
        zero = ModelState("zeros")
        assert np.all(zero.norm() == 0.0)
        fcn = zero.comp_fcn(os.path.join(str(tmp_path), "f.nc"), None)
        vals = fcn.tracer_modules[0].get_tracer_vals_all()
        assert np.all(vals[:, 3:6, 1] == 0.0) and np.any(vals != 0.0)
        from nk_ooc_amd.solver_state import SolverState

        direction = ModelState("gen_init_iterate")
        direction /= direction.norm()
        st = SolverState("Krylov", os.path.join(str(tmp_path), "k"))
        w = zero.comp_jacobian_fcn_state_prod(fcn, direction, os.path.join(str(tmp_path), "k", "w_raw_00.nc"), st)
        # with sigma = 1: w = F(0 + d) - F(0)
        want = direction.comp_fcn(os.path.join(str(tmp_path), "g.nc"), None) - fcn
        got = w.tracer_modules[0].get_tracer_vals_all()
        ref = want.tracer_modules[0].get_tracer_vals_all()
        assert np.allclose(got, ref, rtol=1e-3, atol=1e-6)
        assert np.all(got[:, 3:6, 1] == 0.0)
    finally:
        ModelState.write_files = True
        ModelState.reset_class()
