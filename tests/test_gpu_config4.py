"""BASELINE.json configs[3] in its one-GPU form: the three tracer modules of py_driver_2d together --
`tracer_module_names = iage,phosphorus,forced_{suff}:dye` (the decay variant of `forced` is py_driver_2d's analogue of
test_problem's dye_decay) -- through ModelState / KrylovSolver in the engines' DEFAULT mode
(products on frozen years, Jacobian at the second stage time), three engines = three HIP streams on one GPU, their
forward years running concurrently.  The reference loops over the modules (nk_ooc/py_driver_2d/model_state.py:95-121)
and keeps one Hessenberg / beta per (module, region) (nk_ooc/krylov_solver.py:114-121,159); the CPU oracle does the same."""
import os

import numpy as np
import pytest

from helpers import rel_err
from oracle import krylov
from oracle.grid import default_axes
from oracle.model import Forced, Iage, Phosphorus, Py2dModel

pytestmark = pytest.mark.gpu

DECAY = {"forced_surf_restore_opt": "none", "forced_sms_opt": "decay", "forced_sms_decay_rate": "1.0e-8"}
NAMES = "iage,phosphorus,forced_{suff}:dye"


def _setup(workdir, nz, ny, names=NAMES, **solverinfo):
    from nk_ooc_amd.model_config import ModelConfig
    from nk_ooc_amd.model_state import ModelState
    from nk_ooc_amd.setup_solver import gen_grid_vars_file, make_config

    cfg = make_config(workdir, nz, ny, tracer_module_names=names, extra_modelinfo=DECAY, extra_solverinfo=solverinfo)
    gen_grid_vars_file(cfg["modelinfo"])
    ModelState.reset_class()
    ModelState.model_config_obj = ModelConfig(cfg["modelinfo"])
    return cfg, ModelState


def _structured_dye(ModelState, iterate, nz, ny, seed=3):
    """the forced module's initial iterate is exactly uniform, where the reference's own map amplifies roundoff
    (docs/DESIGN_history_r1-r3.md section 5): give the tracer a smooth positive structure, as the parity tests of `forced` do"""
    rng = np.random.default_rng(seed)
    bump = np.cumsum(np.cumsum(rng.standard_normal((1, nz, ny)), axis=1), axis=2)
    dye = 1.0 + 0.3 * bump / np.max(np.abs(bump))
    tms = iterate.tracer_modules[2]
    tms.eng.upload(dye, out=tms.vec)
    return dye


def test_three_module_krylov(tmp_path):
    from nk_ooc_amd.krylov_solver import KrylovSolver

    nz, ny = 22, 9
    work = str(tmp_path)
    cfg, ModelState = _setup(work, nz, ny, krylov_max_iter="2", krylov_rel_tol="1e-9")
    ModelState.write_files = True
    iterate = ModelState("gen_init_iterate")
    assert [tms.name for tms in iterate.tracer_modules] == ["iage", "phosphorus", "forced_dye"]
    _structured_dye(ModelState, iterate, nz, ny)
    hist_fname = os.path.join(work, "hist_00.nc")
    # as NewtonSolver calls it: the year behind F(x) samples the history the phosphorus preconditioner is made from,
    # and leaves the accepted steps the perturbed years of the products repeat
    fcn = iterate.comp_fcn(os.path.join(work, "fcn_00.nc"), None, hist_fname)
    solverinfo = dict(cfg["solverinfo"], krylov_workdir=os.path.join(work, "krylov_00"))
    solver = KrylovSolver(iterate, solverinfo, False, False, hist_fname)
    solver.solve(os.path.join(work, "increment_00.nc"), fcn)
    beta = solver._solver_state.get_value_saved_state("beta")
    h_mat = solver._solver_state.get_value_saved_state("h_mat")
    assert beta.shape == (3, 1) and h_mat.shape == (3, 3, 2, 1)
    assert solver.get_iteration() == 2
    # every product ran on a frozen year; none was rejected
    assert [tms.eng.frozen_fallbacks() for tms in iterate.tracer_modules] == [0, 0, 0]
    # ---- the oracle with the same three modules ------------------------------------------------------------------------
    depth, ypos = default_axes(nz, ny)
    model = Py2dModel(depth, ypos)
    regions = krylov.Regions(np.ones((nz, ny), dtype=np.int32), np.outer(depth.delta, ypos.delta))
    mods = [krylov.OracleModule(Iage(model), regions, precond="stable"),
            krylov.OracleModule(Phosphorus(model), regions, precond="phosphorus"),
            krylov.OracleModule(Forced(model, "none", 0.0, "decay", 1.0e-8), regions, precond="stable")]
    x = [tms.get_tracer_vals_all().reshape(-1) for tms in iterate.tracer_modules]
    f = [m.comp_fcn(v) for m, v in zip(mods, x)]
    for i in range(3):
        assert np.allclose(fcn.tracer_modules[i].get_tracer_vals_all().reshape(-1), f[i], rtol=1e-3, atol=1e-6), i
    mods[1].precond_po4 = (x[1] + f[1]).reshape(3, nz, ny)[0]
    _, trace = krylov.krylov_solve(mods, x, f, rel_tol=1e-9, max_iter=2)
    # beta carries the preconditioner and F only; the oracle's Hessenberg entries are finite differences of two
    # free-running CPU years (tolerance 1e-6 over sigma = 1e-4 |x|: a few per cent of the largest entry)
    assert rel_err(beta, trace["beta"]) < 1e-3
    want = trace["h_mat"][-1]
    for i in range(3):
        assert rel_err(h_mat[i], want[i]) < 5e-2, (i, h_mat[i].ravel(), want[i].ravel())
    # what the solver returns: the increment of every module against the oracle's, at the CI tolerance of increments
    from nk_ooc_amd import ncio

    data, _ = ncio.read_file(os.path.join(work, "increment_00.nc"))
    assert {"iage", "iage_slow_rest", "po4", "dop", "pop", "dye"} <= set(data)
    inc = trace["krylov_res"][-1]
    tcs = [2, 3, 1]
    names = [["iage", "iage_slow_rest"], ["po4", "dop", "pop"], ["dye"]]
    for i in range(3):
        got = np.stack([data[name] for name in names[i]]).reshape(-1)
        assert got.size == tcs[i] * nz * ny
        assert rel_err(got, inc[i]) < 1.9e-2, i
    ModelState.reset_class()


@pytest.mark.parametrize("names", ["phosphorus", NAMES], ids=["phosphorus", "three_modules"])
def test_krylov_identities_fullsize(tmp_path, names):
    """phosphorus alone (the module of configs[4]) and the three-module mix at 416 x 416 (the grid of configs[2] /
    configs[4]), where no CPU year exists: orthonormal Arnoldi basis and the GMRES residual identity per module -- they
    hold only if the products, the preconditioners, Gram-Schmidt, lin_comb and the checkpointed Hessenbergs of all
    modules are mutually consistent"""
    from nk_ooc_amd.krylov_solver import KrylovSolver, least_squares_coeffs

    n = 416
    work = str(tmp_path)
    cfg, ModelState = _setup(work, n, n, names=names, krylov_max_iter="3", krylov_rel_tol="0.0")
    ModelState.write_files = False
    ntm = len(names.split(","))
    try:
        iterate = ModelState("gen_init_iterate")
        if ntm == 3:
            _structured_dye(ModelState, iterate, n, n)
        hist_fname = os.path.join(work, "hist_00.nc")
        fcn = iterate.comp_fcn(os.path.join(work, "fcn_00.nc"), None, hist_fname)
        solverinfo = dict(cfg["solverinfo"], krylov_workdir=os.path.join(work, "krylov_00"))
        solver = KrylovSolver(iterate, solverinfo, False, False, hist_fname)
        solver.solve(os.path.join(work, "increment_00.nc"), fcn)
        state = solver._solver_state
        assert state.get_iteration() == 3
        beta, hess = state.get_value_saved_state("beta"), state.get_value_saved_state("h_mat")
        assert beta.shape == (ntm, 1) and hess.shape == (ntm, 4, 3, 1)
        basis = [solver._basis(i) for i in range(3)]
        for i in range(3):
            for j in range(i, 3):
                want = 1.0 if i == j else 0.0
                assert np.max(np.abs(basis[i].dot_prod(basis[j])[:, 0] - want)) < 1e-9, (i, j)
        coeff = least_squares_coeffs(beta, hess)
        resid = ModelState.lin_comb_of(coeff, [solver._prod(i) for i in range(3)])
        resid += solver._precond_fcn()
        got = resid.norm()
        for m in range(ntm):
            rhs = np.zeros(4)
            rhs[0] = beta[m, 0]
            lsq_resid = np.linalg.norm(rhs - hess[m, :, :, 0] @ coeff[m, :, 0])
            assert abs(got[m, 0] - lsq_resid) <= 1e-8 * beta[m, 0], m
            assert lsq_resid <= beta[m, 0] * (1.0 + 1e-12)
        assert [tms.eng.frozen_fallbacks() for tms in iterate.tracer_modules] == [0] * ntm
    finally:
        ModelState.write_files = True
        ModelState.reset_class()
