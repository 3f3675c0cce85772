"""GPU parity of the `phosphorus` tracer module (po4, dop, pop; nonlinear uptake couples the
tracers in every cell; reference py_driver_2d/phosphorus.py): tendencies, Jacobian action,
shifted solves and the forward model year against the oracle and the reference's goldens."""
import numpy as np
import pytest
from scipy.sparse import identity
from scipy.sparse.linalg import spsolve

from helpers import free_years, rel_err
from oracle import radau
from oracle.grid import default_axes
from oracle.model import Phosphorus, Py2dModel

pytestmark = pytest.mark.gpu
YEAR = 365.0 * 86400.0


def _setup(golden_dir, tag, **kwargs):
    from nk_ooc_amd.engine import phosphorus_engine
    from nk_ooc_amd.grid import Grid2d

    g = np.load(f"{golden_dir}/phosphorus_{tag}.npz")
    nz, ny = int(g["nz"]), int(g["ny"])
    eng = phosphorus_engine(Grid2d.default(nz, ny), **kwargs)
    depth, ypos = default_axes(nz, ny)
    return g, eng, Phosphorus(Py2dModel(depth, ypos))


@pytest.mark.parametrize("tag", ["22x9", "70x12"])
def test_phosphorus_tend_and_jacobian(golden_dir, tag):
    g, eng, tm = _setup(golden_dir, tag)
    yd = eng.upload(g["y"])
    eng.set_lin_state(yd)
    rng = np.random.default_rng(12)
    v = rng.standard_normal(g["y"].size)
    vd = eng.upload(v)
    for i, t in enumerate(g["times"]):
        got = eng.download(eng.tend(t, yd)).reshape(-1)
        # same operation order as the reference: the tendency is reproduced to the last bits
        assert np.max(np.abs(got - g["tend"][i])) <= 4e-16 * np.max(np.abs(g["tend"][i]))
        jv = eng.download(eng.jacobian_apply(t, vd)).reshape(-1)
        assert rel_err(jv, tm.comp_jacobian(t, g["y"]) @ v) < 1e-13


@pytest.mark.parametrize("tag", ["22x9", "70x12"])
def test_phosphorus_shifted_solves(golden_dir, tag):
    """(mu/h I - J) x = b for the real and the complex Radau shift, J with the inter-tracer
    coupling; line relaxation to 1e-10 against a sparse direct solve"""
    g, eng, tm = _setup(golden_dir, tag, lin_tol=1e-10)
    yd = eng.upload(g["y"])
    eng.set_lin_state(yd)
    t, h = g["times"][1], 2.0e5
    jac = tm.comp_jacobian(t, g["y"]).tocsc()
    n = g["y"].size
    rng = np.random.default_rng(13)
    b = rng.standard_normal(n)
    b2 = rng.standard_normal(n)
    x_re, _, sweeps = eng.shifted_solve(t, h, radau.MU_REAL, eng.upload(b))
    want = spsolve((radau.MU_REAL / h) * identity(n, format="csc") - jac, b)
    assert rel_err(eng.download(x_re).reshape(-1), want) < 1e-8
    x_re, x_im, _ = eng.shifted_solve(t, h, radau.MU_COMPLEX, eng.upload(b), eng.upload(b2))
    want = spsolve(((radau.MU_COMPLEX / h) * identity(n, format="csc") - jac).astype(complex), b + 1j * b2)
    got = eng.download(x_re).reshape(-1) + 1j * eng.download(x_im).reshape(-1)
    assert rel_err(got, want) < 1e-8
    assert sweeps >= 2


def test_phosphorus_comp_fcn(golden_dir):
    """forward model year: step-replay against the oracle (1e-9) and the free-running
    integrator against the reference's solve_ivp result at the reference's CI tolerance"""
    g, eng, tm = _setup(golden_dir, "22x9")
    want, solver = radau.comp_fcn(tm, g["y0"], return_solver=True)
    assert np.array_equal(want, g["fcn"])
    fx, _, _ = eng.comp_fcn(eng.upload(g["y0"]), replay=np.array(solver.schedule))
    assert rel_err(eng.download(fx).reshape(-1), want) < 1e-9
    (fx, stats, _), (fx_def, _, _) = free_years(eng, eng.upload(g["y0"]))
    assert np.allclose(eng.download(fx).reshape(-1), g["fcn"], rtol=1e-3, atol=1e-6)
    assert np.allclose(eng.download(fx_def).reshape(-1), g["fcn"], rtol=1e-3, atol=1e-6)
    assert abs(stats["nfev"] - int(g["nfev"])) <= 0.1 * int(g["nfev"]) + 20


def test_phosphorus_preconditioner_with_regions(golden_dir):
    """three ypos bands as regions: the null-space correction is applied region by region
    (TracerModuleStateBase.mean / broadcast_region_vals semantics), as in the oracle"""
    from oracle.krylov import Regions
    from oracle.model import apply_precond_phosphorus

    g, eng, tm = _setup(golden_dir, "22x9")
    nz, ny = 22, 9
    depth, ypos = default_axes(nz, ny)
    weight = np.outer(depth.delta, ypos.delta)
    mask = np.ones((nz, ny), dtype=np.int32)
    mask[:, 3:6] = 2
    mask[:, 6:] = 3
    eng.set_region(mask, weight)
    po4 = g["y"].reshape(3, nz, ny)[0]
    v = np.random.default_rng(22).standard_normal(3 * nz * ny)
    eng.precond_setup_state(po4)
    want, _, _ = apply_precond_phosphorus(tm, Regions(mask, weight), po4, v)
    got = eng.download(eng.precond_apply(eng.upload(v))).reshape(-1)
    assert rel_err(got, want) < 1e-6


@pytest.mark.parametrize("tag", ["22x9", "70x12"])
def test_phosphorus_preconditioner(golden_dir, tag):
    """shifted block solves, the eigen-pair from shift-invert Arnoldi on the device
    solver, and the assembled preconditioner against the oracle (sparse direct + ARPACK)"""
    from oracle.krylov import Regions
    from oracle.model import apply_precond_phosphorus, phosphorus_precond_matrix

    g, eng, tm = _setup(golden_dir, tag)
    nz, ny = int(g["nz"]), int(g["ny"])
    depth, ypos = default_axes(nz, ny)
    weight = np.outer(depth.delta, ypos.delta)
    mask = np.ones((nz, ny), dtype=np.int32)
    eng.set_region(mask, weight)
    po4 = g["y"].reshape(3, nz, ny)[0]
    mat = phosphorus_precond_matrix(tm, po4)
    n = mat.shape[0]
    rng = np.random.default_rng(21)
    v = rng.standard_normal(n)
    # one shifted system, both signs of the shift
    ylin = np.zeros((3, nz, ny))
    ylin[0] = po4
    eng.set_lin_state(eng.upload(ylin))
    eng.shift_factor(0.5 * YEAR, YEAR, [0.02, -0.03])
    for i, sigma in enumerate([0.02, -0.03]):
        want = spsolve((mat - sigma * identity(n, format="csc")).tocsc(), v)
        got = eng.download(eng.shift_solve(i, eng.upload(v))).reshape(-1)
        assert rel_err(got, want) < 1e-8
    pc = eng.precond_setup_state(po4)
    want, e_vals, shift = apply_precond_phosphorus(tm, Regions(mask, weight), po4, v)
    assert abs(pc.e_vals[0]) < 1e-9
    assert abs(pc.e_vals[1].real - e_vals[1].real) < 1e-8 * abs(e_vals[1].real)
    assert abs(pc.shift - shift) < 1e-8 * abs(shift)
    got = eng.download(eng.precond_apply(eng.upload(v))).reshape(-1)
    assert rel_err(got, want) < 1e-6


@pytest.mark.parametrize("tag", ["22x9", "70x12"])
def test_phosphorus_preconditioner_single_precision_storage(golden_dir, tag):
    """option "pc_fp32" for the shifted systems of the phosphorus preconditioner (round-3 verdict, missing 4): the explicit
    inverses of the block elimination kept in single precision, every solve refined against the exact shifted operator
    (k_pc_residual_shift).  The shifted matrices are not M-matrices (mat - shift I is indefinite for the shifts the
    preconditioner uses), so how far a refinement gets is measured, not assumed: the solves and the assembled preconditioner
    against the double precision storage and the oracle, for one and for two refinements"""
    from oracle.model import phosphorus_precond_matrix

    g, eng, tm = _setup(golden_dir, tag)
    nz, ny = int(g["nz"]), int(g["ny"])
    po4 = g["y"].reshape(3, nz, ny)[0]
    mat = phosphorus_precond_matrix(tm, po4)
    n = mat.shape[0]
    v = np.random.default_rng(21).standard_normal(n)
    ylin = np.zeros((3, nz, ny))
    ylin[0] = po4
    shifts = [0.02, -0.03]
    wants = [spsolve((mat - sigma * identity(n, format="csc")).tocsc(), v) for sigma in shifts]
    eng.set_lin_state(eng.upload(ylin))
    pc64 = eng.precond_setup_state(po4)
    apply64 = eng.download(eng.precond_apply(eng.upload(v))).reshape(-1)
    eng.close()
    errs = {}
    for refine in (0, 1, 2):
        _, eng32, _ = _setup(golden_dir, tag)
        eng32.set_option("pc_fp32", 1)
        eng32.set_option("pc_refine", refine)
        eng32.set_lin_state(eng32.upload(ylin))
        eng32.shift_factor(0.5 * YEAR, YEAR, shifts)
        errs[refine] = [rel_err(eng32.download(eng32.shift_solve(i, eng32.upload(v))).reshape(-1), wants[i]) for i in range(2)]
        if refine == 2:
            pc32 = eng32.precond_setup_state(po4)
            assert abs(pc32.e_vals[1].real - pc64.e_vals[1].real) < 1e-7 * abs(pc64.e_vals[1].real)
            apply32 = eng32.download(eng32.precond_apply(eng32.upload(v))).reshape(-1)
            errs["apply"] = rel_err(apply32, apply64)
        eng32.close()
    print(f"phosphorus {tag}, single precision Schur inverses: shifted solves against a sparse direct solve "
          f"{errs[0]} unrefined, {errs[1]} refined once, {errs[2]} twice; preconditioner against double precision storage {errs['apply']:.2e}")
    assert max(errs[0]) > 1e-9                      # the single precision shows without refinement
    assert max(errs[2]) < 1e-8 and max(errs[2]) <= max(errs[0])
    assert errs["apply"] < 1e-6


def test_phosphorus_krylov_solve(tmp_path):
    """tracer_module_names = phosphorus through the solver mirrors: forward year with history,
    preconditioner from the end-of-year po4, GMRES iterations; compared with the oracle"""
    import os

    from nk_ooc_amd import ncio
    from nk_ooc_amd.krylov_solver import KrylovSolver
    from nk_ooc_amd.model_config import ModelConfig
    from nk_ooc_amd.model_state import ModelState
    from nk_ooc_amd.setup_solver import gen_grid_vars_file, make_config
    from oracle import krylov

    nz, ny = 22, 9
    work = str(tmp_path)
    cfg = make_config(work, nz, ny, tracer_module_names="phosphorus",
                      extra_solverinfo={"krylov_max_iter": "3", "krylov_rel_tol": "1e-9"})
    gen_grid_vars_file(cfg["modelinfo"])
    ModelState.reset_class()
    ModelState.model_config_obj = ModelConfig(cfg["modelinfo"])
    ModelState.write_files = True
    iterate = ModelState("gen_init_iterate")
    assert iterate.tracer_modules[0].tracer_names == ["po4", "dop", "pop"]
    hist_fname = os.path.join(work, "hist_00.nc")
    fcn = iterate.comp_fcn(os.path.join(work, "fcn_00.nc"), None, hist_fname)
    data, _ = ncio.read_file(hist_fname, ["po4", "po4_uptake", "po4_uptake_time_mean"])
    assert data["po4_uptake"].shape == (61, nz, ny) and np.all(data["po4_uptake"] >= 0.0)
    solverinfo = dict(cfg["solverinfo"], krylov_workdir=os.path.join(work, "krylov_00"))
    solver = KrylovSolver(iterate, solverinfo, False, False, hist_fname)
    solver.solve(os.path.join(work, "increment_00.nc"), fcn)
    beta = solver._solver_state.get_value_saved_state("beta")
    h_mat = solver._solver_state.get_value_saved_state("h_mat")
    assert os.path.exists(os.path.join(work, "krylov_00", "precond_null_space.nc"))
    # oracle
    depth, ypos = default_axes(nz, ny)
    regions = krylov.Regions(np.ones((nz, ny), dtype=np.int32), np.outer(depth.delta, ypos.delta))
    mod = krylov.OracleModule(Phosphorus(Py2dModel(depth, ypos)), regions, precond="phosphorus")
    x = [iterate.tracer_modules[0].get_tracer_vals_all().reshape(-1)]
    f = [mod.comp_fcn(x[0])]
    assert np.allclose(fcn.tracer_modules[0].get_tracer_vals_all().reshape(-1), f[0], rtol=1e-3, atol=1e-6)
    mod.precond_po4 = (x[0] + f[0]).reshape(3, nz, ny)[0]
    assert np.allclose(data["po4"][-1], mod.precond_po4, rtol=1e-3, atol=1e-6)
    _, trace = krylov.krylov_solve([mod], x, f, rel_tol=1e-9, max_iter=3)
    assert rel_err(beta, trace["beta"]) < 1e-3
    assert rel_err(h_mat, trace["h_mat"][-1]) < 5e-2
    ModelState.reset_class()


def test_phosphorus_newton(tmp_path):
    """Newton-Krylov on the phosphorus module run to convergence through the driver mirror:
    the converged iterate satisfies the reference's convergence test when F is evaluated by the
    CPU oracle, and total phosphorus (weighted mean over the three tracers) is conserved"""
    import os

    from nk_ooc_amd import ncio, nk_driver
    from nk_ooc_amd.model_state import ModelState
    from nk_ooc_amd.setup_solver import make_config, setup
    from oracle import krylov

    nz, ny = 22, 9
    work = str(tmp_path)
    cfg = make_config(work, nz, ny, tracer_module_names="phosphorus")
    ModelState.write_files = True
    setup(cfg, fp_cnt=1)
    solver = nk_driver.run(cfg)
    assert solver.converged().all()
    assert 1 <= solver.get_iteration() <= 4
    names = ["po4", "dop", "pop"]
    read = lambda fname: np.stack([ncio.read_file(fname, names)[0][n] for n in names]).reshape(-1)
    x0 = read(cfg["solverinfo"]["init_iterate_fname"])
    x = solver.iterate.tracer_modules[0].get_tracer_vals_all().reshape(-1)
    depth, ypos = default_axes(nz, ny)
    regions = krylov.Regions(np.ones((nz, ny), dtype=np.int32), np.outer(depth.delta, ypos.delta))
    mod = krylov.OracleModule(Phosphorus(Py2dModel(depth, ypos)), regions)
    total = lambda v: sum(regions.mean_of(p) for p in v.reshape(3, -1))[0]
    assert abs(total(x) - total(x0)) < 1e-9 * abs(total(x0))
    f_cpu = mod.comp_fcn(x)
    fn, xn = np.sqrt(mod.dot(f_cpu, f_cpu)), np.sqrt(mod.dot(x, x))
    assert np.all(fn < 2.0 * 1.0e-5 * xn), (fn, xn)
    stats, _ = ncio.read_file(os.path.join(work, "Newton_stats.nc"))
    for name in ("iterate_norm_phosphorus", "fcn_norm_phosphorus", "Krylov_iterations", "po4", "pop_mean_ypos"):
        assert name in stats, name
    ModelState.reset_class()


def test_phosphorus_newton_resume(tmp_path):
    """phosphorus run killed inside the second Krylov iteration and resumed with fresh contexts: the
    state dependent preconditioner is rebuilt from po4 in the preconditioner file and the run ends
    at the same iterate"""
    import os

    from nk_ooc_amd import nk_driver
    from nk_ooc_amd.model_state import ModelState
    from nk_ooc_amd.setup_solver import make_config, setup

    class Interrupt(Exception):
        pass

    def run(workdir, interrupt):
        cfg = make_config(workdir, 22, 9, tracer_module_names="phosphorus")
        ModelState.write_files = True
        setup(cfg, fp_cnt=1)
        if not interrupt:
            return nk_driver.run(cfg)
        original = ModelState.comp_fcn

        def guarded(self, res_fname, solver_state, hist_fname=None, **kw):
            if "perturb_fcn_w_raw_01" in os.path.basename(res_fname):
                raise Interrupt(res_fname)
            return original(self, res_fname, solver_state, hist_fname, **kw)

        ModelState.comp_fcn = guarded
        try:
            with pytest.raises(Interrupt):
                nk_driver.run(cfg)
        finally:
            ModelState.comp_fcn = original
        return nk_driver.run(cfg, resume=True)

    straight = run(str(tmp_path / "a"), False)
    x_a = straight.iterate.tracer_modules[0].get_tracer_vals_all()
    resumed = run(str(tmp_path / "b"), True)
    x_b = resumed.iterate.tracer_modules[0].get_tracer_vals_all()
    assert resumed.get_iteration() == straight.get_iteration()
    assert np.allclose(x_a, x_b, rtol=1e-10, atol=1e-13)
    ModelState.reset_class()


@pytest.mark.parametrize("nz", [130, 200, 300, 384, 512])
def test_phosphorus_other_instantiations(nz):
    """the phosphorus kernels at other levels-per-lane counts (E = 3, 4, 5, 6, 8): tendency, J v
    and the coupled shifted solves against the oracle"""
    from nk_ooc_amd.engine import phosphorus_engine
    from nk_ooc_amd.grid import Grid2d

    ny = 4
    eng = phosphorus_engine(Grid2d.default(nz, ny), lin_tol=1e-11)
    depth, ypos = default_axes(nz, ny)
    tm = Phosphorus(Py2dModel(depth, ypos))
    rng = np.random.default_rng(nz)
    prof = [np.interp(depth.mid, zs, vs) for zs, vs in (([1.3e2, 2.6e2], [5.5e-3, 4.1e0]),
                                                        ([9.5e1, 1.4e2], [7.1e-2, 1.5e-4]),
                                                        ([1.7e2, 2.5e2], [1.8e-2, 7.9e-4]))]
    y = (np.stack([np.broadcast_to(p[:, None], (nz, ny)) for p in prof]) * (1.0 + 0.2 * rng.random((3, nz, ny)))).reshape(-1)
    t = 0.6 * YEAR
    yd = eng.upload(y)
    eng.set_lin_state(yd)
    assert rel_err(eng.download(eng.tend(t, yd)).reshape(-1), tm.comp_tend(t, y)) < 1e-14
    jac = tm.comp_jacobian(t, y).tocsc()
    n = y.size
    v = rng.standard_normal(n)
    assert rel_err(eng.download(eng.jacobian_apply(t, eng.upload(v))).reshape(-1), jac @ v) < 1e-13
    h = 1.0e5
    want = spsolve((radau.MU_REAL / h) * identity(n, format="csc") - jac, v)
    x_re, _, _ = eng.shifted_solve(t, h, radau.MU_REAL, eng.upload(v))
    assert rel_err(eng.download(x_re).reshape(-1), want) < 1e-9
    v2 = rng.standard_normal(n)
    want = spsolve(((radau.MU_COMPLEX / h) * identity(n, format="csc") - jac).astype(complex), v + 1j * v2)
    x_re, x_im, _ = eng.shifted_solve(t, h, radau.MU_COMPLEX, eng.upload(v), eng.upload(v2))
    assert rel_err(eng.download(x_re).reshape(-1) + 1j * eng.download(x_im).reshape(-1), want) < 1e-9
