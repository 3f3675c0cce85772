"""GPU parity of the `phosphorus` tracer module (po4, dop, pop; nonlinear uptake couples the
tracers in every cell; reference py_driver_2d/phosphorus.py): tendencies, Jacobian action,
shifted solves and the forward model year against the oracle and the reference's goldens."""
import numpy as np
import pytest
from scipy.sparse import identity
from scipy.sparse.linalg import spsolve

from helpers import rel_err
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
    fx, stats, _ = eng.comp_fcn(eng.upload(g["y0"]))
    assert np.allclose(eng.download(fx).reshape(-1), g["fcn"], rtol=1e-3, atol=1e-6)
    assert abs(stats["nfev"] - int(g["nfev"])) <= 0.1 * int(g["nfev"]) + 20


@pytest.mark.parametrize("tag", ["22x9", "70x12"])
def test_phosphorus_preconditioner(golden_dir, tag):
    """shifted block solves, the eigen-pair from subspace inverse iteration on the device
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
