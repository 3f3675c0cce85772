"""GPU parity of the deterministic kernels against the oracle (C ABI -> HIP)."""
import numpy as np
import pytest
from scipy import sparse
from scipy.sparse import linalg as spl

from helpers import oracle_iage, rel_err

pytestmark = pytest.mark.gpu

YEAR = 365.0 * 86400.0
GRIDS = [(26, 26), (70, 40), (130, 37), (20, 3)]


def make_engine(nz, ny, vv=0.1, kh=1000.0, **kw):
    from nk_ooc_amd.engine import iage_engine
    from nk_ooc_amd.grid import Grid2d

    return iage_engine(Grid2d.default(nz, ny, vv, kh), **kw)


@pytest.mark.parametrize("nz,ny", GRIDS)
def test_roundtrip(nz, ny):
    eng = make_engine(nz, ny)
    rng = np.random.default_rng(0)
    y = rng.standard_normal((2, nz, ny))
    assert np.array_equal(eng.download(eng.upload(y)), y)


@pytest.mark.parametrize("nz,ny", GRIDS)
def test_vmix_coeff(nz, ny):
    eng = make_engine(nz, ny)
    model, _ = oracle_iage(nz, ny)
    for t in [0.0, 0.2613 * YEAR, 0.3 * YEAR, 0.5 * YEAR, 0.7 * YEAR, 0.99 * YEAR]:
        got = eng.vmix_coeff(t)
        want = model.vmix_coeff(t)
        # same operation order; only exp() may differ by an ulp
        assert np.max(np.abs(got - want) / np.abs(want)) < 1e-14, t


@pytest.mark.parametrize("nz,ny", GRIDS)
def test_tend(nz, ny):
    eng = make_engine(nz, ny)
    _, tm = oracle_iage(nz, ny)
    rng = np.random.default_rng(1)
    y = rng.standard_normal((2, nz, ny))
    yd = eng.upload(y)
    for t in [0.0, 0.3 * YEAR, 0.61 * YEAR]:
        got = eng.download(eng.tend(t, yd)).reshape(-1)
        want = tm.comp_tend(t, y.reshape(-1))
        assert rel_err(got, want) < 1e-13, t


def test_tend_golden(golden_dir):
    g = np.load(f"{golden_dir}/static_26x26.npz")
    eng = make_engine(26, 26)
    yd = eng.upload(g["y"])
    for i, t in enumerate(g["times"]):
        got = eng.download(eng.tend(t, yd)).reshape(-1)
        assert rel_err(got, g["tend"][i]) < 1e-13


@pytest.mark.parametrize("nz,ny,vv,kh", [(26, 26, 0.1, 1000.0), (70, 40, 0.1, 1000.0), (20, 3, 0.0, 0.0)])
def test_jacobian_diags(nz, ny, vv, kh):
    eng = make_engine(nz, ny, vv, kh)
    model, tm = oracle_iage(nz, ny, vv, kh)
    for t in [0.0, 0.4 * YEAR]:
        got = eng.jacobian_diags(t)
        up, south, center, north, dn = model.jac_diags(t)
        for tr in range(2):
            want = [up, south, center + tm.diag_extra(tr), north, dn]
            for d in range(5):
                scale = np.max(np.abs(want[d])) or 1.0
                assert np.max(np.abs(got[d, tr] - want[d])) / scale < 1e-14, (t, tr, d)


@pytest.mark.parametrize("nz,ny", [(26, 26), (70, 40), (130, 37)])
@pytest.mark.parametrize("hfrac", [1e-7, 2e-4, 0.01])
def test_shifted_solve(nz, ny, hfrac):
    from oracle import radau

    eng = make_engine(nz, ny, lin_tol=1.0e-13)  # the solver itself, to full accuracy
    _, tm = oracle_iage(nz, ny)
    t_jac = 0.3 * YEAR
    J = tm.comp_jacobian(t_jac).tocsc()
    n = J.shape[0]
    h = hfrac * YEAR
    rng = np.random.default_rng(2)
    b = rng.standard_normal(n)
    bi = rng.standard_normal(n)
    # real system
    want = spl.splu((radau.MU_REAL / h * sparse.identity(n, format="csc") - J).tocsc()).solve(b)
    x_re, _, sweeps = eng.shifted_solve(t_jac, h, radau.MU_REAL, eng.upload(b))
    assert rel_err(eng.download(x_re).reshape(-1), want) < 1e-11, sweeps
    # complex system
    A = (radau.MU_COMPLEX / h * sparse.identity(n, format="csc") - J).tocsc()
    wantc = spl.splu(A).solve(b + 1j * bi)
    x_re, x_im, sweeps = eng.shifted_solve(t_jac, h, radau.MU_COMPLEX, eng.upload(b), eng.upload(bi))
    got = eng.download(x_re).reshape(-1) + 1j * eng.download(x_im).reshape(-1)
    assert rel_err(got, wantc) < 1e-11, sweeps


@pytest.mark.parametrize("nz", [64, 65, 150, 200, 257, 321, 390, 512])
def test_every_levels_per_lane_instantiation(nz):
    """one grid per template instantiation E = ceil(nz / 64) = 1 .. 8 (full and ragged last lanes):
    tendency, shifted solves and a step-replayed year segment against the oracle"""
    from oracle import radau

    ny = 5
    eng = make_engine(nz, ny, lin_tol=1.0e-12)
    _, tm = oracle_iage(nz, ny)
    rng = np.random.default_rng(nz)
    y = 1.0 + rng.random(2 * nz * ny)
    t = 0.3 * YEAR
    assert rel_err(eng.download(eng.tend(t, eng.upload(y))).reshape(-1), tm.comp_tend(t, y)) < 1e-13
    J = tm.comp_jacobian(t).tocsc()
    n = J.shape[0]
    b = rng.standard_normal(n)
    bi = rng.standard_normal(n)
    for h in (1.0e3, 3.0e5):
        want = spl.splu((radau.MU_REAL / h * sparse.identity(n, format="csc") - J).tocsc()).solve(b)
        x_re, _, _ = eng.shifted_solve(t, h, radau.MU_REAL, eng.upload(b))
        assert rel_err(eng.download(x_re).reshape(-1), want) < 1e-10, h
        wantc = spl.splu((radau.MU_COMPLEX / h * sparse.identity(n, format="csc") - J).tocsc()).solve(b + 1j * bi)
        x_re, x_im, _ = eng.shifted_solve(t, h, radau.MU_COMPLEX, eng.upload(b), eng.upload(bi))
        got = eng.download(x_re).reshape(-1) + 1j * eng.download(x_im).reshape(-1)
        assert rel_err(got, wantc) < 1e-10, h
    # the integrator kernels of this instantiation: a short span of the year in step-replay mode
    from nk_ooc_amd.engine import iage_engine
    from nk_ooc_amd.grid import Grid2d

    span = (0.0, 0.02 * YEAR)
    eng2 = iage_engine(Grid2d.default(nz, ny), time_range=span)
    want, solver = radau.comp_fcn(tm, y, time_range=span, return_solver=True)
    fx, _, _ = eng2.comp_fcn(eng2.upload(y), replay=np.array(solver.schedule))
    assert rel_err(eng2.download(fx).reshape(-1), want) < 1e-9
