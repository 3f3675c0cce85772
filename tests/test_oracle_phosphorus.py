"""Oracle of the phosphorus tracer module against the reference's goldens, and the
conditioning of the reference's eigenvalue shift (CPU)."""
import numpy as np
import pytest
from scipy import sparse

from oracle import radau
from oracle.grid import default_axes
from oracle.krylov import Regions
from oracle.model import (Phosphorus, Py2dModel, apply_precond_phosphorus, phosphorus_precond_matrix,
                          phosphorus_small_eigs)


def _module(nz, ny):
    depth, ypos = default_axes(nz, ny)
    return Phosphorus(Py2dModel(depth, ypos)), depth, ypos


@pytest.mark.parametrize("tag", ["22x9", "70x12"])
def test_tend_and_jacobian_bitwise(golden_dir, tag):
    g = np.load(f"{golden_dir}/phosphorus_{tag}.npz")
    tm, _, _ = _module(int(g["nz"]), int(g["ny"]))
    assert np.array_equal(tm.light_lim, g["light_lim"])
    for i, t in enumerate(g["times"]):
        assert np.array_equal(tm.comp_tend(t, g["y"]), g["tend"][i])
        jac = tm.comp_jacobian(t, g["y"]).tocsr()
        ref = sparse.csr_matrix((g[f"jac{i}_data"], g[f"jac{i}_indices"], g[f"jac{i}_indptr"]), shape=jac.shape)
        diff = jac - ref
        assert diff.nnz == 0 or abs(diff).max() == 0.0


def test_forward_year_bitwise(golden_dir):
    g = np.load(f"{golden_dir}/phosphorus_22x9.npz")
    tm, _, _ = _module(22, 9)
    res, solver = radau.comp_fcn(tm, g["y0"], return_solver=True)
    assert np.array_equal(res, g["fcn"])
    assert (solver.stats.nfev, solver.stats.njev, solver.stats.nlu) == (int(g["nfev"]), int(g["njev"]), int(g["nlu"]))


def test_reference_shift_is_start_vector_dependent(golden_dir):
    """`eigs(mat, sigma=0.0)` (phosphorus.py:239) inverts the singular mat: its second
    eigenvalue changes with ARPACK's start vector (even in sign), while shift-invert about a
    small positive sigma reproduces the dense eigenvalues; the preconditioner inherits the
    scatter through `shift`."""
    g = np.load(f"{golden_dir}/phosphorus_22x9.npz")
    tm, depth, ypos = _module(22, 9)
    po4 = g["y"].reshape(3, 22, 9)[0]
    mat = phosphorus_precond_matrix(tm, po4)
    dense = np.linalg.eigvals(mat.toarray())
    dense = dense[np.argsort(np.abs(dense))]
    assert abs(dense[0]) < 1e-10
    good, _ = phosphorus_small_eigs(mat, 0.02)
    assert abs(good[1].real - dense[1].real) < 1e-9 * abs(dense[1])
    n = mat.shape[0]
    reals = []
    for seed in range(4):
        vals, _ = phosphorus_small_eigs(mat, 0.0, v0=np.random.default_rng(seed).standard_normal(n))
        reals.append(vals[1].real)
    scatter = (max(reals) - min(reals)) / abs(dense[1].real)
    assert scatter > 1e-4          # not a property of the matrix
    regions = Regions(np.ones((22, 9), dtype=np.int32), np.outer(depth.delta, ypos.delta))
    v = np.random.default_rng(3).standard_normal(n)
    exact, _, shift = apply_precond_phosphorus(tm, regions, po4, v)
    moved, _, _ = apply_precond_phosphorus(tm, regions, po4, v, shift=0.5 * min(reals))
    rel = np.max(np.abs(moved - exact)) / np.max(np.abs(exact))
    assert rel > 1e-6
    # the extrapolated solve conserves total P: zero weighted mean over the three tracers
    sol = (exact + v).reshape(3, -1)
    assert abs(sum(regions.mean_of(p) for p in sol)[0]) < 1e-9 * np.max(np.abs(sol))
