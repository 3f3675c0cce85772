"""Preconditioner oracle: the reference formula is pinned bit for bit; its roundoff
sensitivity is measured; the backward-stable form of the same operator (which the HIP
path implements) agrees with it to within that sensitivity."""
import numpy as np
import pytest
from scipy.sparse import linalg as spl

from helpers import oracle_iage
from oracle.model import apply_precond_stable

CASES = [("26x26", 26, 26, 0.1, 1000.0), ("20x3_columns", 20, 3, 0.0, 0.0)]


@pytest.mark.parametrize("tag,nz,ny,vv,kh", CASES)
def test_reference_formula_bitwise(golden_dir, tag, nz, ny, vv, kh):
    g = np.load(f"{golden_dir}/precond_{tag}.npz")
    _, tm = oracle_iage(nz, ny, vv, kh)
    assert np.array_equal(tm.apply_precond(g["v"]), g["res"])


@pytest.mark.parametrize("tag,nz,ny,vv,kh", CASES)
def test_stable_form_within_reference_noise(golden_dir, tag, nz, ny, vv, kh):
    g = np.load(f"{golden_dir}/precond_{tag}.npz")
    _, tm = oracle_iage(nz, ny, vv, kh)
    v = g["v"]
    ref = g["res"]
    stable = apply_precond_stable(tm, v)
    # reference's own sensitivity: relative 1e-16 noise on the entries of the explicit
    # product I - A0 A1 A2 (two ulp-level perturbations)
    mat = tm.precond_matrix().tocsc()
    rng = np.random.default_rng(0)
    P = nz * ny
    for tr in range(2):
        s = slice(tr * P, (tr + 1) * P)
        noise = 0.0
        for _ in range(2):
            pert = mat.copy()
            pert.data = pert.data * (1.0 + 1e-16 * rng.standard_normal(pert.nnz))
            noisy = spl.spsolve(pert, v) - v
            noise = max(noise, np.linalg.norm(noisy[s] - ref[s]) / np.linalg.norm(ref[s]))
        err = np.linalg.norm(stable[s] - ref[s]) / np.linalg.norm(ref[s])
        assert err < 10 * noise + 1e-9, (tr, err, noise)
        # and in absolute terms the two agree to the reference CI tolerance for precond_fcn
        assert err < 2.0e-3, (tr, err)
