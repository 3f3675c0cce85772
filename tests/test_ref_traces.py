"""G7 of SURVEY.md section 8(c) on the CPU: the oracle's Krylov restatement against a multi-iteration trace of
the reference's OWN KrylovSolver (tests/golden/krylov_trace_26x26.npz, made by tests/golden/gen_ref_traces.py
running nk_ooc.nk_driver in the build container).  The committed baselines of the reference hold Krylov
iteration 0 only; this pins the Hessenberg growth, the least-squares step, both lin_combs and the residual formed
from the un-orthogonalised products at every iteration.  Also: the PRODUCT's least-squares function against the
fixtures made with the reference's `_comp_krylov_basis_coeffs`."""
import numpy as np
import pytest

from helpers import oracle_iage
from oracle import krylov


@pytest.mark.parametrize("n", [26, 30])
def test_oracle_krylov_against_reference_trace(golden_dir, n):
    g = np.load(f"{golden_dir}/krylov_trace_{n}x{n}.npz")
    assert int(g["n"]) == n
    iters = int(g["k0_iterations"])
    assert iters == 3
    model, tm = oracle_iage(n, n)
    weight = np.outer(model.depth.delta, model.ypos.delta)
    mod = krylov.OracleModule(tm, krylov.Regions(np.ones((n, n), dtype=np.int32), weight), precond="reference")
    x, fcn = g["iterate"][0].reshape(-1), g["fcn"][0].reshape(-1)
    # ONE BLAS thread, as the generator ran (OMP_NUM_THREADS=1): SciPy's Radau itself -- the reference's forward year -- is
    # not bit-reproducible across BLAS thread counts (measured: the 30 x 30 perturbed year differs by 3e-11 between 1 and
    # 8 threads; the 26 x 26 one happens not to)
    from threadpoolctl import threadpool_limits

    with threadpool_limits(limits=1):
        inc, trace = krylov.krylov_solve([mod], [x], [fcn], rel_tol=2.0e-4, max_iter=iters)
    assert trace["iterations"] == iters                       # the same stopping decision at every iteration
    # the forward years are bit-identical (the oracle's Radau reproduces solve_ivp), so is everything built
    # from them with the reference's operation order
    assert np.array_equal(trace["beta"], g["k0_beta"])
    assert np.array_equal(trace["precond_fcn"][0], g["k0_precond_fcn"].reshape(-1))
    for j in range(iters):
        assert np.array_equal(trace["perturb_fcn"][j][0], g["k0_perturb_fcn_w_raw"][j].reshape(-1)), j
        assert np.array_equal(trace["w_raw"][j][0], g["k0_w_raw"][j].reshape(-1)), j
        assert np.array_equal(trace["w"][j][0], g["k0_w"][j].reshape(-1)), j
        assert np.array_equal(trace["basis"][j][0], g["k0_basis"][j].reshape(-1)), j
        assert np.array_equal(trace["krylov_res"][j][0], g["k0_krylov_res"][j].reshape(-1)), j
        assert np.array_equal(trace["resid_norm"][j][0], g["k0_precond_resid_norm"][j]), j
    assert np.array_equal(trace["h_mat"][-1], g["k0_h_mat"])
    assert np.array_equal(inc[0], g["increment"][0].reshape(-1))


def test_oracle_krylov_start_against_reference_trace_52(golden_dir):
    """52 x 52 (six Krylov iterations of a 75 s CPU year each in the reference's run): what needs no forward year --
    the preconditioner formula on F, beta, the first basis vector and the preconditioned products of the reference's
    own w_raw files, bit for bit; the least-squares coefficients and both lin_combs from its Hessenberg"""
    from nk_ooc_amd.krylov_solver import least_squares_coeffs

    g = np.load(f"{golden_dir}/krylov_trace_52x52.npz")
    n, iters = int(g["n"]), int(g["k0_iterations"])
    assert (n, iters) == (52, 6)
    model, tm = oracle_iage(n, n)
    weight = np.outer(model.depth.delta, model.ypos.delta)
    mod = krylov.OracleModule(tm, krylov.Regions(np.ones((n, n), dtype=np.int32), weight), precond="reference")
    fcn = g["fcn"][0].reshape(-1)
    precond_fcn = mod.apply_precond(fcn)
    assert np.array_equal(precond_fcn, g["k0_precond_fcn"].reshape(-1))
    beta = krylov.norm([mod], [precond_fcn])
    assert np.array_equal(beta, g["k0_beta"])
    assert np.array_equal(krylov.div([mod], [-precond_fcn], beta)[0], g["k0_basis"][0].reshape(-1))
    for j in range(iters):
        assert np.array_equal(mod.apply_precond(g["k0_w_raw"][j].reshape(-1)), g["k0_w"][j].reshape(-1)), j
    coeff = least_squares_coeffs(g["k0_beta"], g["k0_h_mat"])
    approx = krylov.lin_comb([mod], coeff, [[g["k0_basis"][j].reshape(-1)] for j in range(iters)])[0]
    assert np.array_equal(approx, g["k0_krylov_res"][iters - 1].reshape(-1))
    resid = krylov.lin_comb([mod], coeff, [[g["k0_w"][j].reshape(-1)] for j in range(iters)])[0] + precond_fcn
    assert np.array_equal(krylov.norm([mod], [resid])[0], g["k0_precond_resid_norm"][iters - 1])


def test_product_least_squares_against_reference_fixture(golden_dir):
    """newton-krylov_ooc_amd/krylov_solver.py least_squares_coeffs == nk_ooc/krylov_solver.py:168-181"""
    from nk_ooc_amd.krylov_solver import comp_krylov_basis_coeffs, least_squares_coeffs

    g = np.load(f"{golden_dir}/lstsq.npz")
    for case in range(3):
        got = least_squares_coeffs(g[f"beta{case}"], g[f"h{case}"])
        assert np.array_equal(got, g[f"coeff{case}"]), case
    assert comp_krylov_basis_coeffs is least_squares_coeffs
    # and on the Hessenberg of the reference's own 3-iteration solve
    t = np.load(f"{golden_dir}/krylov_trace_26x26.npz")
    coeff = least_squares_coeffs(t["k0_beta"], t["k0_h_mat"])
    approx = sum(coeff[0, j, 0] * t["k0_basis"][j] for j in range(3))
    assert np.allclose(approx, t["k0_krylov_res"][2], rtol=1e-12, atol=1e-14)
