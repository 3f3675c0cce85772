"""The oracle's grid / coefficient / tendency / Jacobian restatement against golden
vectors generated from the genuine reference functions (tests/golden/gen_golden.py)."""
import numpy as np
import pytest
from scipy import sparse

from helpers import oracle_iage

TAGS = ["26x26", "30x30", "70x40", "20x3_columns"]


@pytest.mark.parametrize("tag", TAGS)
def test_static_fields_bitwise(golden_dir, tag):
    g = np.load(f"{golden_dir}/static_{tag}.npz")
    model, tm = oracle_iage(int(g["nz"]), int(g["ny"]), float(g["max_abs_vvel"]),
                            float(g["horiz_mix_coeff"]))
    for ax in (model.depth, model.ypos):
        for nm in ("edges", "mid", "delta", "delta_r", "delta_mid", "delta_mid_r"):
            assert np.array_equal(getattr(ax, nm), g[f"{ax.name}_{nm}"]), (ax.name, nm)
    for nm in ("stream", "vvel", "wvel", "hmix_coeff"):
        assert np.array_equal(getattr(model, nm), g[nm]), nm
    for i, t in enumerate(g["times"]):
        assert np.array_equal(model.bldepth(t), g["bldepth"][i])
        assert np.array_equal(model.vmix_coeff(t), g["vmix_coeff"][i])
        assert np.array_equal(tm.comp_tend(t, g["y"]), g["tend"][i])
        jac = tm.comp_jacobian(t).tocsr()
        want = sparse.csr_matrix((g[f"jac{i}_data"], g[f"jac{i}_indices"], g[f"jac{i}_indptr"]),
                                 shape=jac.shape)
        diff = jac - want
        assert diff.nnz == 0 or abs(diff).max() == 0.0


@pytest.mark.parametrize("tag", TAGS[:2])
def test_remap_loop_form(golden_dir, tag):
    """the vectorised conservative remap equals the literal loop restatement"""
    g = np.load(f"{golden_dir}/static_{tag}.npz")
    model, _ = oracle_iage(int(g["nz"]), int(g["ny"]))
    for t in g["times"]:
        bld = model.bldepth(t)
        vec = model._remap_ramp(bld)
        loop = np.stack([model.remap_ramp_loop(x) for x in bld], axis=1)
        assert np.array_equal(vec, loop)


def test_tend_is_affine_with_jacobian():
    """comp_tend(y) = J y + s exactly for iage (reference probe in SURVEY 8c)"""
    model, tm = oracle_iage(12, 9)
    rng = np.random.default_rng(0)
    y = rng.standard_normal(2 * 12 * 9)
    t = 0.37 * 365 * 86400.0
    jac = tm.comp_jacobian(t)
    src = tm.comp_tend(t, np.zeros_like(y))
    assert np.max(np.abs(tm.comp_tend(t, y) - (jac @ y + src))) < 1e-15 * np.max(np.abs(jac @ y))


def test_lstsq_known_answers(golden_dir):
    from oracle.krylov import basis_coeffs

    g = np.load(f"{golden_dir}/lstsq.npz")
    for case in range(3):
        assert np.array_equal(basis_coeffs(g[f"beta{case}"], g[f"h{case}"]), g[f"coeff{case}"])


@pytest.mark.parametrize("tag", ["decay_22x9", "restore_const_22x9"])
def test_forced_module_bitwise(golden_dir, tag):
    """`forced` tracer module (decay analogue of BASELINE.json's dye_decay), reference
    py_driver_2d/forced.py:114-190"""
    from oracle import radau
    from oracle.grid import default_axes
    from oracle.model import Forced, Py2dModel

    g = np.load(f"{golden_dir}/forced_{tag}.npz")
    depth, ypos = default_axes(int(g["nz"]), int(g["ny"]))
    tm = Forced(Py2dModel(depth, ypos), str(g["surf_restore_opt"]), float(g["surf_restore_const"]),
                str(g["sms_opt"]), float(g["sms_decay_rate"]), float(g["sms_const"]))
    for i, t in enumerate(g["times"]):
        assert np.array_equal(tm.comp_tend(t, g["y"]), g["tend"][i])
        jac = tm.comp_jacobian(t).tocsr()
        want = sparse.csr_matrix((g[f"jac{i}_data"], g[f"jac{i}_indices"], g[f"jac{i}_indptr"]), shape=jac.shape)
        diff = jac - want
        assert diff.nnz == 0 or abs(diff).max() == 0.0
    if "fcn" in g:
        res, solver = radau.comp_fcn(tm, g["y0"], return_solver=True)
        assert np.array_equal(res, g["fcn"])
        assert (solver.stats.nfev, solver.stats.njev, solver.stats.nlu) == (int(g["nfev"]), int(g["njev"]), int(g["nlu"]))


def test_remap_two_knots_reference_cases():
    """the cases the reference pins for SpatialAxis.remap_linear_interpolant with two knots
    (tests/test_spatial_axis.py:156-185): 5 uniform layers on [0, 50]"""
    from oracle.model import remap_two_knots

    edges = np.linspace(0.0, 50.0, 6)
    delta_r = 1.0 / np.diff(edges)
    cases = [([-15.0, -5.0], [1.0, 2.0], [2.0, 2.0, 2.0, 2.0, 2.0]),
             ([-15.0, 25.0], [0.0, 8.0], [4.0, 6.0, 7.75, 8.0, 8.0]),
             ([5.0, 25.0], [0.0, 8.0], [0.5, 4.0, 7.5, 8.0, 8.0]),
             ([22.5, 27.5], [0.0, 8.0], [0.0, 0.0, 4.0, 8.0, 8.0])]
    for xv, yv, expected in cases:
        assert np.array_equal(remap_two_knots(edges, delta_r, xv, yv), np.array(expected))
