"""Oracle of the file-driven options of the `forced` tracer module (reference
py_driver_2d/forced.py:42-56,125-153,188-241; utils.py:488-533) against fixtures made with the
reference's own comp_tend / comp_jacobian (tests/golden/gen_golden.py gen_forced_file), and the
host's forcing reader against scipy's interp1d."""
import numpy as np
import pytest
from scipy import interpolate, sparse

from helpers import rel_err

TAGS = ["file_restore_sms_22x9", "file_sink_thres_22x9", "file_restore_decay_70x5"]


def oracle_forced(g):
    from oracle.grid import default_axes
    from oracle.model import Forced, Py2dModel

    depth, ypos = default_axes(int(g["nz"]), int(g["ny"]))
    thres = float(g["sink_thres"])
    return Forced(Py2dModel(depth, ypos), str(g["surf_restore_opt"]), float(g["surf_restore_const"]),
                  str(g["sms_opt"]), float(g["sms_decay_rate"]), 0.0,
                  surf_restore_series=(g["rec_times"], g["restore_vals"]),
                  sms_series=(g["rec_times"], g["sms_vals"]), sink_thres=thres if thres > 0.0 else None)


@pytest.mark.parametrize("tag", TAGS)
def test_forced_file_module_bitwise(golden_dir, tag):
    from oracle import radau

    g = np.load(f"{golden_dir}/forced_{tag}.npz")
    tm = oracle_forced(g)
    for i, t in enumerate(g["times"]):
        assert np.array_equal(tm.comp_tend(t, g["y"]), g["tend"][i])
        jac = tm.comp_jacobian(t, g["y"]).tocsr()
        want = sparse.csr_matrix((g[f"jac{i}_data"], g[f"jac{i}_indices"], g[f"jac{i}_indptr"]), shape=jac.shape)
        diff = jac - want
        assert diff.nnz == 0 or abs(diff).max() <= 1e-15 * abs(want).max()
    if "fcn" in g:
        res, solver = radau.comp_fcn(tm, g["y0"], return_solver=True)
        assert np.allclose(res, g["fcn"], rtol=1e-9, atol=1e-12)
        assert (solver.stats.nfev, solver.stats.njev, solver.stats.nlu) == (int(g["nfev"]), int(g["njev"]), int(g["nlu"]))


@pytest.mark.parametrize("tag", TAGS[:2])
def test_forced_file_precond(golden_dir, tag):
    """product formula of forced.apply_precond_jacobian with the tracer of the three time levels, and
    the backward-stable form of the same operator the HIP path implements"""
    from oracle.model import apply_precond_stable

    g = np.load(f"{golden_dir}/forced_{tag}.npz")
    tm = oracle_forced(g)
    states = list(g["precond_states"])
    res = tm.apply_precond(g["precond_v"], states=states)
    assert rel_err(res, g["precond_res"]) < 1e-6
    stable = apply_precond_stable(tm, g["precond_v"], states=states)
    assert rel_err(stable, g["precond_res"]) < 2e-3     # the explicit product is roundoff limited (test_oracle_precond.py)


def test_forcing_reader_matches_interp1d(tmp_path):
    """forcing.load_forcing = the spatial part of utils.gen_forcing_fcn: scale, then interp1d with
    extrapolation along every axis whose coordinate differs from the model's"""
    from nk_ooc_amd import ncio
    from nk_ooc_amd.forcing import interp_extrap, load_forcing

    rng = np.random.default_rng(3)
    time = np.linspace(0.0, 360.0, 7) * 86400.0
    depth_in = np.array([5.0, 20.0, 60.0, 150.0, 400.0, 1200.0])
    ypos_in = np.linspace(0.0, 5.0e6, 9)
    field = rng.standard_normal((7, 6, 9))
    fname = str(tmp_path / "forcing.nc")
    ncio.write_vars_file(fname, {"time": 7, "depth": 6, "ypos": 9},
                         {"time": (("time",), "f8", {}, time), "depth": (("depth",), "f8", {}, depth_in),
                          "ypos": (("ypos",), "f8", {}, ypos_in), "sms": (("time", "depth", "ypos"), "f8", {}, field)},
                         "test forcing")
    depth_out = np.array([2.0, 10.0, 100.0, 900.0, 2500.0])       # beyond both ends
    ypos_out = np.linspace(-1.0e5, 5.2e6, 12)
    times, vals = load_forcing(fname, "sms", [depth_out, ypos_out], scalef=2.5)
    want = 2.5 * field
    want = interpolate.interp1d(depth_in, want, axis=1, fill_value="extrapolate", assume_sorted=True)(depth_out)
    want = interpolate.interp1d(ypos_in, want, axis=2, fill_value="extrapolate", assume_sorted=True)(ypos_out)
    assert np.array_equal(times, time) and vals.shape == (7, 5, 12)
    assert np.allclose(vals, want, rtol=1e-13, atol=1e-15)
    # same axes: the records are returned untouched
    times, vals = load_forcing(fname, "sms", [depth_in, ypos_in])
    assert np.array_equal(vals, field)
    # time interpolation formula used on the device
    tq = np.array([-10.0, 0.0, 100.0, 359.0, 365.0]) * 86400.0
    got = interp_extrap(time, field, tq, 0)
    assert np.allclose(got, interpolate.interp1d(time, field, axis=0, fill_value="extrapolate", assume_sorted=True)(tq),
                       rtol=1e-13, atol=1e-15)
    with pytest.raises(ValueError):
        load_forcing(fname, "sms", [depth_out])


REF_INPUT = "/root/reference/input/py_driver_2d"


@pytest.mark.skipif(not __import__("os").path.exists(f"{REF_INPUT}/po4_sms.nc"), reason="reference input files absent")
def test_forcing_reader_on_reference_files():
    """the forcing files the reference ships (input/py_driver_2d/po4_surf.nc, po4_sms.nc: 61 records on
    its 40 x 50 default grid): untouched on that grid, interp1d-equal on another"""
    from nk_ooc_amd.forcing import load_forcing
    from oracle.grid import default_axes
    from oracle.model import forcing_on_model_axes
    from nk_ooc_amd import ncio

    depth, ypos = default_axes(40, 50)
    times, surf = load_forcing(f"{REF_INPUT}/po4_surf.nc", "po4", [ypos.mid])
    raw, _ = ncio.read_file(f"{REF_INPUT}/po4_surf.nc")
    assert surf.shape == (61, 50) and np.allclose(surf, raw["po4"], rtol=1e-13)
    assert np.array_equal(times, raw["time"]) and np.all(np.diff(times) > 0.0)
    depth2, ypos2 = default_axes(26, 30)
    raw, _ = ncio.read_file(f"{REF_INPUT}/po4_sms.nc")
    times, sms = load_forcing(f"{REF_INPUT}/po4_sms.nc", "po4_sms", [depth2.mid, ypos2.mid], scalef=0.5)
    want = forcing_on_model_axes(raw["po4_sms"], [raw["depth"], raw["ypos"]], [depth2.mid, ypos2.mid], scalef=0.5)
    assert sms.shape == (61, 26, 30)
    assert np.allclose(sms, want, rtol=1e-12, atol=1e-18 + 1e-13 * np.abs(want).max())
