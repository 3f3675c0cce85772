"""Host-side logic that needs no GPU: checkpoint format, NetCDF3 files, cfg / YAML
reading, grid generation, and that the C-ABI library loads and exports every symbol
include/nk2d.h declares."""
import json
import os
import re

import numpy as np
import pytest

from nk_ooc_amd import _lib, ncio
from nk_ooc_amd.grid import Grid2d
from nk_ooc_amd.model_config import ModelConfig
from nk_ooc_amd.setup_solver import gen_grid_vars_file, make_config
from nk_ooc_amd.solver_state import SolverState

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = os.path.join(ROOT, "tests", "golden", "ref_baselines")


def test_cabi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "nk2d.h")).read()
    body = header[header.index('extern "C"'):]
    declared = set(re.findall(r"\b(nk2d_[a-z0-9_]+)\s*\(", body))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()  # raises if the .so is missing or a symbol is not exported
    assert lib.nk2d_version().startswith(b"nk2d")


def test_grid_matches_golden(golden_dir):
    for tag in ("26x26", "30x30", "20x3_columns"):
        g = np.load(f"{golden_dir}/static_{tag}.npz")
        grid = Grid2d.default(int(g["nz"]), int(g["ny"]), float(g["max_abs_vvel"]),
                              float(g["horiz_mix_coeff"]))
        assert np.array_equal(grid.depth.edges, g["depth_edges"])
        assert np.array_equal(grid.ypos.delta_mid_r, g["ypos_delta_mid_r"])
        for nm in ("stream", "vvel", "wvel", "hmix_coeff"):
            assert np.array_equal(getattr(grid, nm), g[nm]), nm


def test_solver_state_format_and_resume(tmp_path):
    wd = str(tmp_path / "krylov_00")
    st = SolverState("Krylov", wd)
    st.log_step("KrylovSolver._solve0", per_iteration=False)
    beta = np.array([[0.25, 1.0 / 3.0]])
    st.set_value_saved_state("beta", beta)
    st.log_step(f"comp_fcn complete for {wd}/perturb_fcn_w_raw_00.nc")
    st.inc_iteration()
    raw = json.load(open(os.path.join(wd, "Krylov_state.json")))
    assert raw["iteration"] == 1
    assert raw["step_log"] == ["__init__", "KrylovSolver._solve0",
                               f"00:comp_fcn complete for {wd}/perturb_fcn_w_raw_00.nc",
                               "01:inc_iteration"]
    assert raw["beta"] == {"__ndarray__": beta.tolist()}
    text = open(os.path.join(wd, "Krylov_state.json")).read()
    assert text.startswith('{\n  "iteration": 1,\n  "step_log": [\n    "__init__",')
    st2 = SolverState("Krylov", wd, resume=True)
    assert st2.get_iteration() == 1
    assert np.array_equal(st2.get_value_saved_state("beta"), beta)
    assert st2.step_logged("KrylovSolver._solve0", per_iteration=False)
    assert not st2.step_logged("comp_fcn complete for x")
    st3 = SolverState("Krylov", wd, resume=True, rewind=True)
    assert st3.step_was_rewound("inc_iteration")
    with pytest.raises(RuntimeError):
        SolverState("Krylov", str(tmp_path / "x"), resume=False, rewind=True)


def test_reads_reference_checkpoint():
    """a Newton_state.json written by the reference parses with the same decoder"""
    fname = os.path.join(BASE, "ci_py_driver_2d_iage_column_regions", "Newton_state.json")
    import shutil, tempfile
    wd = tempfile.mkdtemp()
    shutil.copy(fname, os.path.join(wd, "Newton_state.json"))
    st = SolverState("Newton", wd, resume=True)
    assert st.get_iteration() == 2
    assert np.array_equal(st.get_value_saved_state("armijo_factor"), np.array([[1.0, 0.0, 0.0]]))
    assert st.step_logged("KrylovSolver instantiated") is False
    assert "01:KrylovSolver instantiated" in st.get_value_saved_state("step_log")


def test_netcdf3_roundtrip_and_reference_files(tmp_path):
    grid = Grid2d.default(20, 3, 0.0, 0.0)
    vals = {"iage": np.arange(60.0).reshape(20, 3), "iage_slow_rest": -np.arange(60.0).reshape(20, 3)}
    fname = str(tmp_path / "basis_00.nc")
    ncio.write_state_file(fname, [grid.depth, grid.ypos], vals, "h")
    ref = os.path.join(BASE, "ci_py_driver_2d_iage_column_regions", "basis_00.nc")
    got, _ = ncio.read_file(fname)
    want, _ = ncio.read_file(ref)
    # same variables (scipy's writer orders the header by shape; readers go by name)
    assert sorted(got) == sorted(want)
    for name in want:
        assert got[name].shape == want[name].shape and got[name].dtype == want[name].dtype
        if name not in vals:
            assert np.array_equal(got[name], want[name]), name  # axis variables identical
    assert np.array_equal(got["iage"], vals["iage"])
    with open(fname, "rb") as f1, open(ref, "rb") as f2:
        assert f1.read(4) == f2.read(4) == b"CDF\x02"  # NetCDF3 64-bit offset
    for name in ("depth", "ypos_edges"):
        assert ncio.read_var_attrs(fname, name) == ncio.read_var_attrs(ref, name)


def test_cfg_yaml_and_grid_vars(tmp_path):
    cfg = make_config(str(tmp_path), 20, 3,
                      extra_modelinfo={"max_abs_vvel": "0.0", "horiz_mix_coeff": "0.0"})
    assert cfg["solverinfo"]["krylov_rel_tol"] == "0.01"
    assert cfg["modelinfo"]["grid_vars_fname"] == os.path.join(str(tmp_path), "grid_vars.nc")
    assert cfg["modelinfo"]["tracer_module_defs_fname"].endswith("input/py_driver_2d/tracer_module_defs.yaml")
    gen_grid_vars_file(cfg["modelinfo"])
    got, _ = ncio.read_file(cfg["modelinfo"]["grid_vars_fname"])
    want, _ = ncio.read_file(os.path.join(BASE, "ci_py_driver_2d_iage_column_regions", "grid_vars.nc"))
    for name in want:
        assert np.array_equal(got[name], want[name]), name
    mc = ModelConfig(cfg["modelinfo"])
    assert mc.region_cnt == 3
    assert list(mc.tracer_module_defs["iage"]["tracers"]) == ["iage", "iage_slow_rest"]
    assert mc.precond_matrix_defs["phosphorus"]["hist_to_precond_varnames"] == ["po4", "time"]
    cfg2 = make_config(str(tmp_path), 8, 8, tracer_module_names="iage,forced_{suff}:a:b")
    gen_grid_vars_file(cfg2["modelinfo"])
    mc2 = ModelConfig(cfg2["modelinfo"])
    assert mc2.modelinfo["tracer_module_names"] == "iage,forced_a,forced_b"
    # the caller's dictionary keeps the unexpanded names: it configures the driver after the set-up
    assert cfg2["modelinfo"]["tracer_module_names"] == "iage,forced_{suff}:a:b"
    assert list(mc2.tracer_module_defs["forced_b"]["tracers"]) == ["b"]
    assert mc2.region_cnt == 1


def test_missing_extension_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libnk2d.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_baseline_cmp(tmp_path):
    """comparer with the reference's CLI: a reference-written file against itself, against a
    copy with one value moved beyond / within tolerance, and against a file with other metadata"""
    import shutil
    import subprocess
    import sys

    from scipy.io import netcdf_file

    from nk_ooc_amd import baseline_cmp

    base = os.path.join(os.path.dirname(__file__), "golden", "ref_baselines", "ci_py_driver_2d_iage")
    assert baseline_cmp.compare("fcn_0000.nc", base, base)
    work = str(tmp_path)
    shutil.copy(os.path.join(base, "fcn_0000.nc"), work)
    with netcdf_file(os.path.join(work, "fcn_0000.nc"), "a") as fptr:
        vals = fptr.variables["iage"][:].copy()
        vals[3, 4] = vals[3, 4] * (1.0 + 1.0e-5) + 1.0e-8
        fptr.variables["iage"][:] = vals
    assert not baseline_cmp.compare("fcn_0000.nc", work, base)
    assert baseline_cmp.compare("fcn_0000.nc", work, base, rtol=1.0e-3, atol=1.0e-6)
    assert not baseline_cmp.metadata_same(os.path.join(base, "fcn_0000.nc"), os.path.join(base, "grid_vars.nc"))
    # command line: exit status 0 / 1 as the reference's CI scripts expect
    cmd = [sys.executable, "-m", "nk_ooc_amd.baseline_cmp", "--fname", "fcn_0000.nc", "--expr_dir", work,
           "--baseline_dir", base]
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert subprocess.run(cmd, env=env, capture_output=True).returncode == 1
    assert subprocess.run(cmd + ["--rtol", "1e-3", "--atol", "1e-6"], env=env, capture_output=True).returncode == 0


def test_propagate_base_matrix_defs_to_all(tmp_path):
    """behaviour the reference pins in tests/test_model_config.py: base entries reach every
    matrix definition, options are keyed by their leading word, nothing is overridden or doubled"""
    from nk_ooc_amd.model_config import propagate_base_matrix_defs_to_all

    cfg = make_config(str(tmp_path), 8, 8, tracer_module_names="phosphorus")
    gen_grid_vars_file(cfg["modelinfo"])
    defs = ModelConfig(cfg["modelinfo"]).precond_matrix_defs
    base, phos = defs["base"], defs["phosphorus"]
    for varname in base["hist_to_precond_varnames"]:
        assert varname in phos["hist_to_precond_varnames"]
    base["hist_to_precond_varnames"].append("new_hist_var")
    propagate_base_matrix_defs_to_all(defs)
    assert "new_hist_var" in phos["hist_to_precond_varnames"]
    base["precond_matrices_opts"] = ["matrix_opt_A sub_opt"]
    propagate_base_matrix_defs_to_all(defs)
    assert "matrix_opt_A sub_opt" in phos["precond_matrices_opts"]
    base["precond_matrices_opts"].append("matrix_opt_B sub_opt_base")
    phos["precond_matrices_opts"].append("matrix_opt_B sub_opt_phosphorus")
    propagate_base_matrix_defs_to_all(defs)
    assert "matrix_opt_B sub_opt_phosphorus" in phos["precond_matrices_opts"]
    assert "matrix_opt_B sub_opt_base" not in phos["precond_matrices_opts"]
    assert phos["precond_matrices_opts"].count("matrix_opt_A sub_opt") == 1


@pytest.mark.parametrize("units_str, expected", [
    ("years m", "years m"), ("mmol / m^3 m", "mmol / m^2"), ("mmol / m^3 / d m", "mmol / m^2 / d"),
    ("1 / d m", "m / d"), ("mol / m^3 m", "mol / m^2"), ("(years) (m)", "years m"),
    ("(mmol / m^3) (m)", "mmol / m^2"), ("(mmol / m^3 / d) (m)", "mmol / m^2 / d"), ("(1 / d) (m)", "m / d"),
    ("(mol / m^3) (m)", "mol / m^2"), ("m years", "years m"), ("m mmol / m^3", "mmol / m^2"),
    ("m mmol / m^3 / d", "mmol / m^2 / d"), ("m 1 / d", "m / d"), ("m mol / m^3", "mol / m^2")])
def test_units_str_format(units_str, expected):
    """the cases the reference pins for its pint-based formatter (tests/test_utils.py:28-50)"""
    from nk_ooc_amd.hist import units_str_format

    assert units_str_format(units_str) == expected


@pytest.mark.parametrize("expr, expected", [
    ("1.0 + 2.0", 3.0), ("1.0 + 2.0 * 3.0", 7.0), ("(1.0 + 2.0) * 3.0", 9.0), ("(1.0 + 2.0) / 3.0", 1.0),
    ("2.0 ** 3.0", 8.0), ("10.0 + -2.0", 8.0), ("10.0 - 2.0", 8.0)])
def test_eval_expr(expr, expected):
    """the cases the reference pins for utils.eval_expr (tests/test_utils.py:11-25)"""
    from nk_ooc_amd.engine import _eval_number

    assert _eval_number(expr) == expected


def test_limiter_known_answers():
    """the Newton increment limiter against the cases the reference pins for utils.min_by_region /
    comp_scalef_lob / comp_scalef_upb (tests/test_utils.py:78-221)"""
    from nk_ooc_amd.limiter import region_min, scalef_for_bound

    vals = np.arange(24.0).reshape(4, 6)
    rows, cols = np.indices(vals.shape)
    for mask, expected in ((np.ones(vals.shape, dtype=np.int32), [0.0]),
                           (1 + rows, vals[:, 0]), (1 + rows // 2, vals[::2, 0]),
                           (1 + cols, vals[0, :]), (1 + cols // 2, vals[0, ::2])):
        mask = mask.astype(np.int32)
        assert np.array_equal(region_min(int(mask.max()), mask, vals), np.asarray(expected))
    assert region_min(2, np.ones((2, 2), dtype=np.int32), np.ones((2, 2)))[1] == np.inf   # empty region

    # seven regions (columns), three cells each; lower bound 0
    nreg = 7
    mask = np.tile(np.arange(1, nreg + 1, dtype=np.int32), (3, 1))
    base = np.ones((3, nreg))
    inc = np.ones((3, nreg))
    inc[0, 1] = -0.5                                   # stays above the bound
    inc[0:2, 2] = [-0.5, -1.0]                         # reaches the bound exactly
    inc[:, 3] = [-0.5, -1.0, -2.0]                     # crosses it: half the increment fits
    base[:, 4:] = 0.0                                  # base on the bound ...
    inc[0, 5] = 0.0                                    # ... zero increments are harmless
    inc[0:2, 6] = [0.0, -1.0]                          # ... a negative one cannot be taken at all
    expected = np.array([1.0, 1.0, 1.0, 0.5, 1.0, 1.0, 0.0])
    assert np.array_equal(scalef_for_bound(nreg, mask, base, inc, 0.0, upper=False), expected)
    assert np.array_equal(scalef_for_bound(nreg, mask, -base, -inc, -0.0, upper=True), expected)
    assert np.array_equal(scalef_for_bound(nreg, mask, base, inc, None, upper=False), np.ones(nreg))
    with pytest.raises(ValueError, match="base < lob"):
        scalef_for_bound(nreg, mask, base - 2.0, inc, 0.0, upper=False)
    with pytest.raises(ValueError, match="base > upb"):
        scalef_for_bound(nreg, mask, base + 2.0, inc + 1.0, 1.5, upper=True)


def test_isclose_all_vars_reference_files(golden_dir):
    """baseline_cmp.isclose_all_vars on the three data files the reference's own test holds
    (input/tests/isclose_{base,same,diff}.nc, tests/test_utils.py:53-76): `same` stores var2 in cm instead
    of m, `diff` also perturbs var1 by 1e-7"""
    from nk_ooc_amd.baseline_cmp import isclose_all_vars
    from nk_ooc_amd.hist import units_conversion_factor

    base, same, diff = (os.path.join(golden_dir, "ref_tests_input", f"isclose_{tag}.nc") for tag in ("base", "same", "diff"))
    for rtol, atol in ((0.0, 0.0), (1.0e-5, 1.0e-5)):
        assert isclose_all_vars(base, base, rtol=rtol, atol=atol)
        assert isclose_all_vars(base, same, rtol=rtol, atol=atol)        # equal once the units are accounted for
    assert not isclose_all_vars(base, diff, rtol=0.0, atol=0.0)
    assert not isclose_all_vars(base, diff, rtol=1.0e-8, atol=1.0e-8)
    assert isclose_all_vars(base, diff, rtol=1.0e-5, atol=1.0e-5)
    assert units_conversion_factor("m", "cm") == 100.0
    assert units_conversion_factor("mmol / m^3", "mol / m^3") == 1.0e-3
    assert units_conversion_factor("m / d", "m / s") == 1.0 / 86400.0
    assert units_conversion_factor("(mmol / m^3) (m)", "mmol / m^2") == 1.0
    assert units_conversion_factor("m", "s") is None and units_conversion_factor("furlong", "m") is None


def test_host_arithmetic_under_address_sanitizer():
    """csrc/nk2d_hostmath.h (interpolation, interval bracketing, the fixed-order sum, the Hessenberg least squares)
    built with a plain g++ under -fsanitize=address,undefined and run on the CPU: `make asan-host`"""
    import shutil
    import subprocess

    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no host compiler")
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "newton-krylov_ooc_amd", "csrc")
    res = subprocess.run(["make", "-C", csrc, "asan-host"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "hostmath ok" in res.stdout


def test_solver_state_file_is_what_json_dump_writes(tmp_path):
    """the state file is put together from cached per-value text (an indented dump runs in json's pure-Python encoder, six
    times per Krylov iteration): byte for byte what json.dump(data, indent=2) writes, after every kind of change"""
    import io
    import json

    from nk_ooc_amd.solver_state import _to_json

    st = SolverState("Krylov", str(tmp_path / "wd"))
    rng = np.random.default_rng(0)

    def check():
        want = io.StringIO()
        json.dump(st._data, want, indent=2, default=_to_json)
        with open(st._path) as fptr:
            assert fptr.read() == want.getvalue()

    check()
    for it in range(5):
        st.inc_iteration()
        check()
        st.log_step('a "step"\nwith a newline')
        check()
        st.set_value_saved_state("h_mat", rng.standard_normal((2, it + 2, it + 1, 1)))
        check()
        st.set_value_saved_state("beta", rng.standard_normal((2, 1)))     # same key, same shape, new values
        check()
        st.set_value_saved_state("flag", bool(it % 2))
        st.set_value_saved_state("name", f'x"y{it}')
        st.set_value_saved_state("lst", [1, 2.5, "a", [3, 4, it]])
        st.set_value_saved_state("empty", [])
        check()
    again = SolverState("Krylov", str(tmp_path / "wd"), resume=True)
    assert np.array_equal(again.get_value_saved_state("h_mat"), st.get_value_saved_state("h_mat"))
