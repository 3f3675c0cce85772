"""SURVEY.md section 8(b), in-container drop-in check: the reference's REAL driver (nk_ooc/nk_driver.py:38-67 ->
NewtonSolver -> KrylovSolver -> SolverState, unmodified, imported from /root/reference) drives this repository's
plugin package `py_driver_2d_hip` (tests/ref_plugin/, INTEGRATION.md variant B), found through `nk_ooc.__path__`
exactly as `get_model_state_class` looks plugins up (model_state_base.py:627-667).

This container has no GPU and the GPU box has no reference tree, so the plugin's device calls are answered here
by a stand-in installed with `_backend.set_backend` (SciPy's Radau on the oracle's tendency functions -- bit-identical
to the reference's forward year -- and the reference's preconditioner formula).  What is under test is everything
else: plugin discovery, constructor plumbing, the re-routing of `integrate.solve_ivp` and of
`iage.apply_precond_jacobian`, file formats and step logs under the real solvers.  With that stand-in the
plugin run must reproduce the reference's own run of the same case BIT FOR BIT, file by file.
Skipped where /root/reference is absent."""
import json
import os
import sys

import numpy as np
import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "nk_ooc")), reason="needs the reference tree")


class OracleBackend:
    """answers the plugin's two device calls on the CPU"""

    def __init__(self, nz, ny, vv, kh):
        from helpers import oracle_iage

        self.model, self.tm = oracle_iage(nz, ny, vv, kh)
        self.calls = {"forward_year": 0, "precond_apply": 0}

    def forward_year(self, tracer_module, y0, t_eval):
        from scipy import integrate

        assert tracer_module.name == "iage"
        self.calls["forward_year"] += 1
        year = 365.0 * 86400.0
        sol = integrate.solve_ivp(self.tm.comp_tend, (0.0, year), np.asarray(y0).reshape(-1), "Radau", t_eval,
                                  max_step=year * 0.01, atol=1.0e-6, rtol=1.0e-6, jac=self.tm.comp_jacobian)
        return sol.t, sol.y

    def precond_apply(self, tracer_module, vals):
        self.calls["precond_apply"] += 1
        return self.tm.apply_precond(np.asarray(vals).reshape(-1))


def _run(workdir, model_name, nz, ny):
    from nk_ooc import nk_driver
    from nk_ooc.py_driver_2d import setup_solver

    os.makedirs(workdir, exist_ok=True)
    override = os.path.join(workdir, "override.cfg")
    input_dir = os.path.join(REF, "input", "py_driver_2d")
    with open(override, "w") as fptr:
        fptr.write(f"[DEFAULT]\nmodel_name = {model_name}\n\n[modelinfo]\ndepth_nlevs = {nz}\nypos_nlevs = {ny}\n"
                   "max_abs_vvel = 0.0\nhoriz_mix_coeff = 0.0\nreinvoke = False\n"
                   f"tracer_module_defs_fname = {input_dir}/tracer_module_defs.yaml\n\n"
                   "[solverinfo]\nnewton_max_iter = 1\n")
    cfgs = ",".join([os.path.join(input_dir, "newton_krylov.cfg"), os.path.join(input_dir, "model_params.cfg"), override])
    # `reinvoke = False` in the override is what --persist sets (share.py:24-30 offers that flag for the
    # reference's own two model names only)
    common = ["--tracer_module_names", "iage", "--cfg_fnames", cfgs, "--workdir", workdir]
    # inputs (grid_vars, init_iterate) always come from the reference's own set-up module
    setup_solver.main(setup_solver.parse_args(["--fp_cnt", "1", "--model_name", "py_driver_2d"] + common))
    with pytest.raises(RuntimeError, match="maximum Newton iterations"):
        nk_driver.main(nk_driver.parse_args(["--model_name", model_name] + common))


def _vec(fname):
    from scipy.io import netcdf_file

    with netcdf_file(fname, "r", mmap=False) as fptr:
        return np.stack([np.array(fptr.variables[name].data) for name in ("iage", "iage_slow_rest")])


def test_reference_driver_runs_the_plugin(tmp_path, monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ref_harness import shims

    shims.install()
    monkeypatch.syspath_prepend(REF)
    monkeypatch.setenv("USER", os.environ.get("USER", "nk2d"))
    import nk_ooc

    plugin_dir = os.path.join(ROOT, "tests", "ref_plugin")
    if plugin_dir not in nk_ooc.__path__:
        nk_ooc.__path__.append(plugin_dir)
    from nk_ooc.model_state_base import get_model_state_class
    from nk_ooc.py_driver_2d.model_state import ModelState as RefModelState
    from nk_ooc.py_driver_2d_hip import _backend
    from nk_ooc.py_driver_2d_hip.model_state import ModelState as PluginModelState

    import logging

    assert get_model_state_class("py_driver_2d_hip", logging.DEBUG) is PluginModelState
    assert issubclass(PluginModelState, RefModelState)

    nz, ny = 20, 3          # the ci_py_driver_2d_iage_column_regions case: three column regions
    ref_dir, plug_dir = str(tmp_path / "ref"), str(tmp_path / "plugin")
    _run(ref_dir, "py_driver_2d", nz, ny)
    RefModelState.class_vars_set = False        # new work directory: axes are read again
    backend = OracleBackend(nz, ny, 0.0, 0.0)
    _backend.set_backend(backend)
    try:
        _run(plug_dir, "py_driver_2d_hip", nz, ny)
    finally:
        _backend.set_backend(None)
        RefModelState.class_vars_set = False
    # the plugin answered every forward year and every preconditioner apply of the Newton iteration
    assert backend.calls["forward_year"] >= 4 and backend.calls["precond_apply"] >= 2
    # same files, bit for bit
    names = ["fcn_00.nc", "increment_00.nc", "iterate_01.nc", "fcn_01.nc"]
    kdir = "krylov_00"
    names += [os.path.join(kdir, n) for n in sorted(os.listdir(os.path.join(ref_dir, kdir)))
              if n.endswith(".nc") and not n.startswith(("Krylov_stats", "precond_00"))]
    for name in names:
        assert np.array_equal(_vec(os.path.join(plug_dir, name)), _vec(os.path.join(ref_dir, name))), name
    for name in ("Newton_state.json", os.path.join(kdir, "Krylov_state.json")):
        got = json.loads(open(os.path.join(plug_dir, name)).read().replace(plug_dir, "$workdir"))
        want = json.loads(open(os.path.join(ref_dir, name)).read().replace(ref_dir, "$workdir"))
        assert got == want, name
    assert got["iteration"] >= 1 and np.asarray(got["h_mat"]["__ndarray__"]).shape[-1] == 3
