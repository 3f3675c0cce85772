"""Newton outer loop on the GPU (row f1): the ci_py_driver_2d_iage_column_regions case run to
convergence, compared with the reference's committed baselines (iterate_01 at the CI script's
rtol 1.9e-2; Newton_state.json step log) and with an oracle Newton iteration on the CPU."""
import json
import os

import numpy as np
import pytest

from helpers import oracle_iage
from oracle import krylov

pytestmark = pytest.mark.gpu

BASE = os.path.join(os.path.dirname(__file__), "golden", "ref_baselines", "ci_py_driver_2d_iage_column_regions")


def _read_state(fname):
    from nk_ooc_amd import ncio

    data, _ = ncio.read_file(fname, ["iage", "iage_slow_rest"])
    return np.stack([data["iage"], data["iage_slow_rest"]]).reshape(-1)


def test_newton_column_regions(tmp_path):
    from nk_ooc_amd import ncio, nk_driver
    from nk_ooc_amd.model_state import ModelState
    from nk_ooc_amd.setup_solver import make_config, setup

    workdir = str(tmp_path)
    cfg = make_config(workdir, 20, 3, extra_modelinfo={"max_abs_vvel": "0.0", "horiz_mix_coeff": "0.0"})
    ModelState.write_files = True
    setup(cfg, fp_cnt=1)
    # the set-up files reproduce the reference's (CI tolerance for the forward year)
    gen = os.path.join(workdir, "gen_init_iterate")
    assert np.array_equal(_read_state(os.path.join(gen, "init_iterate_0000.nc")),
                          _read_state(os.path.join(BASE, "init_iterate_0000.nc")))
    assert np.allclose(_read_state(os.path.join(gen, "fcn_0000.nc")),
                       _read_state(os.path.join(BASE, "fcn_0000.nc")), rtol=1e-3, atol=1e-6)
    assert np.allclose(_read_state(cfg["solverinfo"]["init_iterate_fname"]),
                       _read_state(os.path.join(BASE, "init_iterate.nc")), rtol=1e-3, atol=1e-6)

    # history file of the set-up year against the reference's committed hist_0000.nc
    # (scripts/ci_py_driver_2d_iage_column_regions.sh compares it at atol 1e-6, rtol 1e-3)
    got_hist, _ = ncio.read_file(os.path.join(gen, "hist_0000.nc"))
    want_hist, _ = ncio.read_file(os.path.join(BASE, "hist_0000.nc"))
    assert set(want_hist) <= set(got_hist), set(want_hist) - set(got_hist)
    for name, want in want_hist.items():
        assert got_hist[name].shape == want.shape, name
        assert np.all(np.isclose(got_hist[name], want, rtol=1.0e-3, atol=1.0e-6)), name

    solver = nk_driver.run(cfg)
    assert solver.converged().all()
    n_newton = solver.get_iteration()
    assert 1 <= n_newton <= 3

    # files of the first Newton iteration against the reference's committed ones
    for name, kw in (("increment_00.nc", dict(rtol=1.9e-2, atol=2e-9)), ("iterate_01.nc", dict(rtol=1.9e-2, atol=2e-9))):
        assert np.all(np.isclose(_read_state(os.path.join(workdir, name)),
                                 _read_state(os.path.join(BASE, name)), **kw)), name

    # the reference CI's own check (scripts/ci_py_driver_2d_iage_column_regions.sh): baseline_cmp per file
    from nk_ooc_amd import baseline_cmp

    for name, rtol, atol in (("iterate_01.nc", 1.9e-2, 2e-9), ("increment_00.nc", 1.9e-2, 2e-9)):
        assert baseline_cmp.compare(name, workdir, BASE, rtol=rtol, atol=atol), name
    assert baseline_cmp.compare("hist_0000.nc", gen, BASE, rtol=1.0e-3, atol=1.0e-6)

    # Newton_state.json: same schema and the same sequence of checkpointed actions
    got = json.load(open(os.path.join(workdir, "Newton_state.json")))
    want = json.load(open(os.path.join(BASE, "Newton_state.json")))
    norm = lambda s: s.replace(workdir, "WORKDIR")
    want_log = [s.replace("HOME/ci_py_driver_2d_iage_column_regions_workdir", "WORKDIR") for s in want["step_log"]]
    got_log = [norm(s) for s in got["step_log"]]
    # the reference run needed 2 Newton iterations; compare the log of the iterations both ran
    n_cmp = min(n_newton, want["iteration"])
    cut = lambda log: [s for s in log if not s[:2].isdigit() or int(s[:2]) < n_cmp]
    assert cut(got_log) == cut(want_log)
    assert set(got) == set(want)
    assert np.asarray(got["armijo_factor"]["__ndarray__"]).shape == (1, 3)

    # history / stats files exist with the reference's variables
    hist, _ = ncio.read_file(os.path.join(workdir, "hist_00.nc"))
    for name in ("time", "bldepth", "vert_mixing_coeff", "iage", "iage_time_mean", "iage_slow_rest_depth_ypos_int"):
        assert name in hist, name
    assert hist["iage"].shape == (61, 20, 3)
    stats, _ = ncio.read_file(os.path.join(workdir, "Newton_stats.nc"))
    for name in ("iterate_norm_iage", "fcn_mean_iage", "increment_norm_iage", "Armijo_factor_iage",
                 "Krylov_iterations", "iage", "iage_mean_ypos"):
        assert name in stats, name
    assert stats["iterate_norm_iage"].shape == (n_newton + 1, 3)

    # converged iterate against an oracle Newton run (CPU): F(x*) = 0 to newton_rel_tol 1e-5
    model, tm = oracle_iage(20, 3, 0.0, 0.0)
    grid, _ = ncio.read_file(os.path.join(workdir, "grid_vars.nc"))
    mod = krylov.OracleModule(tm, krylov.Regions(grid["region_mask"], grid["grid_weight"]), precond="stable")
    x = _read_state(cfg["solverinfo"]["init_iterate_fname"])
    for _ in range(n_newton):
        f = mod.comp_fcn(x)
        inc, _ = krylov.krylov_solve([mod], [x], [f], rel_tol=0.01)
        prov = x + inc[0]
        x = prov + mod.comp_fcn(prov)  # Armijo factor 1, one post-Newton fixed-point iteration
    x_gpu = solver.iterate.tracer_modules[0].get_tracer_vals_all().reshape(-1)
    # (i) two independent Newton runs agree well inside the reference CI tolerance for iterates
    # (1.9e-2); they stop at |F| < 1e-5 |x|, and slow deep-ocean modes turn that into ~1e-3 in x
    assert np.max(np.abs(x_gpu - x)) <= 2e-3 * np.max(np.abs(x)), np.max(np.abs(x_gpu - x))
    # (ii) the GPU's converged iterate satisfies the reference's convergence test when F is
    # evaluated by the CPU oracle: |F(x)| < newton_rel_tol |x| per region (factor 2 for the 1e-6
    # integrator noise of two free-running forward years)
    f_cpu = mod.comp_fcn(x_gpu)
    fn = np.sqrt(mod.dot(f_cpu, f_cpu))
    xn = np.sqrt(mod.dot(x_gpu, x_gpu))
    assert np.all(fn < 2.0 * 1.0e-5 * xn), (fn, xn)
    ModelState.reset_class()


@pytest.mark.parametrize("stop_at", ["prov_fcn_Armijo_00", "perturb_fcn_w_raw_01", "prov_fcn_fp_01"])
def test_newton_resume_after_interruption(tmp_path, stop_at):
    """the out-of-core contract of the Newton loop: a run killed inside a forward year (line
    search, second Krylov iteration, fixed-point year) and resumed from its JSON / NetCDF trail in
    a fresh set of contexts ends with the same iterate, bit for bit, and the same step log"""
    from nk_ooc_amd import nk_driver
    from nk_ooc_amd.model_state import ModelState
    from nk_ooc_amd.setup_solver import make_config, setup

    class Interrupt(Exception):
        pass

    def run(workdir, interrupt):
        cfg = make_config(workdir, 20, 3, extra_modelinfo={"max_abs_vvel": "0.0", "horiz_mix_coeff": "0.0"},
                          extra_solverinfo={"post_newton_fp_iter": "2"})
        ModelState.write_files = True
        setup(cfg, fp_cnt=1)
        if not interrupt:
            return nk_driver.run(cfg), cfg
        original = ModelState.comp_fcn

        def guarded(self, res_fname, solver_state, hist_fname=None, **kw):
            if stop_at in os.path.basename(res_fname):
                raise Interrupt(res_fname)
            return original(self, res_fname, solver_state, hist_fname, **kw)

        ModelState.comp_fcn = guarded
        try:
            with pytest.raises(Interrupt):
                nk_driver.run(cfg)
        finally:
            ModelState.comp_fcn = original
        return nk_driver.run(cfg, resume=True), cfg

    straight, cfg_a = run(str(tmp_path / "a"), False)
    x_a = straight.iterate.tracer_modules[0].get_tracer_vals_all()
    n_a = straight.get_iteration()
    log_a = json.load(open(os.path.join(cfg_a["solverinfo"]["workdir"], "Newton_state.json")))["step_log"]
    resumed, cfg_b = run(str(tmp_path / "b"), True)
    x_b = resumed.iterate.tracer_modules[0].get_tracer_vals_all()
    log_b = json.load(open(os.path.join(cfg_b["solverinfo"]["workdir"], "Newton_state.json")))["step_log"]
    assert resumed.get_iteration() == n_a
    assert np.array_equal(x_a, x_b)
    norm = lambda log, cfg: [s.replace(cfg["solverinfo"]["workdir"], "WORK") for s in log]
    assert norm(log_a, cfg_a) == norm(log_b, cfg_b)
    ModelState.reset_class()


def test_ci_py_driver_2d_iage_setup(tmp_path):
    """scripts/ci_py_driver_2d_iage.sh on the GPU: the 30 x 30 set-up with one fixed-point year, compared file by
    file with the reference's committed baselines by the comparer with the reference's CLI semantics
    (grid_vars.nc at the default tolerances; fcn_0000 / init_iterate / init_iterate_0000 at atol 1e-6,
    rtol 1e-3); hist_0000.nc, 2.3 MB in the reference tree, against a reduced copy of it (every variable; the 3-d ones
    at every 10th time sample, tests/golden/gen_hist_subset.py) at the same tolerances"""
    from nk_ooc_amd import baseline_cmp
    from nk_ooc_amd.model_state import ModelState
    from nk_ooc_amd.setup_solver import make_config, setup

    base = os.path.join(os.path.dirname(__file__), "golden", "ref_baselines", "ci_py_driver_2d_iage")
    workdir = str(tmp_path)
    cfg = make_config(workdir, 30, 30)
    ModelState.write_files = True
    try:
        setup(cfg, fp_cnt=1)
    finally:
        ModelState.reset_class()
    assert baseline_cmp.compare("grid_vars.nc", workdir, base)
    gen = os.path.join(workdir, "gen_init_iterate")
    for fname in ("fcn_0000.nc", "init_iterate_0000.nc"):
        assert baseline_cmp.compare(fname, gen, base, rtol=1.0e-3, atol=1.0e-6), fname
    assert baseline_cmp.compare("init_iterate.nc", os.path.dirname(cfg["solverinfo"]["init_iterate_fname"]), base,
                                rtol=1.0e-3, atol=1.0e-6)
    from nk_ooc_amd import ncio

    want = np.load(os.path.join(base, "hist_0000_subset.npz"))
    got, _ = ncio.read_file(os.path.join(gen, "hist_0000.nc"))
    dims = ncio.read_var_dims(os.path.join(gen, "hist_0000.nc"), list(got))
    names = [key[4:] for key in want.files if key.startswith("var_")]
    assert set(names) <= set(got), set(names) - set(got)
    for name in names:
        ref = want["var_" + name]
        assert ",".join(dims[name]) == str(want["dims_" + name]), name
        val = got[name][want["time_index"]] if got[name].ndim == 3 else got[name]
        assert val.shape == ref.shape, name
        assert np.all(np.isclose(val, ref, rtol=1.0e-3, atol=1.0e-6)), (name, float(np.max(np.abs(val - ref))))
