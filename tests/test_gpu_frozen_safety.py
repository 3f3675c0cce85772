"""The safety net of the products on frozen years (round-2 ADVICE items 1, 3, 5):

* a frozen year whose recorded Newton iteration count is not enough at some step is RESUMED from the checkpoint before
  that step with one more iteration there (at most twice), instead of being thrown away for a free-running year;
* what cannot be repaired -- or a step whose sampled error estimate is out of bounds -- returns -7: `nk2d_jvp` then runs a
  free-running year (counted), with a norm hook (sharded module) the error reaches the caller;
* a schedule carries the fingerprint of the context, options and build that recorded it; a frozen year refuses any other (-8);
* the side files of `ModelState` are tied to the values they belong to and forgotten when the name is written again.
"""
import os

import numpy as np
import pytest

from helpers import oracle_iage, rel_err

pytestmark = pytest.mark.gpu


def _engine(n):
    from nk_ooc_amd.engine import iage_engine
    from nk_ooc_amd.grid import Grid2d

    eng = iage_engine(Grid2d.default(n, n))
    eng.set_option("device_ctl", 0)
    model, tm = oracle_iage(n, n)
    eng.set_region(np.ones((n, n), dtype=np.int32), np.outer(model.depth.delta, model.ypos.delta))
    return eng, model, tm


def _state(eng, model, n, seed=23):
    rng = np.random.default_rng(seed)
    col = np.interp(model.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x0 = np.stack([np.broadcast_to(col[:, None], (n, n))] * 2) + 0.01 * rng.standard_normal((2, n, n))
    v = np.cumsum(rng.standard_normal(x0.shape), axis=1)
    x, vd = eng.upload(x0), eng.upload(v)
    vd = eng.scale(vd, 1.0 / np.sqrt(eng.dot(vd, vd)))
    return x0, x, eng.download(vd), vd


def _starved(sched, lo=200, drop=2):
    """the schedule with `drop` Newton iterations fewer at the first step beyond row `lo` that has more than `drop`"""
    bad = sched.copy()
    k = next(i for i in range(lo, len(bad)) if bad[i, 3] >= drop + 1)
    bad[k, 3] -= drop
    return bad, k


def test_schedule_rows_carry_error_and_fingerprint():
    eng, model, _ = _engine(26)
    _, x, _, _ = _state(eng, model, 26)
    _, st, sched = eng.comp_fcn(x, record=True)
    assert sched.shape == (st["nsteps"], 8)
    assert np.all(sched[:, 6] > 0.0) and np.all(sched[:, 6] <= 1.0)          # every recorded step was accepted
    fp = eng.schedule_fingerprint()
    assert fp >= 1.0 and fp == float(int(fp)) and np.all(sched[:, 7] == fp)
    # a year by launches stamps its steps the same way as one that ran as a command stream
    eng.set_option("stream_years", 0)
    _, _, sched3 = eng.comp_fcn(x, record=True)
    assert np.all(sched3[:, 7] == fp) and np.all(sched3[:, 6] <= 1.0) and np.all(sched3[:, 6] > 0.0)
    eng.close()


@pytest.mark.parametrize("n", [26, 104])
def test_starved_step_is_resumed_from_a_checkpoint(n):
    """Newton iterations taken away at one step in the middle of the year: the check after the year finds the step and the
    year resumes from the checkpoint before it with one more iteration there, until the step passes SciPy's convergence
    test (with the slack of the check: one iteration short of the recorded count may pass) -- no free-running year is
    needed, and a year brought back to the recorded counts is the recorded year bit for bit"""
    eng, model, _ = _engine(n)
    _, x, _, _ = _state(eng, model, n)
    fx, st, sched = eng.comp_fcn(x, record=True)
    assert len(sched) > 400
    want = eng.download(fx)
    scale = np.max(np.abs(want))
    seen = set()
    for drop in (1, 2, 3):
        if not np.any(sched[200:, 3] >= drop + 1):
            continue
        bad, k = _starved(sched, drop=drop)
        before = eng.frozen_resumes()
        fx2, st2 = eng.comp_fcn_frozen(x, bad)
        resumed = st2["nresumed"]
        seen.add(resumed)
        assert 0 <= resumed <= min(drop, 2) and eng.frozen_resumes() - before == resumed
        assert eng.frozen_fallbacks() == 0
        got = eng.download(fx2)
        if resumed == drop:
            assert np.array_equal(got, want)                   # back at the recorded counts: the recorded year
        # a step accepted by the check has converged to SciPy's tolerance (x slack): the year is the recorded one to that
        assert np.max(np.abs(got - want)) < 1e-6 * scale, (drop, resumed)
        # work: the year plus, per resume, the part from the checkpoint before step k
        if resumed:
            assert st2["nsteps"] >= st["nsteps"] + resumed * (len(sched) - k)
    assert max(seen) >= 1                                      # at least one of the starved schedules had to be repaired
    eng.close()


def test_resumed_product_matches_the_oracles_frozen_product():
    """the product of a perturbed state on a starved schedule: resumed twice, it is the product on the recorded steps -- the
    CPU oracle's, which differences two replays of the ORIGINAL schedule"""
    from oracle import radau

    n = 26
    eng, model, tm = _engine(n)
    x0, x, v, vd = _state(eng, model, n)
    fx, _, sched = eng.comp_fcn(x, record=True)
    bad, _ = _starved(sched, drop=3) if np.any(sched[200:, 3] >= 4) else _starved(sched)
    w, sigma, stp = eng.jvp(x, fx, vd, sched=bad)
    assert stp["nresumed"] >= 1 and eng.frozen_fallbacks() == 0
    rows = [(r[0], r[1], r[2], int(r[3]), r[4], r[5]) for r in sched]
    f0 = radau.comp_fcn(tm, x0.reshape(-1), replay=rows)
    f1 = radau.comp_fcn(tm, (x0 + sigma[0] * v).reshape(-1), replay=rows)
    assert rel_err(eng.download(w).reshape(-1), (f1 - f0) / sigma[0]) < 2e-3
    eng.close()


def test_unrepairable_schedule_falls_back_and_is_counted():
    """one iteration everywhere: two resumes cannot repair it.  comp_fcn_frozen raises (-7); nk2d_jvp runs a free-running
    year instead and counts; with a norm hook (a shard of a sharded module cannot fall back alone) the error is the caller's"""
    from nk_ooc_amd.engine import Nk2dFrozenMismatch

    n = 26
    eng, model, _ = _engine(n)
    _, x, _, vd = _state(eng, model, n)
    fx, _, sched = eng.comp_fcn(x, record=True)
    bad = sched.copy()
    bad[:, 3] = 1.0
    with pytest.raises(Nk2dFrozenMismatch):
        eng.comp_fcn_frozen(x, bad)
    assert eng.frozen_fallbacks() == 1 and eng.frozen_resumes() == 2
    w, sigma, stp = eng.jvp(x, fx, vd, sched=bad)
    assert eng.frozen_fallbacks() == 2
    w_free, _, _ = eng.jvp(x, fx, vd, sched=None)
    assert np.array_equal(eng.download(w), eng.download(w_free))
    eng.set_norm_hook(lambda val: val, 2.0 * n * n)
    with pytest.raises(Nk2dFrozenMismatch):
        eng.jvp(x, fx, vd, sched=bad)
    eng.set_norm_hook(None, 0.0)
    eng.close()


def test_schedule_of_other_options_is_refused():
    from nk_ooc_amd.engine import Nk2dScheduleMismatch

    n = 26
    eng, model, _ = _engine(n)
    _, x, _, vd = _state(eng, model, n)
    fx, _, sched = eng.comp_fcn(x, record=True)
    fp = eng.schedule_fingerprint()
    for name, val, back in (("lin_tol", 1.0e-2, 3.0e-2), ("jac_stage", 0, 1), ("min_sweeps", 2, 1)):
        eng.set_option(name, val)
        assert eng.schedule_fingerprint() != fp, name
        with pytest.raises(Nk2dScheduleMismatch):
            eng.comp_fcn_frozen(x, sched)
        eng.set_option(name, back)
        assert eng.schedule_fingerprint() == fp
    # a schedule from elsewhere (no fingerprint: the oracle's steps) is for step-replay mode only
    with pytest.raises(Nk2dScheduleMismatch):
        eng.comp_fcn_frozen(x, sched[:, :6])
    # another grid
    eng2, model2, _ = _engine(30)
    assert eng2.schedule_fingerprint() != fp
    # the product falls back to free-running years (nothing is counted as a rejected frozen year: none ran)
    eng.set_option("lin_tol", 1.0e-2)
    w, _, stp = eng.jvp(x, fx, vd, sched=sched)
    assert eng.frozen_fallbacks() == 0 and stp["nresumed"] == 0
    eng.close()
    eng2.close()


@pytest.mark.parametrize("one_launch", [0, 1])
def test_error_estimates_of_frozen_years(one_launch):
    """SciPy's error estimate on the steps of a frozen year (sampled by default, every step on request): for the perturbed
    state of a product it stays where the recorded steps were accepted (<= 1, give or take sigma); a state the recorded steps
    were not made for is refused.  Launch by launch, and in the one-launch year (three phases of their own around the
    sampled steps; single-sweep steps only)"""
    from nk_ooc_amd.engine import Nk2dFrozenMismatch

    n = 52
    eng, model, _ = _engine(n)
    eng.set_option("frozen_persistent", one_launch)
    x0, x, v, vd = _state(eng, model, n)
    fx, st, sched = eng.comp_fcn(x, record=True)
    sigma = 1.0e-4 * np.sqrt(eng.dot(x, x))[0]
    xp = eng.axpby(1.0, x, sigma, vd)
    _, st_def = eng.comp_fcn_frozen(xp, sched)
    assert eng.counter("frozen_persistent_years") == one_launch
    assert abs(st_def["nerr_checked"] - len(sched) / 128.0) <= 2 and 0.0 < st_def["max_err"] <= 1.05
    eng.set_option("frozen_err_check", 1)
    fx_all, st_all = eng.comp_fcn_frozen(x, sched)
    assert np.array_equal(eng.download(fx_all), eng.download(fx))          # checked or not, the recorded year again
    # every step whose solves take at most two sweeps (one launch: one sweep) is checked; the estimates are the recorded ones
    assert st_all["nerr_checked"] > (0.5 if one_launch else 0.8) * len(sched)
    assert abs(st_all["max_err"] - sched[:, 6].max()) < 0.05
    _, st_p = eng.comp_fcn_frozen(xp, sched)
    assert st_p["max_err"] <= 1.05
    eng.set_option("frozen_err_check", 0)
    _, st_off = eng.comp_fcn_frozen(xp, sched)
    assert st_off["nerr_checked"] == 0
    if not one_launch:
        assert st_off["nlaunch"] < st_all["nlaunch"]
    # a different state altogether (3 x the tracer, with structure): the recorded steps do not control its error
    eng.set_option("frozen_err_check", 8)
    rng = np.random.default_rng(5)
    far = eng.upload(3.0 * x0 * (1.0 + 0.3 * rng.standard_normal(x0.shape)))
    with pytest.raises(Nk2dFrozenMismatch):
        eng.comp_fcn_frozen(far, sched)
    eng.close()


def test_side_files_belong_to_their_values(tmp_path):
    from nk_ooc_amd import ncio
    from nk_ooc_amd.model_config import ModelConfig
    from nk_ooc_amd.model_state import ModelState, sched_path
    from nk_ooc_amd.setup_solver import gen_grid_vars_file, make_config

    cfg = make_config(str(tmp_path), 22, 9)
    gen_grid_vars_file(cfg["modelinfo"])
    ModelState.reset_class()
    ModelState.write_files = True
    ModelState.model_config_obj = ModelConfig(cfg["modelinfo"])
    try:
        iterate = ModelState("gen_init_iterate")
        fname = os.path.join(str(tmp_path), "fcn_00.nc")
        fcn = iterate.comp_fcn(fname, None)
        assert os.path.exists(sched_path(fname)) and fcn._sched["iage"].shape[1] == 8
        # a resumed run: nothing in memory, the side file is found and belongs to the values in the file
        ModelState._resident.clear()
        ModelState._sched_by_name.clear()
        again = ModelState(fname)
        assert np.array_equal(again._sched["iage"], fcn._sched["iage"])
        # the file rewritten by hand with other values: the side file no longer belongs to them
        data, _ = ncio.read_file(fname, ["iage", "iage_slow_rest"])
        other = iterate * 2.0
        other.dump(os.path.join(str(tmp_path), "other.nc"), "test")
        os.replace(os.path.join(str(tmp_path), "other.nc"), fname)
        ModelState._resident.clear()
        ModelState._sched_by_name.clear()
        assert os.path.exists(sched_path(fname))
        assert ModelState(fname)._sched is None
        # written again through dump(): side file and remembered schedule are gone; so they are after a frozen year
        fcn2 = iterate.comp_fcn(fname, None)
        assert ModelState(fname)._sched is not None
        iterate.dump(fname, "test")
        assert not os.path.exists(sched_path(fname)) and ModelState(fname)._sched is None
        fcn3 = iterate.comp_fcn(fname, None)
        iterate.comp_fcn(fname, None, frozen=fcn3._sched)
        assert not os.path.exists(sched_path(fname)) and ModelState(fname)._sched is None
        assert fcn2 is not None
    finally:
        ModelState.reset_class()
