"""The frozen year of a small grid as ONE launch on a schedule cache (k_frozen_persistent): the same
device functions as the launch-per-phase path, so the same bits -- for the recorded state (the recorded year again) and for
a perturbed one; the cache is built once per schedule; modules it is not made for, larger grids and years that do not pass
their check take the launch-per-phase path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _iage(n, ny=None):
    from nk_ooc_amd.engine import iage_engine
    from nk_ooc_amd.grid import Grid2d

    eng = iage_engine(Grid2d.default(n, ny or n))
    eng.set_option("device_ctl", 0)
    eng.set_option("frozen_alloc_async", 0)     # (a cache above 8 GB -- 104 x 104 -- is otherwise allocated by a thread of its own,
    return eng                                  #  the years of the meantime running launch by launch)


def _state(eng, seed=3):
    rng = np.random.default_rng(seed)
    tc, nz, ny = eng.shape
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x0 = np.stack([np.broadcast_to(col[:, None], (nz, ny))] * tc) + 0.01 * rng.standard_normal(eng.shape)
    return x0, eng.upload(x0), eng.upload(x0 * (1.0 + 1.0e-4 * rng.standard_normal(x0.shape)))


@pytest.mark.parametrize("n", [26, 52, 104])
@pytest.mark.parametrize("mode", ["default", "scipy_decisions"])
def test_one_launch_year_is_the_launch_per_phase_year(n, mode):
    eng = _iage(n)
    if mode == "scipy_decisions":
        # SciPy's Jacobian reuse: rows that keep the Jacobian (and the factorisation) of an earlier step, Jacobian times at
        # step starts -- the cache rebuilds both per row from (t_jac, h_lu)
        eng.set_option("jac_stage", -1)
        eng.set_option("jac_fresh", 0)
    _, x, xp = _state(eng)
    fx, st, sched = eng.comp_fcn(x, record=True)
    want = eng.download(fx)
    eng.set_option("frozen_persistent", 0)
    fx_l, st_l = eng.comp_fcn_frozen(x, sched)
    fxp_l, _ = eng.comp_fcn_frozen(xp, sched)
    assert np.array_equal(eng.download(fx_l), want) and eng.counter("frozen_persistent_years") == 0
    eng.set_option("frozen_persistent", 1)
    fx_p, st_p = eng.comp_fcn_frozen(x, sched)
    assert eng.counter("frozen_persistent_years") == 1 and eng.counter("frozen_cache_builds") == 1
    assert np.array_equal(eng.download(fx_p), want)                          # the recorded year, bit for bit
    fxp_p, st_pp = eng.comp_fcn_frozen(xp, sched)
    assert np.array_equal(eng.download(fxp_p), eng.download(fxp_l))          # and the launch-per-phase year of another state
    assert eng.counter("frozen_persistent_years") == 2 and eng.counter("frozen_cache_builds") == 1   # one cache per schedule
    # which flavour ran: a four-wave team per column (option "frozen_team", the default up to two levels per lane)
    assert eng.counter("frozen_team_years") == 2
    for key in ("nsteps", "nnewton"):
        assert st_p[key] == st_l[key], key
    # (both evaluate the error estimate of every 128th step -- a tendency and a solve each; the one-launch year leaves out
    # the first step and steps with two-sweep solves)
    assert 0 < st_p["nerr_checked"] <= st_l["nerr_checked"] and st_p["max_err"] > 0.0
    assert st_p["nlaunch"] < 20 < st_l["nlaunch"]
    assert st_pp["seconds"] < st_l["seconds"]
    # a new schedule (another state's year): a new cache
    fx2, _, sched2 = eng.comp_fcn(xp, record=True)
    fx2_p, _ = eng.comp_fcn_frozen(xp, sched2)
    assert np.array_equal(eng.download(fx2_p), eng.download(fx2)) and eng.counter("frozen_cache_builds") == 2
    eng.close()


def test_products_and_gmres_run_on_it():
    """nk2d_jvp / nk2d_gmres_solve with a schedule installed: the same numbers with and without the one-launch year"""
    n = 26
    eng = _iage(n)
    eng.set_region(np.ones((n, n), dtype=np.int32), np.outer(eng.grid.depth.delta, eng.grid.ypos.delta))
    _, x, _ = _state(eng)
    fx, _, sched = eng.comp_fcn(x, record=True)
    out = {}
    for flag in (0, 1):
        eng.set_option("frozen_persistent", flag)
        inc, info = eng.gmres_solve(x, fx, 0.0, 0, 3, sched=sched)
        out[flag] = (eng.download(inc), info["h_mat"].copy(), info["beta"].copy())
    assert eng.counter("frozen_persistent_years") == 3
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    eng.close()


def test_forced_modules_and_column_grids():
    """the decay variant of forced (one tracer), a forced module with forcing files (its planes carry the forcing bundle), and
    a grid without lateral processes (20 x 3 columns)"""
    from nk_ooc_amd.engine import ModuleEngine, forced_engine
    from nk_ooc_amd.grid import Grid2d

    rng = np.random.default_rng(8)
    cases = []
    eng = forced_engine(Grid2d.default(22, 9), {"forced_surf_restore_opt": "none", "forced_sms_opt": "decay",
                                                 "forced_sms_decay_rate": "1.0e-8"})
    cases.append((eng, 1.0 + 0.2 * rng.standard_normal((1, 22, 9))))
    nz, ny = 100, 11
    times = np.array([-10.0, 40.0, 95.0, 200.0, 300.0]) * 86400.0
    eng = ModuleEngine(Grid2d.default(nz, ny), tc=1, surf_rate=(24.0 / 86400.0,), module_kind=2,
                       restore_series=(times, 1.0 + 0.2 * rng.standard_normal((5, ny))),
                       sms_series=(times, 3.0e-8 * rng.standard_normal((5, nz, ny))), time_range=(0.0, 40.0 * 86400.0))
    cases.append((eng, 0.6 + 0.2 * rng.standard_normal((1, nz, ny))))
    from nk_ooc_amd.engine import iage_engine

    eng = iage_engine(Grid2d.default(20, 3, 0.0, 0.0))
    cases.append((eng, 1.0 + 0.1 * rng.standard_normal((2, 20, 3))))
    for eng, x0 in cases:
        eng.set_option("device_ctl", 0)
        x = eng.upload(x0)
        fx, st, sched = eng.comp_fcn(x, record=True)
        fx_p, st_p = eng.comp_fcn_frozen(x, sched)
        assert eng.counter("frozen_persistent_years") == 1, eng.shape
        assert np.array_equal(eng.download(fx_p), eng.download(fx))
        xp = eng.upload(x0 * (1.0 + 1.0e-5 * rng.standard_normal(x0.shape)))
        fx_pp, _ = eng.comp_fcn_frozen(xp, sched)
        eng.set_option("frozen_persistent", 0)
        fx_lp, _ = eng.comp_fcn_frozen(xp, sched)
        assert np.array_equal(eng.download(fx_pp), eng.download(fx_lp))
        eng.close()


@pytest.mark.parametrize("case", ["iage_26", "iage_52_two_sweeps", "forced_decay_22x9"])
def test_team_and_wave_per_column_flavours_agree(case):
    """the four-wave team inside the one-launch year (option "frozen_team" 1, the default up to two levels per lane) against the wave
    per column, with and without the validation fences, each against the launch-per-phase year: the recorded and a perturbed state,
    bit for bit -- also with two-sweep solves (inner tolerance 1e-3), whose second launch of an iteration has no stage part"""
    from nk_ooc_amd.engine import forced_engine
    from nk_ooc_amd.grid import Grid2d

    rng = np.random.default_rng(11)
    if case == "forced_decay_22x9":
        eng = forced_engine(Grid2d.default(22, 9), {"forced_surf_restore_opt": "none", "forced_sms_opt": "decay",
                                                     "forced_sms_decay_rate": "1.0e-8"})
        x0 = 1.0 + 0.2 * rng.standard_normal((1, 22, 9))
    else:
        n = 26 if case == "iage_26" else 52
        eng = _iage(n)
        if case == "iage_52_two_sweeps":
            eng.set_option("lin_tol", 1.0e-3)
        x0, _, _ = _state(eng)
    eng.set_option("device_ctl", 0)
    x = eng.upload(x0)
    xp = eng.upload(x0 * (1.0 + 1.0e-4 * rng.standard_normal(x0.shape)))
    fx, st, sched = eng.comp_fcn(x, record=True)
    if case == "iage_52_two_sweeps":
        assert st["nsweeps"] > 1.5 * st["nnewton"]
    eng.set_option("frozen_persistent", 0)
    want = [eng.download(eng.comp_fcn_frozen(v, sched)[0]) for v in (x, xp)]
    assert np.array_equal(want[0], eng.download(fx))
    eng.set_option("frozen_persistent", 1)
    # (team or a wave per column, columns per workgroup of the latter, release / acquire fences around every hand-over)
    for team, wpb, fences, years in ((1, 2, 0, 2), (0, 4, 0, 2), (0, 1, 0, 2), (1, 2, 1, 4), (0, 2, 1, 4)):
        eng.set_option("frozen_team", team)
        eng.set_option("frozen_wpb", wpb)
        eng.set_option("year_fences", fences)
        got = [eng.download(eng.comp_fcn_frozen(v, sched)[0]) for v in (x, xp)]
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (case, team, wpb, fences)
        assert eng.counter("frozen_team_years") == years
    assert eng.counter("frozen_persistent_years") == 10 and eng.frozen_fallbacks() == 0
    eng.close()


@pytest.mark.parametrize("nz", [250, 320, 384, 512])
def test_every_levels_per_lane_instantiation_of_the_one_launch_year(nz):
    """four, five, six and eight levels per lane (a wave per column, neighbour hand-over) on a narrow grid:
    the one-launch year against the launch-per-phase year, recorded and perturbed state, bit for bit -- and faster"""
    ny = 48
    eng = _iage(nz, ny)
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x0 = np.stack([np.broadcast_to(col[:, None], (nz, ny))] * 2).copy()
    x = eng.upload(x0)
    xp = eng.upload(x0 * (1.0 + 1.0e-4 * np.outer(np.sin(3.0 * np.linspace(0.0, 1.0, nz)), np.cos(2.0 * np.linspace(0.0, 1.0, ny)))[None]))
    fx, _, sched = eng.comp_fcn(x, record=True)
    eng.set_option("frozen_persistent", 0)
    want = [eng.download(eng.comp_fcn_frozen(v, sched)[0]) for v in (x, xp)]
    _, st_l = eng.comp_fcn_frozen(xp, sched)
    eng.set_option("frozen_persistent", 1)
    got = [eng.download(eng.comp_fcn_frozen(v, sched)[0]) for v in (x, xp)]
    _, st_p = eng.comp_fcn_frozen(xp, sched)
    assert eng.counter("frozen_persistent_years") == 3
    assert np.array_equal(want[0], eng.download(fx)) and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert st_p["nerr_checked"] > 0 and st_p["seconds"] < st_l["seconds"]
    # adjacent columns of one tracer to a workgroup, or one ypos column with all its tracers (and the step's constants shared in LDS)
    for by_col in (0, 2):
        eng.set_option("frozen_by_column", by_col)
        assert np.array_equal(eng.download(eng.comp_fcn_frozen(xp, sched)[0]), want[1]), by_col
    eng.close()


def test_full_size_year_in_one_launch():
    """416 x 416 (seven levels per lane, a wave per column, workgroups hand over to their neighbours): the one-launch year on
    the schedule cache -- 2 600 steps' planes and factorisations, 100 GB of the 288 -- against the launch-per-phase year: bit for
    bit for the recorded and for a perturbed state, the launch itself faster than the year of launches, and a second schedule
    rebuilds the cache in place"""
    n = 416
    eng = _iage(n)
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x0 = np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy()
    x = eng.upload(x0)
    # a smooth perturbation (white noise is not a state the recorded steps control: the sampled error estimates object)
    zz = np.linspace(0.0, 1.0, n)
    xp = eng.upload(x0 * (1.0 + 1.0e-4 * np.outer(np.sin(3.0 * zz), np.cos(2.0 * zz))[None]))
    fx, st, sched = eng.comp_fcn(x, record=True)
    eng.set_option("frozen_persistent", 0)
    want = [eng.download(eng.comp_fcn_frozen(v, sched)[0]) for v in (x, xp)]
    _, st_l = eng.comp_fcn_frozen(xp, sched)
    eng.set_option("frozen_persistent", 1)
    got = [eng.download(eng.comp_fcn_frozen(v, sched)[0]) for v in (x, xp)]
    _, st_p = eng.comp_fcn_frozen(xp, sched)
    assert eng.counter("frozen_persistent_years") == 3 and eng.counter("frozen_cache_builds") == 1
    assert np.array_equal(want[0], eng.download(fx)) and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert st_p["nlaunch"] < 20 < st_l["nlaunch"] and st_p["seconds"] < 0.9 * st_l["seconds"]
    print(f"416^2 frozen year: {1e3 * st_l['seconds']:.1f} ms launch by launch, {1e3 * st_p['seconds']:.1f} ms in one launch")
    # adjacent columns of one tracer to a workgroup, four or one of them (option "frozen_wpb"): the same bits
    eng.set_option("frozen_by_column", 0)
    for wpb in (4, 1):
        eng.set_option("frozen_wpb", wpb)
        assert np.array_equal(eng.download(eng.comp_fcn_frozen(xp, sched)[0]), want[1]), wpb
    eng.set_option("frozen_wpb", 2)
    eng.set_option("frozen_by_column", 1)
    assert eng.counter("frozen_persistent_years") == 5
    # what lives in LDS for the year (bits: coefficients, W, the step's mixing columns and Jacobian diagonals, pivots) and which
    # columns share a workgroup: the same bits whatever the choice
    for by_col, lds in ((0, 0), (0, 1), (0, 3), (2, 3), (2, 7), (2, 15)):
        eng.set_option("frozen_by_column", by_col)
        eng.set_option("frozen_coef_lds", lds)
        assert np.array_equal(eng.download(eng.comp_fcn_frozen(xp, sched)[0]), want[1]), (by_col, lds)
    eng.set_option("frozen_by_column", 1)
    eng.set_option("frozen_coef_lds", 15)
    assert eng.counter("frozen_persistent_years") == 11
    fx2, _, sched2 = eng.comp_fcn(xp, record=True)
    fx2_p, _ = eng.comp_fcn_frozen(xp, sched2)
    assert np.array_equal(eng.download(fx2_p), eng.download(fx2)) and eng.counter("frozen_cache_builds") == 2
    # option "frozen_cache_after": that many years of a schedule launch by launch before its cache is built
    eng.set_option("frozen_cache_after", 3)
    fx3, _, sched3 = eng.comp_fcn(x, record=True)       # (the first schedule again: its cache was replaced)
    years = eng.counter("frozen_persistent_years")
    for k in range(4):
        fx3_p, _ = eng.comp_fcn_frozen(x, sched3)
        assert np.array_equal(eng.download(fx3_p), eng.download(fx3))
        assert eng.counter("frozen_persistent_years") == years + (1 if k == 3 else 0)
    assert eng.counter("frozen_cache_builds") == 3
    eng.close()
    # the default for a cache of this size: its 120 GB are allocated by a thread of the library's own (hipMalloc of that size
    # has been seen to take from 0.03 to 3 s), the years of the meantime run launch by launch -- the same bits either way
    from nk_ooc_amd.engine import iage_engine
    from nk_ooc_amd.grid import Grid2d

    eng = iage_engine(Grid2d.default(n, n))
    eng.set_option("device_ctl", 0)
    x = eng.upload(x0)
    fx, _, sched = eng.comp_fcn(x, record=True)
    first = None
    for k in range(150):        # (up to half a minute of years launch by launch; seen: one)
        fx_k, st_k = eng.comp_fcn_frozen(x, sched)
        assert np.array_equal(eng.download(fx_k), eng.download(fx))
        if st_k["nlaunch"] < 20:
            first = k
            break
    assert first is not None and first >= 1 and eng.counter("frozen_cache_builds") == 1
    print(f"416^2: the cache was there for year {first + 1} of the schedule")
    eng.close()


def test_what_it_is_not_for_takes_the_other_path():
    from nk_ooc_amd.engine import phosphorus_engine
    from nk_ooc_amd.grid import Grid2d

    # a Jacobian that reads the state
    ph = phosphorus_engine(Grid2d.default(30, 12))
    rng = np.random.default_rng(8)
    x0 = np.stack([2.0 + 0.1 * rng.standard_normal((30, 12)), 0.05 + 0.005 * rng.standard_normal((30, 12)),
                   0.01 + 0.001 * rng.standard_normal((30, 12))])
    x = ph.upload(x0)
    fx, _, sched = ph.comp_fcn(x, record=True)
    fx2, _ = ph.comp_fcn_frozen(x, sched)
    assert np.array_equal(ph.download(fx2), ph.download(fx)) and ph.counter("frozen_persistent_years") == 0
    ph.close()
    # more levels per lane than the option admits; a cache larger than allowed
    eng = _iage(130, 9)
    _, x, _ = _state(eng)
    fx, _, sched = eng.comp_fcn(x, record=True)
    eng.set_option("frozen_persistent_max_e", 2)
    eng.comp_fcn_frozen(x, sched)
    assert eng.counter("frozen_persistent_years") == 0
    eng.set_option("frozen_persistent_max_e", 3)
    fx3, _ = eng.comp_fcn_frozen(x, sched)
    assert eng.counter("frozen_persistent_years") == 1 and np.array_equal(eng.download(fx3), eng.download(fx))
    eng.set_option("frozen_cache_gb", 1.0e-3)
    eng.comp_fcn_frozen(x, sched)
    assert eng.counter("frozen_persistent_years") == 1
    eng.close()
    # a year that does not pass its check goes back to the launch-per-phase path, which resumes it from a checkpoint
    eng = _iage(26)
    _, x, _ = _state(eng)
    fx, _, sched = eng.comp_fcn(x, record=True)
    bad = sched.copy()
    k = next(i for i in range(200, len(bad)) if bad[i, 3] >= 4)
    bad[k, 3] -= 3
    fx4, st4 = eng.comp_fcn_frozen(x, bad)
    assert eng.counter("frozen_persistent_years") == 0 and st4["nresumed"] >= 1 and eng.frozen_fallbacks() == 0
    assert np.allclose(eng.download(fx4), eng.download(fx), rtol=0.0, atol=1e-6 * np.max(np.abs(eng.download(fx))))
    # and a barrier that times out hands the year over as well
    eng.set_option("barrier_timeout_ms", 0.0)
    fx5, st5 = eng.comp_fcn_frozen(x, sched)
    assert st5["nbarrier_timeouts"] >= 1 and np.array_equal(eng.download(fx5), eng.download(fx))
    eng.close()
