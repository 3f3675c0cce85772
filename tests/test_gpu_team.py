"""Column teams and pairs (`k_newton_team`, `k_newton_pair`, option "team") are launch-shape choices of the
Newton-iteration kernel of the launch-per-phase path: every value is computed by the same operations in the same order
as with one wave per column, so a forward year must come out BIT-IDENTICAL -- results, step schedule and counters --
for every module kind (iage; phosphorus, whose waves read the other tracers of their column; a forced module with
forcing files) and for grids that do not fill the last workgroup or the eight XCDs evenly."""
import numpy as np
import pytest

from helpers import oracle_iage

pytestmark = pytest.mark.gpu


def make_engine(nz, ny, vv=0.1, kh=1000.0, **kw):
    from nk_ooc_amd.engine import iage_engine
    from nk_ooc_amd.grid import Grid2d

    return iage_engine(Grid2d.default(nz, ny, vv, kh), **kw)


def _year_variants(eng, x, counters=("nfev", "njev", "nlu", "nsteps", "nrejected", "nnewton", "nsweeps")):
    eng.set_option("stream_years", 0)       # (the launch shapes are those of the years that are launched)
    ref = None
    for team, xcd in ((0, 0), (1, 0), (2, 0)):       # 2: a pair of waves per column
        eng.set_option("team", team)
        fx, stats, sched = eng.comp_fcn(x, record=True)
        got = (eng.download(fx), sched, {k: stats[k] for k in counters})
        if ref is None:
            ref = got
            continue
        assert np.array_equal(got[0], ref[0]), (team, xcd)
        assert np.array_equal(got[1], ref[1]), (team, xcd)
        assert got[2] == ref[2], (team, xcd)
    eng.set_option("team", -1)
    return ref


@pytest.mark.parametrize("nz,ny", [(26, 26), (70, 13), (130, 21)])
def test_team_iage_bitwise(nz, ny):
    eng = make_engine(nz, ny)
    model, _ = oracle_iage(nz, ny)
    rng = np.random.default_rng(5)
    col = np.interp(model.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x = eng.upload(np.stack([np.broadcast_to(col[:, None], (nz, ny))] * 2) + 0.01 * rng.standard_normal((2, nz, ny)))
    res, _, counters = _year_variants(eng, x)
    assert np.isfinite(res).all() and counters["nsteps"] > 100


def test_team_min_sweeps_two_and_replay():
    """the launches of a two-sweep iteration (stage + first sweep; second sweep in delta form + update) and of a
    step-replayed year (lin_tol 1e-3: up to several sweeps, right-hand sides written and read back)"""
    eng = make_engine(26, 26)
    model, _ = oracle_iage(26, 26)
    col = np.interp(model.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x = eng.upload(np.stack([np.broadcast_to(col[:, None], (26, 26))] * 2).copy())
    eng.set_option("stream_years", 0)       # (by launches)
    eng.set_option("min_sweeps", 2)
    out = []
    for team in (0, 1):
        eng.set_option("team", team)
        fx, stats, sched = eng.comp_fcn(x, record=True)
        out.append((eng.download(fx), sched, stats["nsweeps"]))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
    eng.set_option("min_sweeps", 1)
    eng.set_option("lin_tol", 1.0e-3)
    rep = []
    for team in (0, 1):
        eng.set_option("team", team)
        fx, _, _ = eng.comp_fcn(x, replay=out[0][1])
        rep.append(eng.download(fx))
    assert np.array_equal(rep[0], rep[1])
    with pytest.raises(Exception, match="team"):
        eng.set_option("team", 3)


def test_team_phosphorus_bitwise():
    from nk_ooc_amd.engine import phosphorus_engine
    from nk_ooc_amd.grid import Grid2d

    grid = Grid2d.default(30, 12)
    eng = phosphorus_engine(grid)
    rng = np.random.default_rng(2)
    x0 = np.empty((3, 30, 12))
    x0[0] = 2.0 + 0.1 * rng.standard_normal((30, 12))
    x0[1] = 0.05 + 0.005 * rng.standard_normal((30, 12))
    x0[2] = 0.01 + 0.001 * rng.standard_normal((30, 12))
    res, _, counters = _year_variants(eng, eng.upload(x0))
    assert np.isfinite(res).all() and counters["nsteps"] > 50
    eng.close()


@pytest.mark.parametrize("nz", [130, 416])
def test_team_forced_file_bitwise(nz):
    """kind 2 (forcing records interpolated in time, thresholded sink) at 3 and 7 levels per lane, a short year"""
    from nk_ooc_amd.engine import ModuleEngine
    from nk_ooc_amd.grid import Grid2d

    ny = 11
    rng = np.random.default_rng(nz)
    times = np.array([-10.0, 40.0, 95.0, 200.0, 300.0]) * 86400.0
    restore = 1.0 + 0.2 * rng.standard_normal((5, ny))
    sms = 3.0e-8 * rng.standard_normal((5, nz, ny))
    eng = ModuleEngine(Grid2d.default(nz, ny), tc=1, surf_rate=(24.0 / 86400.0,), module_kind=2,
                       restore_series=(times, restore), sms_series=(times, sms), sink_thres=0.4,
                       time_range=(0.0, 30.0 * 86400.0))
    x = eng.upload(0.6 + 0.2 * rng.standard_normal((1, nz, ny)))
    res, _, counters = _year_variants(eng, x)
    assert np.isfinite(res).all() and counters["nsteps"] > 10
    eng.close()


def test_profile_replay_and_shapes():
    """measurement plumbing of bench.py: launch-shape tallies of a year and the back-to-back replay of each shape;
    the replay writes to scratch only -- the next year is bit-identical to the one before it"""
    eng = make_engine(26, 26)
    model, _ = oracle_iage(26, 26)
    col = np.interp(model.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x = eng.upload(np.stack([np.broadcast_to(col[:, None], (26, 26))] * 2).copy())
    eng.set_option("stream_years", 0)       # (by launches)
    with pytest.raises(Exception, match="forward year"):
        eng.profile_replay(0, 4)
    eng.profile_reset(0)
    fx, stats, _ = eng.comp_fcn(x)
    first = eng.download(fx)
    shapes = eng.profile_shapes()
    totals = eng.profile_totals()
    assert sum(shapes["counts"]) + stats["nlu"] // 2 >= totals["launches"] - stats["nlu"]   # the rest factorise
    assert sum(shapes["counts"]) > 0 and all(b >= 0.0 for b in shapes["bytes"])
    for shape in range(3):
        rep = eng.profile_replay(shape, 20)
        assert 0.5 < rep["avg_us"] < 500.0 and rep["bytes"] > 0.0
    with pytest.raises(Exception, match="shape"):
        eng.profile_replay(3, 4)
    fx2, _, _ = eng.comp_fcn(x)
    assert np.array_equal(eng.download(fx2), first)


@pytest.mark.parametrize("nz,ny", [(26, 26), (130, 21), (416, 9)])
def test_pair_frozen_year_bitwise(nz, ny):
    """pairs also end frozen steps (k_newton_pair with the FinalArgs epilogue on the stage wave): a year frozen on the
    recorded steps is the recorded year whichever launch shape runs it"""
    eng = make_engine(nz, ny)
    model, _ = oracle_iage(nz, ny)
    rng = np.random.default_rng(6)
    col = np.interp(model.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x = eng.upload(np.stack([np.broadcast_to(col[:, None], (nz, ny))] * 2) + 0.01 * rng.standard_normal((2, nz, ny)))
    eng.set_option("stream_years", 0)       # (the year by launches: its launch count is compared below)
    fx, st, sched = eng.comp_fcn(x, record=True)
    want = eng.download(fx)
    for team in (0, 1, 2):
        eng.set_option("team", team)
        fx2, st2 = eng.comp_fcn_frozen(x, sched)
        assert np.array_equal(eng.download(fx2), want), team
        assert st2["nlaunch"] < 0.6 * st["nlaunch"]
    eng.close()
