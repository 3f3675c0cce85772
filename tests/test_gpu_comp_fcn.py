"""GPU parity of the forward year (comp_fcn) against the oracle.

Two modes ("parity reality check"):
* step replay -- the HIP integrator consumes the accepted-step schedule recorded by
  the oracle (which reproduces SciPy's Radau bit for bit); the map is then smooth and
  the result must agree to 1e-10 relative;
* free running -- the HIP controller takes its own decisions; the reference's own
  reproducibility floor is ~1e-6 (a 1e-15 input perturbation already flips step
  decisions), so the comparison uses the reference CI tolerance atol 1e-6 / rtol 1e-3
  (scripts/ci_py_driver_2d_iage.sh:38).
"""
import numpy as np
import pytest

from helpers import free_years, oracle_iage, rel_err

pytestmark = pytest.mark.gpu


def make_engine(nz, ny, vv=0.1, kh=1000.0, **kw):
    from nk_ooc_amd.engine import iage_engine
    from nk_ooc_amd.grid import Grid2d

    return iage_engine(Grid2d.default(nz, ny, vv, kh), **kw)


CASES = [("20x3_columns", 20, 3, 0.0, 0.0), ("26x26", 26, 26, 0.1, 1000.0),
         ("26x26_bumpy", 26, 26, 0.1, 1000.0)]


@pytest.mark.parametrize("tag,nz,ny,vv,kh", CASES)
def test_comp_fcn_replay(golden_dir, tag, nz, ny, vv, kh):
    from oracle import radau

    g = np.load(f"{golden_dir}/comp_fcn_{tag}.npz")
    _, tm = oracle_iage(nz, ny, vv, kh)
    want, solver = radau.comp_fcn(tm, g["y0"], return_solver=True)
    assert np.array_equal(want, g["fcn"])  # the oracle reproduces the reference run
    sched = np.array(solver.schedule, dtype=np.float64)
    eng = make_engine(nz, ny, vv, kh)
    fx, stats, _ = eng.comp_fcn(eng.upload(g["y0"]), replay=sched)
    got = eng.download(fx).reshape(-1)
    assert stats["nsteps"] == len(sched)
    assert rel_err(got, want) < 1e-10, rel_err(got, want)


@pytest.mark.parametrize("tag,nz,ny,vv,kh", CASES)
def test_comp_fcn_free(golden_dir, tag, nz, ny, vv, kh):
    g = np.load(f"{golden_dir}/comp_fcn_{tag}.npz")
    eng = make_engine(nz, ny, vv, kh)
    (fx, stats, sched), (fx_def, stats_def, _) = free_years(eng, eng.upload(g["y0"]), record=True)
    got = eng.download(fx).reshape(-1)
    # reference CI tolerance for fcn files, in both controller modes
    assert np.allclose(got, g["fcn"], rtol=1.0e-3, atol=1.0e-6), np.max(np.abs(got - g["fcn"]))
    assert np.allclose(eng.download(fx_def).reshape(-1), g["fcn"], rtol=1.0e-3, atol=1.0e-6)
    assert len(sched) == stats["nsteps"]
    # with SciPy's Jacobian reuse the controller takes the same kind of path as SciPy's: counters within 10 %;
    # the default mode needs no more tendency evaluations than that
    for key in ("nfev", "njev", "nlu"):
        assert abs(stats[key] - int(g[key])) <= 0.1 * int(g[key]) + 5, (key, stats[key], int(g[key]))
    assert stats_def["nfev"] <= 1.05 * stats["nfev"] and stats_def["njev"] >= stats_def["nsteps"]
    # a recorded schedule replays to the same answer (smooth map).  The replay solves the stage systems to 1e-3
    # (include/nk2d.h); the recorded year uses that inner tolerance too, so that both are the same arithmetic -- with
    # the default 3e-2 short steps take single-sweep solves and the two agree to SciPy's Newton tolerance only (1e-9)
    from nk_ooc_amd.engine import DEFAULT_LIN_TOL

    eng.set_option("lin_tol", 1.0e-3)
    try:
        fx1, _, sched1 = eng.comp_fcn(eng.upload(g["y0"]), record=True)
    finally:
        eng.set_option("lin_tol", DEFAULT_LIN_TOL)
    fx2, _, _ = eng.comp_fcn(eng.upload(g["y0"]), replay=sched1)
    assert rel_err(eng.download(fx2), eng.download(fx1)) < 1e-10
    fx3, _, _ = eng.comp_fcn(eng.upload(g["y0"]), replay=sched)
    assert rel_err(eng.download(fx3), eng.download(fx)) < 1e-7


def test_controller_variants_take_identical_decisions():
    """the host's controller driving launches and the same controller driving ONE resident kernel through a command stream
    (csrc/nk2d_stream.h) follow the same path bit for bit -- here under SciPy's decisions with at least two sweeps per
    solve (the round-1 rule); options that change what is integrated (single-launch iterations, Jacobian at every step
    start, RADAU5's growth rule) change the step sequence, not the ODE or its tolerances"""
    import numpy as np

    eng = make_engine(26, 26)
    rng = np.random.default_rng(11)
    model, _ = oracle_iage(26, 26)
    col = np.interp(model.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x = eng.upload(np.stack([np.broadcast_to(col[:, None], (26, 26))] * 2) + 0.01 * rng.standard_normal((2, 26, 26)))
    results = []
    eng.set_option("jac_fresh", 0)
    eng.set_option("growth_cap", 0)
    eng.set_option("jac_stage", -1)
    eng.set_option("min_sweeps", 2)
    for stream in (0, 1):
        eng.set_option("stream_years", stream)
        fx, stats, sched = eng.comp_fcn(x, record=True)
        results.append((eng.download(fx), stats, sched))
    assert eng.counter("stream_years_run") == 1
    for res, stats, sched in results[1:]:
        assert np.array_equal(res, results[0][0])
        assert np.array_equal(sched, results[0][2])
        for key in ("nfev", "njev", "nlu", "nsteps", "nrejected", "nnewton"):
            assert stats[key] == results[0][1][key], key
    # single-launch iterations for one-sweep solves (the default) change the inner accuracy of short steps, not the ODE
    eng.set_option("min_sweeps", 1)
    fx_s, stats_s, _ = eng.comp_fcn(x)
    assert np.allclose(eng.download(fx_s), results[0][0], rtol=1e-3, atol=1e-6)
    assert stats_s["nsweeps"] < 0.8 * results[0][1]["nsweeps"]
    assert abs(stats_s["nnewton"] - results[0][1]["nnewton"]) <= 0.05 * results[0][1]["nnewton"]
    # jac_fresh (the engine's default) leaves SciPy's decision sequence but solves the same ODE to the same tolerance
    eng.set_option("jac_fresh", 1)
    fx, stats, _ = eng.comp_fcn(x)
    assert np.allclose(eng.download(fx), results[0][0], rtol=1e-3, atol=1e-6)
    assert stats["njev"] == stats["nsteps"] + 1 or stats["njev"] >= stats["nsteps"]
    # growth_cap 1.0 (RADAU5's rule: no step growth after a Newton failure) changes the step sequence, not the ODE
    # or its tolerances: fewer Newton iterations, the same year to the integrator's accuracy
    eng.set_option("growth_cap", 1.0)
    fx_cap, stats_cap, _ = eng.comp_fcn(x)
    eng.set_option("growth_cap", 0.0)
    assert np.allclose(eng.download(fx_cap), results[0][0], rtol=1e-3, atol=1e-5)
    assert stats_cap["nnewton"] < stats["nnewton"]
    with pytest.raises(Exception, match="growth_cap"):
        eng.set_option("growth_cap", -1.0)


def test_factor_storage_precision_does_not_move_the_result(golden_dir):
    """the line factorisation is an approximate inverse inside Newton iterations that re-evaluate
    the exact residual: reading it from its single precision copy (option) or from the double
    precision arrays (default) gives the same step-replayed forward year to 1e-11, and both match the oracle"""
    from oracle import radau

    g = np.load(f"{golden_dir}/comp_fcn_26x26_bumpy.npz")
    _, tm = oracle_iage(26, 26, 0.1, 1000.0)
    want, solver = radau.comp_fcn(tm, g["y0"], return_solver=True)
    sched = np.array(solver.schedule, dtype=np.float64)
    got = {}
    for flag in (0.0, 1.0):
        eng = make_engine(26, 26, 0.1, 1000.0)
        eng.set_option("factor_fp32", flag)
        fx, _, _ = eng.comp_fcn(eng.upload(g["y0"]), replay=sched)
        got[flag] = eng.download(fx).reshape(-1)
        assert rel_err(got[flag], want) < 1e-10
    assert rel_err(got[1.0], got[0.0]) < 1e-11


@pytest.mark.parametrize("vv,kh", [(3.0, 3.0e6), (0.0, 1.0e7)])
def test_comp_fcn_strong_lateral_coupling(vv, kh):
    """lateral advection / mixing far above the defaults (per grid spacing: several times the coupling of the
    416 x 416 case, on 20 columns): the line relaxation then needs
    many sweeps per solve (contraction bound close to 1) -- replayed year to 1e-10, free-running year at the
    reference CI tolerance, counters within 10 % of the oracle's (SciPy's) own"""
    from oracle import radau

    nz, ny = 24, 20
    _, tm = oracle_iage(nz, ny, vv, kh)
    col = np.interp(tm.model.depth.mid, [55.0, 200.0], [0.0, 2.0])
    y0 = (np.stack([np.broadcast_to(col[:, None], (nz, ny))] * 2)
          * (1.0 + 0.1 * np.random.default_rng(4).standard_normal((2, nz, ny)))).reshape(-1)
    want, solver = radau.comp_fcn(tm, y0, return_solver=True)
    eng = make_engine(nz, ny, vv, kh)
    fx, stats, _ = eng.comp_fcn(eng.upload(y0), replay=np.array(solver.schedule, dtype=np.float64))
    assert rel_err(eng.download(fx).reshape(-1), want) < 1e-10
    assert stats["nsweeps"] > 3 * stats["nnewton"]          # beyond the two-sweep minimum
    (fx, stats, _), (fx_def, _, _) = free_years(eng, eng.upload(y0))
    assert np.allclose(eng.download(fx).reshape(-1), want, rtol=1.0e-3, atol=1.0e-6)
    assert np.allclose(eng.download(fx_def).reshape(-1), want, rtol=1.0e-3, atol=1.0e-6)
    for key, ref in (("nfev", solver.stats.nfev), ("njev", solver.stats.njev), ("nlu", solver.stats.nlu)):
        assert abs(stats[key] - ref) <= 0.1 * ref + 5, (key, stats[key], ref)


@pytest.mark.parametrize("nz,ny,vv,kh", [(20, 3, 0.0, 0.0), (26, 26, 0.1, 1000.0)])
def test_default_mode_schedule_replayed_by_the_oracle(nz, ny, vv, kh):
    """the production mode from the other side: a free-running year in the engines' default mode (Jacobian at the second
    stage time of every attempt) records its steps; the CPU oracle -- SciPy's sparse LU, its own Jacobian and mixing
    coefficient functions -- replays exactly those steps and Jacobian times, and so does the device with the inner
    tolerance of a replay.  1e-10, as for SciPy's own schedules: stage planes, stage-time Jacobian and step boundary of
    the default mode compute what the restated reference functions compute."""
    from oracle import radau

    eng = make_engine(nz, ny, vv, kh)
    model, tm = oracle_iage(nz, ny, vv, kh)
    rng = np.random.default_rng(17)
    col = np.interp(model.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x0 = np.stack([np.broadcast_to(col[:, None], (nz, ny))] * 2) + 0.01 * rng.standard_normal((2, nz, ny))
    x = eng.upload(x0)
    eng.set_option("device_ctl", 0)
    fx, st, sched = eng.comp_fcn(x, record=True)
    off_start = int(np.sum(sched[:, 4] != sched[:, 0]))
    assert off_start > 0.9 * len(sched)              # the Jacobian times are stage times, not step starts
    rows = [(r[0], r[1], r[2], int(r[3]), r[4], r[5]) for r in sched]
    want = radau.comp_fcn(tm, x0.reshape(-1), replay=rows)
    got, _, _ = eng.comp_fcn(x, replay=sched)
    assert rel_err(eng.download(got).reshape(-1), want) < 1e-10
    # and the free-running year itself is that map to the Newton tolerance
    assert np.allclose(eng.download(fx).reshape(-1), want, rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("nz,ny,vv,kh", [(20, 3, 0.0, 0.0), (26, 26, 0.1, 1000.0)])
def test_frozen_product_against_the_oracle(nz, ny, vv, kh):
    """the finite-difference product on frozen years, device against CPU oracle: both difference two years on the SAME
    recorded steps (the oracle with SciPy's sparse LU), sigma as the reference takes it.  (52 x 52, whose two oracle years take
    160 s, and the deep grids of the benchmarked instantiation: tests/test_gpu_oracle_deep.py, in worker processes.)"""
    from oracle import radau

    eng = make_engine(nz, ny, vv, kh)
    model, tm = oracle_iage(nz, ny, vv, kh)
    weight = np.outer(model.depth.delta, model.ypos.delta)
    eng.set_region(np.ones((nz, ny), dtype=np.int32), weight)
    rng = np.random.default_rng(23)
    col = np.interp(model.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x0 = np.stack([np.broadcast_to(col[:, None], (nz, ny))] * 2) + 0.01 * rng.standard_normal((2, nz, ny))
    v = np.cumsum(rng.standard_normal(x0.shape), axis=1)
    x, vd = eng.upload(x0), eng.upload(v)
    vd = eng.scale(vd, 1.0 / np.sqrt(eng.dot(vd, vd)))
    v = eng.download(vd)
    eng.set_option("device_ctl", 0)
    fx, _, sched = eng.comp_fcn(x, record=True)
    w, sigma, stp = eng.jvp(x, fx, vd, sched=sched)
    assert eng.frozen_fallbacks() == 0 and stp["nrejected"] == 0
    rows = [(r[0], r[1], r[2], int(r[3]), r[4], r[5]) for r in sched]
    # the two CPU replays side by side (SuperLU releases the GIL; at 52 x 52 each takes two minutes on the GPU box's host)
    from concurrent.futures import ThreadPoolExecutor

    _, tm2 = oracle_iage(nz, ny, vv, kh)
    with ThreadPoolExecutor(max_workers=2) as pool:
        job0 = pool.submit(radau.comp_fcn, tm, x0.reshape(-1), replay=rows)
        job1 = pool.submit(radau.comp_fcn, tm2, (x0 + sigma[0] * v).reshape(-1), replay=rows)
        f0, f1 = job0.result(), job1.result()
    w_oracle = (f1 - f0) / sigma[0]
    assert rel_err(eng.download(w).reshape(-1), w_oracle) < 2e-3


def test_year_with_history_samples_is_the_same_year():
    """nk2d_comp_fcn_hist (the year that gives F(x) AND the 61 samples of hist_NN.nc, scipy ivp.py:707-723): the samples cost
    the steps that hold one their separate launches and nothing else -- the same decisions, the same F(x) bit for bit, the
    first sample the start, the last one the end of the year; an interior one against the CPU oracle's dense output"""
    from nk_ooc_amd.engine import iage_engine
    from nk_ooc_amd.grid import Grid2d

    n = 52
    eng = iage_engine(Grid2d.default(n, n))
    eng.set_option("stream_years", 0)       # (launch counts are compared below; as a command stream: tests/test_gpu_stream.py)
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    y0 = np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy()
    x = eng.upload(y0)
    fx, st, sched = eng.comp_fcn(x, record=True)
    t_eval = np.linspace(0.0, 365.0 * 86400.0, 61)
    fxh, sth, hist = eng.comp_fcn_hist(x, t_eval)
    assert np.array_equal(eng.download(fx), eng.download(fxh))
    for key in ("nsteps", "nrejected", "nnewton", "nfev"):
        assert st[key] == sth[key], key
    # (njev / nlu count launches that evaluate or factorise: the unfused boundary of a sample step books a Jacobian the
    # fused one shares with the next attempt -- a handful per year)
    assert abs(st["njev"] - sth["njev"]) <= 61 and abs(st["nlu"] - sth["nlu"]) <= 122
    # at most the 61 steps with a sample went without the fused boundary (a handful of launches each)
    assert st["nlaunch"] <= sth["nlaunch"] <= st["nlaunch"] + 61 * 12
    assert np.array_equal(hist[0], y0)
    # F(x) of iage is the end state minus the start state (iage.py / model_state.py comp_fcn)
    assert np.allclose(hist[-1] - y0, eng.download(fx), rtol=0.0, atol=1e-12 * np.abs(hist[-1]).max())
    # monotone ageing in the interior: every sample lies between its neighbours' extremes (a swapped or stale buffer would not)
    mid = hist[1:-1]
    assert np.all(np.isfinite(hist)) and np.all(mid.max(axis=(1, 2, 3)) <= hist[-1].max() * (1 + 1e-9) + 1e-12)
    # a sample is the dense output of the accepted step that holds it: the state a replayed year reaches at the end of that
    # step and at the end of the one before bracket it
    k = 30
    row = np.searchsorted(sched[:, 1], t_eval[k])
    assert sched[row, 0] < t_eval[k] <= sched[row, 1]
