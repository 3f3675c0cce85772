"""The forward year as a command stream (csrc/nk2d_stream.hip, nk2d_stream.h): ONE resident kernel executes the launches of
the host-controlled year as commands, workgroups handing over to their lateral neighbours.  The controller is the host's
own -- same decisions -- and the commands run the device functions of the launches they replace, so the year is the
host-controlled year BIT FOR BIT: F(x), every accepted step of its schedule, every counter.  (Against the CPU oracle the
launch path is pinned in test_gpu_comp_fcn.py / test_gpu_oracle_deep.py; what equals it bit for bit is pinned with it.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _iage(nz, ny):
    from nk_ooc_amd.engine import iage_engine
    from nk_ooc_amd.grid import Grid2d

    eng = iage_engine(Grid2d.default(nz, ny))
    eng.set_option("stream_years", 0)       # (the library's default is 1: the tests switch it on where they compare)
    return eng


def _state(eng, seed=5):
    rng = np.random.default_rng(seed)
    tc, nz, ny = eng.shape
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    return np.stack([np.broadcast_to(col[:, None], (nz, ny))] * tc) + 0.01 * rng.standard_normal(eng.shape)


@pytest.mark.parametrize("nz,ny", [(26, 26), (52, 52), (130, 20), (250, 12), (416, 8), (512, 6)])
@pytest.mark.parametrize("mode", ["default", "scipy_decisions"])
def test_stream_year_is_the_host_controlled_year(nz, ny, mode):
    if mode == "scipy_decisions" and nz > 130:
        pytest.skip("SciPy's decisions: the two shallow grids and one of three levels per lane")
    eng = _iage(nz, ny)
    if mode == "scipy_decisions":
        eng.set_option("jac_stage", -1)
        eng.set_option("jac_fresh", 0)
    x = eng.upload(_state(eng))
    fx, st, sched = eng.comp_fcn(x, record=True)
    want = eng.download(fx)
    eng.set_option("stream_years", 1)
    fx_s, st_s, sched_s = eng.comp_fcn(x, record=True)
    assert eng.counter("stream_years_run") == 1 and eng.counter("stream_timeouts") == 0
    assert np.array_equal(sched_s, sched)                       # every accepted step, Newton count, Jacobian time, error
    assert np.array_equal(eng.download(fx_s), want)
    for key in ("nsteps", "nrejected", "nnewton", "nfev", "njev", "nlu", "nsolve", "nsweeps"):
        assert st_s[key] == st[key], key
    # a year is a handful of launches: the kernel, and whatever has no command (first attempt, several-sweep estimates)
    assert st_s["nlaunch"] < 0.2 * st["nlaunch"]
    assert eng.counter("stream_commands") > st["nnewton"]
    # and again (the ring's stamps go on, the flags are not reset): the same year
    fx_2, _, _ = eng.comp_fcn(x)
    assert np.array_equal(eng.download(fx_2), want) and eng.counter("stream_years_run") == 2
    eng.close()


@pytest.mark.parametrize("nz,ny", [(26, 26), (52, 52), (130, 20), (416, 8)])
def test_frozen_year_as_a_stream(nz, ny):
    """the frozen year of a product (a recorded schedule replayed for another state) as a command stream:
    the launch-per-phase frozen year bit for bit, for the recorded state (= the recorded year) and for a perturbed one,
    with the Newton check and the sampled error estimates of every frozen year; and a step replay of the same schedule"""
    eng = _iage(nz, ny)
    eng.set_option("frozen_persistent", 0)
    x0 = _state(eng)
    x = eng.upload(x0)
    zz, yy = np.linspace(0.0, 1.0, nz), np.linspace(0.0, 1.0, ny)
    xp = eng.upload(x0 * (1.0 + 1.0e-4 * np.outer(np.sin(3.0 * zz), np.cos(2.0 * yy))[None]))
    fx, st, sched = eng.comp_fcn(x, record=True)
    want = [eng.download(eng.comp_fcn_frozen(v, sched)[0]) for v in (x, xp)]
    _, st_l = eng.comp_fcn_frozen(xp, sched)
    replay_l = eng.download(eng.comp_fcn(xp, replay=sched)[0])
    eng.set_option("stream_years", 2)
    got = [eng.download(eng.comp_fcn_frozen(v, sched)[0]) for v in (x, xp)]
    _, st_s = eng.comp_fcn_frozen(xp, sched)
    assert eng.counter("stream_years_run") == 3 and eng.counter("stream_timeouts") == 0
    assert np.array_equal(want[0], eng.download(fx)) and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    for key in ("nsteps", "nnewton", "nfev", "nerr_checked"):
        assert st_s[key] == st_l[key], key
    assert st_s["max_err"] == st_l["max_err"] and st_s["nlaunch"] < 0.2 * st_l["nlaunch"]
    assert np.array_equal(eng.download(eng.comp_fcn(xp, replay=sched)[0]), replay_l)
    # the product through it
    vd = eng.upload(np.cumsum(np.random.default_rng(3).standard_normal(x0.shape), axis=1))
    eng.set_region(np.ones((nz, ny), dtype=np.int32), np.outer(eng.grid.depth.delta, eng.grid.ypos.delta))
    w_s, _, _ = eng.jvp(x, fx, vd, sched=sched)
    eng.set_option("stream_years", 0)
    w_l, _, _ = eng.jvp(x, fx, vd, sched=sched)
    assert np.array_equal(eng.download(w_s), eng.download(w_l)) and eng.frozen_fallbacks() == 0
    eng.close()


def _phos_state(eng, rng):
    tc, nz, ny = eng.shape
    prof = [np.interp(eng.grid.depth.mid, zs, vs) for zs, vs in (([1.3e2, 2.6e2], [5.5e-3, 4.1e0]), ([9.5e1, 1.4e2], [7.1e-2, 1.5e-4]),
                                                                 ([1.7e2, 2.5e2], [1.8e-2, 7.9e-4]))]
    return np.stack([np.broadcast_to(p[:, None], (nz, ny)) for p in prof]) * (1.0 + 0.05 * rng.random((3, nz, ny)))


@pytest.mark.parametrize("case", ["phosphorus_30x12", "phosphorus_130x6", "phosphorus_416x4", "forced_decay_26x26", "forced_file_sink_thres_22x9",
                                  "forced_file_restore_sms_22x9"])
def test_other_module_kinds_as_streams(case, golden_dir, tmp_path):
    """the modules whose Jacobian reads the state (phosphorus: three coupled tracers, a Jacobian command per attempt; forced with a
    thresholded sink) and the file-driven forced module (its forcing fields ride with the mixing planes): free-running year and
    frozen year as command streams against the same years by launches, bit for bit"""
    from nk_ooc_amd.engine import forced_engine, phosphorus_engine
    from nk_ooc_amd.grid import Grid2d

    rng = np.random.default_rng(12)
    if case.startswith("phosphorus"):
        nz, ny = (int(v) for v in case.split("_")[1].split("x"))
        eng = phosphorus_engine(Grid2d.default(nz, ny))
        x0 = _phos_state(eng, rng)
    elif case.startswith("forced_decay"):
        eng = forced_engine(Grid2d.default(26, 26), {"forced_surf_restore_opt": "none", "forced_sms_opt": "decay", "forced_sms_decay_rate": "1.0e-8"})
        bump = np.cumsum(np.cumsum(rng.standard_normal((1, 26, 26)), axis=1), axis=2)
        x0 = 1.0 + 0.3 * bump / np.max(np.abs(bump))
    else:
        from test_gpu_forced import _file_modelinfo

        g = np.load(f"{golden_dir}/{case}.npz")
        eng = forced_engine(Grid2d.default(int(g["nz"]), int(g["ny"])), _file_modelinfo(g, tmp_path))
        x0 = np.asarray(g["y0"]).reshape(eng.shape)
    eng.set_option("stream_years", 0)
    eng.set_option("frozen_persistent", 0)
    x = eng.upload(x0)
    fx, st, sched = eng.comp_fcn(x, record=True)
    xp = eng.upload(x0 * (1.0 + 1.0e-5 * np.cos(np.linspace(0.0, 3.0, x0.shape[1]))[None, :, None]))
    fz = eng.download(eng.comp_fcn_frozen(xp, sched)[0])
    eng.set_option("stream_years", 3)
    fx_s, st_s, sched_s = eng.comp_fcn(x, record=True)
    assert eng.counter("stream_years_run") == 1 and eng.counter("stream_timeouts") == 0
    assert np.array_equal(sched_s, sched) and np.array_equal(eng.download(fx_s), eng.download(fx))
    for key in ("nsteps", "nrejected", "nnewton", "nfev", "njev", "nlu"):
        assert st_s[key] == st[key], key
    assert st_s["nlaunch"] < 0.25 * st["nlaunch"]
    assert np.array_equal(eng.download(eng.comp_fcn_frozen(xp, sched)[0]), fz) and eng.counter("stream_years_run") == 2
    eng.close()


@pytest.mark.parametrize("nz,ny,forced", [(320, 6, True), (416, 4, True), (260, 300, False)])
def test_phosphorus_two_waves_to_a_simd(nz, ny, forced):
    """option "stream_two_waves": the flavour of the resident kernel within 256 registers (what does not fit lives in scratch
    memory), two waves to a SIMD -- taken by itself where the one-wave kernel cannot hold every column (260 x 300: 900 columns
    of five levels per lane, one ypos column per workgroup instead of two), forced on the narrow grids: the same years bit for bit"""
    from nk_ooc_amd.engine import phosphorus_engine
    from nk_ooc_amd.grid import Grid2d

    rng = np.random.default_rng(21)
    res = {}
    for two in (0, 2 if forced else 1):
        eng = phosphorus_engine(Grid2d.default(nz, ny))
        eng.set_option("stream_two_waves", two)
        eng.set_option("stream_years", 3)
        eng.set_option("frozen_persistent", 0)
        x0 = _phos_state(eng, np.random.default_rng(21))
        x = eng.upload(x0)
        fx, st, sched = eng.comp_fcn(x, record=True)
        xp = eng.upload(x0 * (1.0 + 1.0e-5 * np.cos(np.linspace(0.0, 3.0, nz))[None, :, None]))
        fz = eng.download(eng.comp_fcn_frozen(xp, sched)[0])
        assert eng.counter("stream_years_run") == 2 and eng.counter("stream_timeouts") == 0
        assert eng.counter("stream_two_waves_kernel") == (1 if two else 0)
        res[two] = (eng.download(fx), fz, sched, eng.counter("stream_columns_per_workgroup"), st["seconds"])
        eng.close()
    a, b = res[0], res[2 if forced else 1]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    if not forced:
        assert a[3] == 2 and b[3] == 1, (a[3], b[3])
    print(f"phosphorus {nz} x {ny}: free-running year {a[4]:.3f} s one wave to a SIMD ({a[3]} columns per workgroup), "
          f"{b[4]:.3f} s two ({b[3]})")
    del rng


def test_year_with_history_samples_as_a_stream():
    """the 61 samples of a history file have no command: each ends the kernel, runs its launches, the next command starts the
    kernel again -- same year, same samples"""
    n = 52
    eng = _iage(n, n)
    x = eng.upload(_state(eng))
    t_eval = np.linspace(0.0, 365.0 * 86400.0, 61)
    fx, st, hist = eng.comp_fcn_hist(x, t_eval)
    eng.set_option("stream_years", 1)
    fx_s, st_s, hist_s = eng.comp_fcn_hist(x, t_eval)
    assert np.array_equal(eng.download(fx_s), eng.download(fx)) and np.array_equal(hist_s, hist)
    assert eng.counter("stream_years_run") == 1 and eng.counter("stream_launches") >= 50
    eng.close()


def test_a_kernel_that_gives_up_hands_the_year_back():
    """time limit zero: the first wait of the kernel gives up; the year is rerun by launches, counted, and is the same year"""
    eng = _iage(52, 52)
    x = eng.upload(_state(eng))
    fx, st, _ = eng.comp_fcn(x)
    eng.set_option("stream_years", 1)
    eng.set_option("barrier_timeout_ms", 0)
    fx_s, st_s, _ = eng.comp_fcn(x)
    assert np.array_equal(eng.download(fx_s), eng.download(fx))
    assert st_s["nbarrier_timeouts"] == 1 and eng.counter("stream_timeouts") >= 1 and eng.counter("stream_years_run") == 0
    eng.set_option("barrier_timeout_ms", 2000)
    fx_s, st_s, _ = eng.comp_fcn(x)
    assert np.array_equal(eng.download(fx_s), eng.download(fx)) and eng.counter("stream_years_run") == 1
    eng.close()


def test_commands_through_the_relay_wave(monkeypatch):
    """NK2D_STREAM_RELAY=1: the host writes its commands into pinned memory and a relay wave of the kernel copies them into HBM (what a
    device without a large BAR gets): the same year"""
    eng = _iage(52, 52)
    x = eng.upload(_state(eng))
    fx, st, sched = eng.comp_fcn(x, record=True)
    eng.close()
    monkeypatch.setenv("NK2D_STREAM_RELAY", "1")
    eng = _iage(52, 52)
    x = eng.upload(_state(eng))
    eng.set_option("stream_years", 1)
    fx_s, st_s, sched_s = eng.comp_fcn(x, record=True)
    assert eng.counter("stream_years_run") == 1 and eng.counter("stream_timeouts") == 0
    assert np.array_equal(sched_s, sched) and st_s["nlaunch"] < 0.2 * st["nlaunch"]
    eng.close()


def test_two_engines_driven_from_two_threads():
    """two contexts, a host thread each (as ModelState drives its tracer modules): their resident kernels take turns -- or run side by
    side where their waves leave the chip room -- and each year is the year of its engine alone"""
    from concurrent.futures import ThreadPoolExecutor

    engs = [_iage(52, 52), _iage(130, 20)]
    xs = [e.upload(_state(e, seed=9 + i)) for i, e in enumerate(engs)]
    want = [e.download(e.comp_fcn(x)[0]) for e, x in zip(engs, xs)]
    for e in engs:
        e.set_option("stream_years", 1)

    def years(k):
        return [engs[k].download(engs[k].comp_fcn(xs[k])[0]) for _ in range(3)]

    with ThreadPoolExecutor(max_workers=2) as pool:
        got = list(pool.map(years, range(2)))
    for k in range(2):
        for g in got[k]:
            assert np.array_equal(g, want[k])
        assert engs[k].counter("stream_years_run") == 3 and engs[k].counter("stream_timeouts") == 0
        engs[k].close()


def test_full_size_stream_year():
    """416 x 416: the year that produces F(x), bit for bit, and faster than by launches"""
    n = 416
    eng = _iage(n, n)
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
    fx, st, sched = eng.comp_fcn(x, record=True)
    fx, st, sched = eng.comp_fcn(x, record=True)
    eng.set_option("stream_years", 1)
    fx_s, st_s, sched_s = eng.comp_fcn(x, record=True)
    fx_s, st_s, sched_s = eng.comp_fcn(x, record=True)
    assert np.array_equal(sched_s, sched) and np.array_equal(eng.download(fx_s), eng.download(fx))
    assert eng.counter("stream_years_run") == 2 and eng.counter("stream_timeouts") == 0
    print(f"416^2 free-running year: {st['seconds']:.3f} s by launches ({st['nlaunch']} launches), "
          f"{st_s['seconds']:.3f} s as a command stream ({st_s['nlaunch']} launches, {eng.counter('stream_commands') // 2} commands)")
    assert st_s["seconds"] < st["seconds"]
    eng.close()
