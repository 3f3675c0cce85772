"""The Krylov loop behind the C ABI: nk2d_jvp and nk2d_gmres_solve against the
Python mirror of KrylovSolver (which the other tests pin to the reference and the oracle), the fused
multi-dot / CGS-2 building blocks, and the norm hook that couples the Radau controllers of a module whose
tracers are sharded over contexts (section 8(e), level 2)."""
import os
import threading

import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu


def _setup(tmp_path, nz, ny, vv=0.1, kh=1000.0, **solverinfo):
    from nk_ooc_amd.model_config import ModelConfig
    from nk_ooc_amd.model_state import ModelState
    from nk_ooc_amd.setup_solver import gen_grid_vars_file, make_config

    extra = {"max_abs_vvel": repr(vv), "horiz_mix_coeff": repr(kh)}
    cfg = make_config(str(tmp_path), nz, ny, extra_solverinfo=solverinfo, extra_modelinfo=extra)
    gen_grid_vars_file(cfg["modelinfo"])
    ModelState.reset_class()
    ModelState.write_files = True
    ModelState.model_config_obj = ModelConfig(cfg["modelinfo"])
    iterate = ModelState("gen_init_iterate")
    iterate += iterate.comp_fcn(os.path.join(str(tmp_path), "fcn_init.nc"), None)
    fcn = iterate.comp_fcn(os.path.join(str(tmp_path), "fcn_00.nc"), None)
    return cfg, iterate, fcn


@pytest.mark.parametrize("nz,ny,vv,kh", [(26, 26, 0.1, 1000.0), (20, 3, 0.0, 0.0)])
def test_jvp_entry_point_is_the_python_sequence(tmp_path, nz, ny, vv, kh):
    """one C call == perturb + forward year + difference done call by call (model_state_base.py:492-527);
    the 20 x 3 case has three column regions, i.e. three different sigmas"""
    from nk_ooc_amd.model_state import ModelState

    _, iterate, fcn = _setup(tmp_path, nz, ny, vv, kh)
    eng = iterate.tracer_modules[0].eng
    x, fx = iterate.tracer_modules[0].vec, fcn.tracer_modules[0].vec
    rng = np.random.default_rng(3)
    v = eng.upload(rng.standard_normal(eng.shape))
    v = eng.scale(v, 1.0 / np.sqrt(eng.dot(v, v)))
    sigma = 1.0e-4 * np.sqrt(eng.dot(x, x))
    fpert, _, _ = eng.comp_fcn(eng.axpby(1.0, x, 1.0, eng.scale(v, sigma)))
    want = eng.download(eng.diff_scale(fpert, fx, 1.0 / sigma))
    keep = eng.new_vec()
    w, sig, stats = eng.jvp(x, fx, v, perturb_fcn=keep)
    assert np.array_equal(sig, sigma) and len(sig) == eng.nreg
    assert np.array_equal(eng.download(w), want)
    assert np.array_equal(eng.download(keep), eng.download(fpert))
    assert stats["nsteps"] > 50
    ModelState.reset_class()


@pytest.mark.parametrize("nz,ny,vv,kh", [(26, 26, 0.1, 1000.0), (20, 3, 0.0, 0.0)])
def test_gmres_entry_point_against_krylov_solver(tmp_path, nz, ny, vv, kh):
    from nk_ooc_amd.krylov_solver import KrylovSolver
    from nk_ooc_amd.model_state import ModelState

    iters = 4
    cfg, iterate, fcn = _setup(tmp_path, nz, ny, vv, kh, krylov_rel_tol="1.0e-30", krylov_max_iter=str(iters))
    solverinfo = dict(cfg["solverinfo"], krylov_workdir=os.path.join(str(tmp_path), "krylov_00"))
    solver = KrylovSolver(iterate, solverinfo, False, False, None)
    inc = solver.solve(os.path.join(str(tmp_path), "increment_00.nc"), fcn)
    state = solver._solver_state
    beta, h_mat = state.get_value_saved_state("beta"), state.get_value_saved_state("h_mat")
    eng = iterate.tracer_modules[0].eng
    # the perturbed years repeat the accepted steps of the year that produced fcn, there (ModelState carries them with
    # fcn) and here (handed to the C entry point)
    sched = fcn._sched["iage"]
    assert len(sched) > 50
    got, info = eng.gmres_solve(iterate.tracer_modules[0].vec, fcn.tracer_modules[0].vec, 1.0e-30, 0, iters, sched=sched)
    assert info["iters"] == iters == solver.get_iteration()
    # same kernels in the same order up to the least-squares solve: the Krylov space is identical
    assert np.array_equal(info["beta"], beta[0])
    assert np.array_equal(info["h_mat"], h_mat[0])
    # Givens QR here, LAPACK gelsd there: the same minimiser to rounding
    assert rel_err(eng.download(got), eng.download(inc.tracer_modules[0].vec)) < 1e-10
    from nk_ooc_amd.krylov_solver import least_squares_coeffs

    assert np.allclose(info["coeff"], least_squares_coeffs(beta, h_mat)[0], rtol=1e-10, atol=1e-14)
    # stopping rule: with a loose tolerance and min_iter the loop stops where the reference's would
    _, early = eng.gmres_solve(iterate.tracer_modules[0].vec, fcn.tracer_modules[0].vec, 0.9, 2, iters, sched=sched)
    want = next(k + 1 for k in range(iters)
                if k + 1 >= 2 and (info["resid_norm"][k] < 0.9 * info["beta"]).all())
    assert early["iters"] == want
    ModelState.reset_class()


def test_gmres_entry_point_refuses_what_it_cannot_do(tmp_path):
    from nk_ooc_amd.engine import Nk2dError, phosphorus_engine
    from nk_ooc_amd.grid import Grid2d

    eng = phosphorus_engine(Grid2d.default(22, 9))
    x = eng.upload(np.ones(eng.shape))
    with pytest.raises(Nk2dError, match="phosphorus"):
        eng.gmres_solve(x, x, 1e-2, 0, 2)
    with pytest.raises(Nk2dError, match="max_iter"):
        eng.gmres_solve(x, x, 1e-2, 0, 0)


def test_multi_dot_and_cgs2():
    from nk_ooc_amd.dist import ShardComm, ShardedVectorSpace
    from nk_ooc_amd.engine import iage_engine
    from nk_ooc_amd.grid import Grid2d

    nz, ny = 70, 12
    eng = iage_engine(Grid2d.default(nz, ny))
    mask = np.ones((nz, ny), dtype=np.int32)
    mask[:, 5:] = 2
    mask[3, 4] = 0
    eng.set_region(mask, np.outer(np.linspace(1.0, 3.0, nz), np.linspace(2.0, 1.0, ny)))
    rng = np.random.default_rng(5)
    vecs = [eng.upload(rng.standard_normal(eng.shape)) for _ in range(6)]
    # orthonormal basis by the library's modified Gram-Schmidt
    basis = []
    for v in vecs[:5]:
        eng.mgs(v, basis)
        basis.append(eng.scale(v, 1.0 / np.sqrt(eng.dot(v, v))))
    w = vecs[5]
    dots = eng.multi_dot(w, basis)
    assert np.array_equal(dots, np.stack([eng.dot(w, b) for b in basis]))   # same products, same association
    w_mgs = w.copy()
    h_mgs = eng.mgs(w_mgs, basis)
    w_cgs = w.copy()
    h_cgs = ShardedVectorSpace(eng, ShardComm()).cgs2(w_cgs, basis)
    assert np.allclose(h_cgs, h_mgs, rtol=1e-12, atol=1e-14)
    assert rel_err(eng.download(w_cgs), eng.download(w_mgs)) < 1e-13
    assert np.max(np.abs(eng.multi_dot(w_cgs, basis))) < 1e-15 * np.sqrt(eng.dot(w, w)).max() * 10


def test_norm_hook_couples_two_single_tracer_engines():
    """iage's two tracers on two contexts, their Radau controllers coupled through the hook (two host
    threads, the all-reduce is a barrier): both take the same decisions, and the one-context engine
    REPLAYING their schedule reproduces their year to 1e-10 -- sharding changes the association of the norm
    sums (one more rounding), nothing else.  Free-running, the two layouts agree at the CI tolerance."""
    from nk_ooc_amd.dist import iage_shard_engine
    from nk_ooc_amd.engine import iage_engine
    from nk_ooc_amd.grid import Grid2d

    nz = ny = 26
    grid = Grid2d.default(nz, ny)
    col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    y0 = np.stack([np.broadcast_to(col[:, None], (nz, ny))] * 2) * (
        1.0 + 0.05 * np.random.default_rng(2).standard_normal((2, nz, ny)))

    class BarrierComm:
        """all-reduce over two threads of one process"""

        def __init__(self):
            self.barrier = threading.Barrier(2)
            self.slots = [0.0, 0.0]
            self.calls = 0

        def bind(self, rank):
            def allreduce(arr):
                self.slots[rank] = np.array(arr, dtype=np.float64)
                self.barrier.wait()
                total = self.slots[0] + self.slots[1]
                self.barrier.wait()
                if rank == 0:
                    self.calls += 1
                return total

            def allreduce_scalar(val):
                return float(allreduce(np.array([val]))[0])
            return type("C", (), {"allreduce_scalar": staticmethod(allreduce_scalar), "allreduce": staticmethod(allreduce)})

    comm = BarrierComm()
    # inner tolerance 1e-3 = what a replay solves to: the shards' year and its replay are then the same arithmetic
    shards = [iage_shard_engine(grid, r, comm.bind(r), lin_tol=1.0e-3) for r in range(2)]
    for eng in shards:
        eng.set_option("jac_fresh", 0)
        eng.set_option("growth_cap", 0)
    out = [None, None]

    def run(rank):
        eng = shards[rank]
        fx, stats, sched = eng.comp_fcn(eng.upload(y0[rank:rank + 1]), record=True)
        out[rank] = (eng.download(fx), stats, sched)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert all(o is not None for o in out)
    # identical accepted-step schedules, error estimates included (module-wide norms through the hook); the last column
    # is each shard's own fingerprint (its own tracer's description)
    assert np.array_equal(out[0][2][:, :7], out[1][2][:, :7])
    assert out[0][2][0, 7] != out[1][2][0, 7]
    for key in ("nsteps", "nrejected", "nnewton", "nfev", "njev", "nlu"):
        assert out[0][1][key] == out[1][1][key], key
    # Round 3, the vector hook: the norm of a Newton iteration travels with the norm of the iteration queued behind it (or
    # with the error estimate queued behind an iteration predicted to be the last) -- fewer than half the collectives a
    # norm-by-norm hook needs (one per Newton iteration and one per error estimate, counted below), the same decisions
    paired_calls = comm.calls
    reads = out[0][1]["nnewton"] + out[0][1]["nsteps"] + out[0][1]["nrejected"]
    paired = [o for o in out]
    comm.calls = 0
    for rank, eng in enumerate(shards):
        eng.set_norm_hook(comm.bind(rank).allreduce_scalar, 2.0 * nz * ny)
    threads = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert comm.calls >= reads                               # norm by norm: one all-reduce per norm the controller read
    assert paired_calls < 0.5 * comm.calls, (paired_calls, comm.calls)
    for rank in range(2):
        assert np.array_equal(out[rank][2][:, :7], paired[rank][2][:, :7])      # identical schedules
        assert np.array_equal(out[rank][0], paired[rank][0])                      # identical years, bit for bit
    # ... and with ONE iteration queued ahead instead of two (option "hook_spec_depth"): more collectives, the same year
    scalar_calls, comm.calls = comm.calls, 0
    for rank, eng in enumerate(shards):
        eng.set_norm_hook(comm.bind(rank).allreduce, 2.0 * nz * ny, vector=True)
        eng.set_option("hook_spec_depth", 1)
    threads = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert paired_calls < comm.calls < scalar_calls
    for rank in range(2):
        assert np.array_equal(out[rank][2][:, :7], paired[rank][2][:, :7])
        assert np.array_equal(out[rank][0], paired[rank][0])
    sharded = np.concatenate([out[0][0], out[1][0]])
    whole = iage_engine(grid, lin_tol=1.0e-3)
    whole.set_option("device_ctl", 0)       # host control, as a hooked engine runs (the hook lives on the host)
    whole.set_option("jac_fresh", 0)
    whole.set_option("growth_cap", 0)
    fx, _, _ = whole.comp_fcn(whole.upload(y0), replay=out[0][2])
    assert rel_err(whole.download(fx), sharded) < 1e-10
    fx_free, stats_free, _ = whole.comp_fcn(whole.upload(y0))
    assert np.allclose(whole.download(fx_free), sharded, rtol=1e-3, atol=1e-6)
    assert abs(stats_free["nsteps"] - out[0][1]["nsteps"]) <= 0.05 * stats_free["nsteps"] + 3
    # an identity hook on the one-context engine changes nothing
    whole.set_norm_hook(lambda s: s, 2.0 * nz * ny)
    fx_id, stats_id, _ = whole.comp_fcn(whole.upload(y0))
    assert np.array_equal(whole.download(fx_id), whole.download(fx_free))
    whole.set_norm_hook(None, 0.0)


def test_plugin_backend_on_the_device():
    """the HIP side of the variant-B plugin (tests/ref_plugin/py_driver_2d_hip/_backend.py) without the reference
    around it: what it hands the reference's comp_fcn in place of solve_ivp's result, and the preconditioner.
    (The reference side of the same plugin runs under the real nk_driver in tests/test_ref_dropin.py.)"""
    import sys
    import types

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_plugin"))
    from py_driver_2d_hip import _backend

    from nk_ooc_amd.engine import iage_engine
    from nk_ooc_amd.grid import Grid2d

    nz, ny = 26, 26
    grid = Grid2d.default(nz, ny)
    weight = np.outer(grid.depth.delta, grid.ypos.delta)
    mask = np.ones((nz, ny), dtype=np.int32)
    modelinfo = {"depth_axisname": "depth", "ypos_axisname": "ypos", "depth_units": "m", "ypos_units": "m",
                 "max_abs_vvel": "0.1", "horiz_mix_coeff": "1000.0"}
    ms_cls = types.SimpleNamespace(model_config_obj=types.SimpleNamespace(modelinfo=modelinfo),
                                   depth=grid.depth, ypos=grid.ypos)
    tm = types.SimpleNamespace(name="iage", _tracer_module_def={"tracers": {"iage": {}, "iage_slow_rest": {}}},
                               get_grid_vars=lambda name: {"region_mask": mask, "grid_weight": weight})
    backend = _backend.HipBackend(ms_cls)
    col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    y0 = np.stack([np.broadcast_to(col[:, None], (nz, ny))] * 2).copy()
    year = 365.0 * 86400.0
    eng = iage_engine(grid)
    fx, _, _ = eng.comp_fcn(eng.upload(y0))
    want = eng.download(fx)
    t, y = backend.forward_year(tm, y0.reshape(-1), np.array([0.0, year]))
    assert y.shape == (y0.size, 2) and np.array_equal(y[:, 0], y0.reshape(-1))
    assert np.array_equal(y[:, 1], y0.reshape(-1) + want.reshape(-1))         # what `sol.y[:, -1]` is read from
    # the steps of that year, as the plugin's ModelState attaches them to the result; a year frozen on them is the
    # year again, and the perturbed year of a product around it takes no decisions of its own
    scheds = backend.last_schedules()
    assert set(scheds) == {"iage"} and len(scheds["iage"]) > 100
    _, y_frozen = backend.forward_year(tm, y0.reshape(-1), np.array([0.0, year]), frozen=scheds)
    assert np.array_equal(y_frozen, y)
    t_eval = np.linspace(0.0, year, 61)
    t, y = backend.forward_year(tm, y0.reshape(-1), t_eval)
    assert y.shape == (y0.size, 61) and np.array_equal(t, t_eval)
    # the sampled year runs under host control, the plain one (26 levels) in the persistent kernel: two free-running
    # years of the same state
    assert np.allclose(y[:, -1] - y0.reshape(-1), want.reshape(-1), rtol=1.0e-3, atol=1.0e-6)
    v = np.random.default_rng(1).standard_normal(y0.shape)
    assert np.array_equal(backend.precond_apply(tm, v), eng.download(eng.precond_apply(eng.upload(v))))
    with pytest.raises(NotImplementedError):
        backend.engine(types.SimpleNamespace(name="phosphorus"))


def test_frozen_year_reproduces_the_recorded_year_and_checks_newton():
    """internal numerical differentiation (nk2d_comp_fcn_frozen / nk2d_set_frozen_schedule): on the recorded state a
    frozen year IS the recorded year, bit for bit, in the engines' default mode and with SciPy's decisions; the product
    (F(x + sigma v) - F(x)) / sigma through frozen years is far closer to the product of years integrated 10^4 times
    tighter than through free-running ones; a state the recorded Newton counts do not converge for is refused (-7)
    and nk2d_jvp falls back to a free-running year"""
    from nk_ooc_amd.engine import Nk2dFrozenMismatch, iage_engine
    from nk_ooc_amd.grid import Grid2d

    n = 26
    grid = Grid2d.default(n, n)
    eng = iage_engine(grid)
    tight = iage_engine(grid, rtol=1.0e-10, atol=1.0e-10, lin_tol=1.0e-10)
    tight.set_option("jac_fresh", 0)
    rng = np.random.default_rng(3)
    col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2) + 0.01 * rng.standard_normal((2, n, n)))
    xh = eng.download(x)
    v = np.cumsum(np.cumsum(rng.standard_normal(xh.shape), axis=1), axis=2)
    v /= np.sqrt(np.sum(v * v))
    xt = tight.upload(xh)
    fxt, _, _ = tight.comp_fcn(xt)
    w_ref = tight.download(tight.jvp(xt, fxt, tight.upload(v))[0])
    for ctl, fresh in ((0, 1), (1, 1), (0, 0)):       # (base year by launches / as a command stream)
        eng.set_option("stream_years", ctl)
        eng.set_option("jac_fresh", fresh)
        fx, st, sched = eng.comp_fcn(x, record=True)
        assert np.array_equal(eng.last_schedule(), sched) and len(sched) == st["nsteps"]
        fx2, st2 = eng.comp_fcn_frozen(x, sched)
        assert np.array_equal(eng.download(fx2), eng.download(fx)), (ctl, fresh)
        # (the free run also counts the iterations of its rejected and failed attempts)
        assert st2["nsteps"] == st["nsteps"] and st2["nnewton"] == int(sched[:, 3].sum()) <= st["nnewton"]
        assert st2["nrejected"] == 0
        w_frozen = eng.download(eng.jvp(x, fx, eng.upload(v), sched=sched)[0])
        err_frozen = rel_err(w_frozen, w_ref)
        assert err_frozen < 2.0e-3, (ctl, fresh, err_frozen)
    assert eng.frozen_fallbacks() == 0
    # the recorded counts of a year that converged at once (a uniform state under pure decay: no transport at all)
    # are not enough for a perturbed state with structure
    eng.set_option("stream_years", 1)
    eng.set_option("jac_fresh", 1)
    flat = eng.upload(np.zeros((2, n, n)))
    f_flat, _, s_flat = eng.comp_fcn(flat, record=True)
    bumpy = eng.upload(50.0 * rng.standard_normal((2, n, n)))
    try:
        eng.comp_fcn_frozen(bumpy, s_flat)
        refused = False
    except Nk2dFrozenMismatch:
        refused = True
    if refused:
        assert eng.frozen_fallbacks() == 1
        # nk2d_jvp: same situation, falls back to a free-running perturbed year and still returns a product
        w, _, stp = eng.jvp(flat, f_flat, bumpy, sched=s_flat)
        assert np.isfinite(eng.download(w)).all()
    eng.close()
    tight.close()


def test_gmres_residuals_are_true_with_frozen_products():
    """what GMRES reports after j iterations is the TRUE preconditioned residual of its iterate -- measured with the
    derivative of the map integrated 10^4 times tighter -- when the products run on frozen years
    (tools/probe_gmres_quality.py: with free-running products the solver reports 1e-6 where it has 2e-4)"""
    from nk_ooc_amd.engine import iage_engine
    from nk_ooc_amd.grid import Grid2d

    n, k = 26, 4
    grid = Grid2d.default(n, n)
    eng = iage_engine(grid)
    tight = iage_engine(grid, rtol=1.0e-10, atol=1.0e-10, lin_tol=1.0e-10)
    tight.set_option("jac_fresh", 0)
    weight = np.outer(grid.depth.delta, grid.ypos.delta)
    for e in (eng, tight):
        e.set_region(np.ones((n, n), dtype=np.int32), weight)
    col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
    x = eng.axpby(1.0, x, 1.0, eng.comp_fcn(x)[0])
    fx, _, sched = eng.comp_fcn(x, record=True)
    xt = tight.upload(eng.download(x))
    fxt, _, sched_t = tight.comp_fcn(xt, record=True)
    eng.precond_setup()
    tight.precond_setup()
    inc, info = eng.gmres_solve(x, fx, 0.0, 0, k, sched=sched)
    reported = info["resid_norm"][:, 0] / info["beta"][0]
    assert reported[-1] < 1.0e-3 * reported[0]
    inc_t = tight.upload(eng.download(inc))
    nrm = np.sqrt(tight.dot(inc_t, inc_t)[0])
    w, _, _ = tight.jvp(xt, fxt, tight.scale(inc_t, 1.0 / nrm), sched=sched_t)
    r = tight.axpby(1.0, fxt, nrm, w)
    pr, p0 = tight.precond_apply(r), tight.precond_apply(fxt)
    true = float(np.sqrt(tight.dot(pr, pr)[0] / tight.dot(p0, p0)[0]))
    assert abs(true - reported[-1]) < 0.05 * reported[-1], (true, reported)
    assert eng.frozen_fallbacks() == 0
    eng.close()
    tight.close()


@pytest.mark.parametrize("kind", ["phosphorus", "forced_file", "forced_file_threshold"])
def test_frozen_year_other_module_kinds(kind):
    """a year frozen on its own recorded steps is the recorded year, bit for bit, for the modules whose Jacobian reads
    the state (phosphorus, a thresholded sink: Jacobian at the step start, step boundary launches of their own) and for a
    forced module with forcing files (planes carry the forcing bundle; steps end in their last Newton launch)"""
    from nk_ooc_amd.engine import ModuleEngine, phosphorus_engine
    from nk_ooc_amd.grid import Grid2d

    rng = np.random.default_rng(8)
    if kind == "phosphorus":
        eng = phosphorus_engine(Grid2d.default(30, 12))
        x0 = np.stack([2.0 + 0.1 * rng.standard_normal((30, 12)), 0.05 + 0.005 * rng.standard_normal((30, 12)),
                       0.01 + 0.001 * rng.standard_normal((30, 12))])
    else:
        nz, ny = 130, 11
        times = np.array([-10.0, 40.0, 95.0, 200.0, 300.0]) * 86400.0
        restore = 1.0 + 0.2 * rng.standard_normal((5, ny))
        sms = 3.0e-8 * rng.standard_normal((5, nz, ny))
        extra = {"sink_thres": 0.4} if kind == "forced_file_threshold" else {}
        eng = ModuleEngine(Grid2d.default(nz, ny), tc=1, surf_rate=(24.0 / 86400.0,), module_kind=2,
                           restore_series=(times, restore), sms_series=(times, sms),
                           time_range=(0.0, 40.0 * 86400.0), **extra)
        x0 = 0.6 + 0.2 * rng.standard_normal((1, nz, ny))
    x = eng.upload(x0)
    fx, st, sched = eng.comp_fcn(x, record=True)
    assert len(sched) == st["nsteps"] > 10
    fx2, st2 = eng.comp_fcn_frozen(x, sched)
    assert np.array_equal(eng.download(fx2), eng.download(fx))
    assert st2["nnewton"] == int(sched[:, 3].sum())
    # and a slightly different state passes the Newton check of the frozen year
    xp = eng.upload(x0 * (1.0 + 1.0e-5 * rng.standard_normal(x0.shape)))
    fx3, _ = eng.comp_fcn_frozen(xp, sched)
    assert np.isfinite(eng.download(fx3)).all() and eng.frozen_fallbacks() == 0
    eng.close()
