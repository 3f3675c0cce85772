"""Device engine of ONE tracer module: a thin object layer over the nk2d C ABI.

`ModuleEngine` owns an `nk2d_ctx` (one HIP stream on one GPU) and hands out
device-resident vectors (`DevVec`).  Nothing here computes on the host; every
method is a ctypes call into `csrc/libnk2d.so`.
"""

import ctypes
import os

import numpy as np

from . import _lib
from .grid import BLDEPTH_MIN, YEAR, bldepth_time_knots


# guaranteed contraction of the inner line-relaxation solves per simplified Newton iteration (inexact
# Newton); step-replay mode caps it at 1e-3
DEFAULT_LIN_TOL = 3.0e-2
# 1: the Jacobian is re-evaluated at the start of every Radau step (two small launches) instead of being
# reused across steps by SciPy's heuristic (made for a Jacobian that costs a Python double loop and two
# SuperLU factorisations): same ODE, same error control, ~15 % fewer simplified-Newton iterations
# (DESIGN.md section 3).  0 (or NK2D_JAC_FRESH=0 in the environment) follows SciPy decision for decision.
DEFAULT_JAC_FRESH = 1
# Stage whose time the Jacobian of a step attempt is taken at (nk2d_set_option "jac_stage"): 1 = t + 0.645 h, the
# second Radau node, instead of the step start (-1, SciPy).  The vertical mixing coefficient changes over a step;
# the simplified Newton iteration and the filter of the error estimate both work with ONE Jacobian for the three
# stages, and the one near the middle of the stage times (0.155, 0.645, 1) h is at most 0.49 h away from any of them
# instead of a whole h: at 416^2 a forward year takes 2617 steps and 10.9 k Newton iterations instead of 3568 and
# 17.5 k (0.46 s against 0.67 s), the result 0.06 of the CI tolerance away from the one with the Jacobian at the step
# start (tools/probe_jac_stage.py).  The plane of that time is computed for the stage anyway --
# the launch that computes it derives the Jacobian planes from it.  Modules whose Jacobian also reads the state
# (phosphorus, a thresholded sink) take the mixing plane of that time and the state of the step start (phosphorus
# 416^2: 2109 steps / 0.80 s instead of 2972 / 1.16 s).  NK2D_JAC_STAGE in the environment overrides.
DEFAULT_JAC_STAGE = 1
# largest growth factor of the step size after a step whose simplified Newton iteration failed at first and was
# repeated with half the step size (nk2d_set_option "growth_cap"): 1.0 is the rule of Hairer & Wanner's RADAU5
# (no growth), which SciPy's Radau dropped -- it tries up to 10 h again and fails on a third of its attempts here.
# The rule saves 9-13 % of a forward year but changes the step sizes, and with them the history samples by up
# to 1.8 times the tolerance of the reference's CI comparison (tools/probe_hist_modes.py): off by default.
DEFAULT_GROWTH_CAP = 0.0
# Free-running forward years run as COMMAND STREAMS by default (library option "stream_years", csrc/nk2d_stream.h): one
# resident kernel executes the host controller's launches as commands.  NK2D_STREAM_YEARS in the environment overrides
# (0: launch by launch; 1: free-running years; 3: frozen years too).


class Nk2dError(RuntimeError):
    pass


class Nk2dFrozenMismatch(Nk2dError):
    """a frozen year (comp_fcn_frozen) whose recorded Newton iteration counts do not converge for the state given"""


class Nk2dScheduleMismatch(Nk2dError):
    """a frozen year asked to repeat a schedule that was recorded under other options, another grid or another build of
    the library (its fingerprint is not this engine's)"""


def _dp(arr):
    return arr.ctypes.data_as(_lib.c_double_p)


def sched_rows(sched):
    """a schedule as [n, SCHED_WIDTH] rows.  Rows from elsewhere -- the oracle's / SciPy's accepted steps (t, t_new, h,
    n_newton, t_jac, h_lu) -- are padded with err = 0 and fingerprint = 0 (never checked; only step-replay mode takes them)"""
    sched = np.asarray(sched, dtype=np.float64)
    if sched.ndim != 2:
        raise ValueError(f"a schedule is a 2-d array of rows with 6 or {_lib.SCHED_WIDTH} columns")
    if sched.shape[1] == 6:
        sched = np.concatenate((sched, np.zeros((sched.shape[0], _lib.SCHED_WIDTH - 6))), axis=1)
    if sched.shape[1] != _lib.SCHED_WIDTH:
        raise ValueError(f"schedule rows must have 6 or {_lib.SCHED_WIDTH} columns")
    return np.ascontiguousarray(sched)


class DevVec:
    """a state vector (tracer, depth, ypos) of one tracer module, resident in HBM"""

    __slots__ = ("eng", "ptr")

    def __init__(self, eng, ptr):
        self.eng = eng
        self.ptr = ptr

    def __del__(self):
        eng, ptr = self.eng, self.ptr
        self.ptr = None
        if ptr is not None and eng is not None and eng._handle is not None:
            eng._lib.nk2d_vec_free(eng._handle, ptr)

    def to_host(self):
        return self.eng.download(self)

    def copy(self):
        res = self.eng.new_vec()
        self.eng._chk(self.eng._lib.nk2d_vec_copy(self.eng._ctx, res.ptr, self.ptr))
        return res


class PendingDownload:
    """second half of ModuleEngine.download_begin; ended exactly once (by result(), or when dropped)"""

    __slots__ = ("eng", "ticket")

    def __init__(self, eng, ticket):
        self.eng = eng
        self.ticket = ticket

    def result(self):
        eng, ticket = self.eng, self.ticket
        if ticket is None:
            raise RuntimeError("download already taken")
        self.ticket = None
        host = np.empty(eng.shape)
        if eng._handle is None:
            raise RuntimeError("download: the engine was closed before its values were taken")
        if eng._lib.nk2d_vec_download_end(eng._handle, ticket, _dp(host)) != 0:
            raise RuntimeError("nk2d_vec_download_end: the copy to the host failed")
        return host

    def __del__(self):
        eng, ticket = self.eng, self.ticket
        self.ticket = None
        if ticket is not None and eng is not None and eng._handle is not None:
            eng._lib.nk2d_vec_download_end(eng._handle, ticket, None)


class ModuleEngine:
    """HIP engine for one tracer module on one (depth, ypos) grid"""

    def __init__(self, grid, tc, surf_rate=(), decay_rate=(), const_src=0.0, surf_target=(),
                 device_id=0,
                 time_range=(0.0, YEAR), rtol=1.0e-6, atol=1.0e-6, max_step_frac=0.01,
                 lin_tol=None, module_kind=0, phos_params=None, light_lim=None,
                 restore_series=None, sms_series=None, sink_thres=None):
        self._lib = _lib.load()
        self._handle = None
        self.grid = grid
        self.nz = len(grid.depth)
        self.ny = len(grid.ypos)
        self.tc = int(tc)
        self._device_id = int(device_id)
        self.shape = (self.tc, self.nz, self.ny)
        self.nreg = 1
        desc = _lib.Desc()
        desc.nz, desc.ny, desc.tc, desc.device_id = self.nz, self.ny, self.tc, int(device_id)
        keep = [np.ascontiguousarray(a, dtype=np.float64) for a in (
            grid.depth.edges, grid.ypos.edges, grid.vvel, grid.wvel, grid.hmix_coeff,
            grid.bldepth_max)]
        (desc.depth_edges, desc.ypos_edges, desc.vvel, desc.wvel, desc.hmix_coeff,
         desc.bldepth_max) = [_dp(a) for a in keep]
        desc.bldepth_min = BLDEPTH_MIN
        tvals, fvals = bldepth_time_knots()
        desc.bld_tvals = (ctypes.c_double * 4)(*tvals)
        desc.bld_fvals = (ctypes.c_double * 4)(*fvals)
        desc.vmix_log_shallow = float(np.log(1.0e1))
        desc.vmix_log_deep = float(np.log(5.0e-4))
        desc.vmix_half_width = 20.0
        surf = list(surf_rate) + [0.0] * (_lib.MAX_TRACERS - len(surf_rate))
        decay = list(decay_rate) + [0.0] * (_lib.MAX_TRACERS - len(decay_rate))
        target = list(surf_target) + [0.0] * (_lib.MAX_TRACERS - len(surf_target))
        desc.surf_rate = (ctypes.c_double * _lib.MAX_TRACERS)(*surf)
        desc.surf_target = (ctypes.c_double * _lib.MAX_TRACERS)(*target)
        desc.decay_rate = (ctypes.c_double * _lib.MAX_TRACERS)(*decay)
        desc.const_src = float(const_src)
        desc.t0, desc.t1 = float(time_range[0]), float(time_range[1])
        desc.rtol, desc.atol = float(rtol), float(atol)
        desc.max_step_frac = float(max_step_frac)
        if lin_tol is None:
            lin_tol = float(os.environ.get("NK2D_LIN_TOL", DEFAULT_LIN_TOL))
        desc.lin_tol = float(lin_tol)
        desc.module_kind = int(module_kind)
        if module_kind == 1:
            keep.append(np.ascontiguousarray(light_lim, dtype=np.float64))
            assert keep[-1].shape == (self.nz, self.ny)
            desc.light_lim = _dp(keep[-1])
            desc.phos_params = (ctypes.c_double * 6)(*[float(v) for v in phos_params])
        if module_kind == 2:
            # forcing records on the model axes (forcing.load_forcing): (times, values)
            if restore_series is not None:
                times, vals = [np.ascontiguousarray(a, dtype=np.float64) for a in restore_series]
                assert vals.shape == (len(times), self.ny)
                keep += [times, vals]
                desc.restore_nrec, desc.restore_times, desc.restore_vals = len(times), _dp(times), _dp(vals)
            if sms_series is not None:
                times, vals = [np.ascontiguousarray(a, dtype=np.float64) for a in sms_series]
                assert vals.shape == (len(times), self.nz, self.ny)
                keep += [times, vals]
                desc.sms_nrec, desc.sms_times, desc.sms_vals = len(times), _dp(times), _dp(vals)
            desc.sink_thres = 0.0 if sink_thres is None else float(sink_thres)
        self.state_dependent_precond = module_kind == 2 and sms_series is not None and bool(sink_thres)
        self.module_kind = int(module_kind)
        self.light_lim = keep[-1] if module_kind == 1 else None
        self.phos = dict(zip(PHOSPHORUS_PARAM_NAMES, phos_params)) if module_kind == 1 else None
        ctx = ctypes.c_void_p()
        rc = self._lib.nk2d_create(ctypes.byref(desc), ctypes.byref(ctx))
        if rc != 0:
            msg = self._lib.nk2d_last_error(ctx).decode() if ctx else "allocation failed"
            if ctx:
                self._lib.nk2d_destroy(ctx)
            raise Nk2dError(f"nk2d_create failed ({rc}): {msg}")
        self._handle = ctx
        self.device_ctl = 0
        self._precond_ready = False
        if "NK2D_STREAM_YEARS" in os.environ:
            self.set_option("stream_years", float(os.environ["NK2D_STREAM_YEARS"]))
        self.set_option("jac_fresh", float(os.environ.get("NK2D_JAC_FRESH", DEFAULT_JAC_FRESH)))
        self.set_option("growth_cap", float(os.environ.get("NK2D_GROWTH_CAP", DEFAULT_GROWTH_CAP)))
        self.set_option("jac_stage", float(os.environ.get("NK2D_JAC_STAGE", DEFAULT_JAC_STAGE)))

    @property
    def _ctx(self):
        """the library context; a closed engine raises instead of handing NULL to the C ABI"""
        if self._handle is None:
            raise Nk2dError("this engine is closed (ModelState.reset_class / ModuleEngine.close)")
        return self._handle

    def close(self):
        if self._handle is not None:
            # (a vector file queued on the checkpoint trail's writer thread may still hold a download of this engine)
            from . import trail

            trail.drain()
            self._lib.nk2d_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def _chk(self, rc):
        if rc == -7:
            raise Nk2dFrozenMismatch(f"nk2d call failed ({rc}): {self._lib.nk2d_last_error(self._ctx).decode()}")
        if rc == -8:
            raise Nk2dScheduleMismatch(f"nk2d call failed ({rc}): {self._lib.nk2d_last_error(self._ctx).decode()}")
        if rc != 0:
            raise Nk2dError(f"nk2d call failed ({rc}): {self._lib.nk2d_last_error(self._ctx).decode()}")

    # ---- regions ------------------------------------------------------------
    def set_region(self, mask, weight):
        mask = np.ascontiguousarray(mask, dtype=np.int32)
        weight = np.ascontiguousarray(weight, dtype=np.float64)
        if mask.shape != (self.nz, self.ny) or weight.shape != (self.nz, self.ny):
            raise ValueError("region mask / weight must have shape (nz, ny)")
        nreg = int(mask.max())
        self._chk(self._lib.nk2d_set_region(
            self._ctx, mask.ctypes.data_as(_lib.c_int32_p), _dp(weight), nreg))
        self.nreg = nreg
        self._region = (mask, weight)
        self._ones_vec = None       # cached by the host mirrors; its region scaling changed

    # ---- vectors ------------------------------------------------------------
    def new_vec(self):
        ptr = ctypes.c_void_p()
        self._chk(self._lib.nk2d_vec_alloc(self._ctx, ctypes.byref(ptr)))
        return DevVec(self, ptr)

    def upload(self, host, out=None):
        host = np.ascontiguousarray(host, dtype=np.float64).reshape(self.shape)
        out = self.new_vec() if out is None else out
        self._chk(self._lib.nk2d_vec_upload(self._ctx, out.ptr, _dp(host)))
        return out

    def download(self, vec):
        host = np.empty(self.shape)
        self._chk(self._lib.nk2d_vec_download(self._ctx, vec.ptr, _dp(host)))
        return host

    def download_begin(self, vec):
        """first half of a download (nk2d_vec_download_begin): the copy into a pinned buffer of its own is queued on the
        engine's stream and this returns at once; `.result()` of what it returns -- on any thread -- waits for that copy
        alone and gives the host array.  For the checkpoint trail's writer thread (trail.py)."""
        ticket = ctypes.c_void_p()
        self._chk(self._lib.nk2d_vec_download_begin(self._ctx, vec.ptr, ctypes.byref(ticket)))
        return PendingDownload(self, ticket)

    def vec_tensor(self, vec):
        """zero-copy torch tensor over the HBM of a state vector (flat, the library's packed column layout): for collectives
        that are elementwise over whole vectors (dist.ColumnComm).  The caller orders the engine's stream against torch's."""
        import torch

        nv = self.tc * self.ny * ((self.nz + 63) // 64) * 64

        class _Holder:
            pass

        holder = _Holder()
        holder.__cuda_array_interface__ = {"shape": (nv,), "typestr": "<f8", "data": (int(vec.ptr.value), False),
                                           "version": 2, "strides": None}
        holder._keep = vec
        return torch.as_tensor(holder, device=torch.device("cuda", self._device_id))

    def set_option(self, name, value):
        self._chk(self._lib.nk2d_set_option(self._ctx, name.encode(), float(value)))
        if name == "device_ctl":
            self.device_ctl = int(value)

    def sync(self):
        self._chk(self._lib.nk2d_sync(self._ctx))

    # ---- deterministic kernels ------------------------------------------------
    def tend(self, t, y, out=None):
        out = self.new_vec() if out is None else out
        self._chk(self._lib.nk2d_tend(self._ctx, float(t), y.ptr, out.ptr))
        return out

    def vmix_coeff(self, t):
        host = np.empty((self.nz - 1, self.ny))
        self._chk(self._lib.nk2d_vmix_coeff(self._ctx, float(t), _dp(host)))
        return host

    def jacobian_diags(self, t):
        """(5, tc, nz, ny): up, south, centre, north, down"""
        host = np.empty((5, self.tc, self.nz, self.ny))
        self._chk(self._lib.nk2d_jacobian_diags(self._ctx, float(t), _dp(host)))
        return host

    def set_lin_state(self, y):
        """state the stand-alone Jacobian calls linearise about (phosphorus only uses it)"""
        self._chk(self._lib.nk2d_set_lin_state(self._ctx, y.ptr))

    def jacobian_apply(self, t, v):
        """J(t, lin_state) v as a new device vector"""
        out = self.new_vec()
        self._chk(self._lib.nk2d_jacobian_apply(self._ctx, float(t), v.ptr, out.ptr))
        return out

    def shifted_solve(self, t_jac, h, mu, b_re, b_im=None):
        """x = ((mu/h) I - J(t_jac))^-1 b; returns (x_re, x_im or None, sweeps)"""
        mu = complex(mu)
        x_re = self.new_vec()
        x_im = self.new_vec() if mu.imag != 0.0 else None
        sweeps = ctypes.c_int32()
        self._chk(self._lib.nk2d_shifted_solve(
            self._ctx, float(t_jac), float(h), mu.real, mu.imag, b_re.ptr,
            b_im.ptr if b_im is not None else None, x_re.ptr,
            x_im.ptr if x_im is not None else None, ctypes.byref(sweeps)))
        return x_re, x_im, sweeps.value

    # ---- the forward year -------------------------------------------------------
    def comp_fcn(self, x, out=None, replay=None, record=False, record_cap=65536):
        """F(x) = y(T) - x.  Returns (fx, stats dict, schedule or None)."""
        out = self.new_vec() if out is None else out
        stats = _lib.Stats()
        rp, rn = None, 0
        if replay is not None:
            replay = sched_rows(replay)
            rp, rn = _dp(replay), replay.shape[0]
        rec, recn = None, ctypes.c_int64(0)
        if record:
            rec = np.zeros((record_cap, _lib.SCHED_WIDTH))
        self._chk(self._lib.nk2d_comp_fcn(
            self._ctx, x.ptr, out.ptr, ctypes.byref(stats), rp, rn,
            _dp(rec) if rec is not None else None, record_cap if record else 0,
            ctypes.byref(recn)))
        sched = rec[: recn.value].copy() if record else None
        return out, stats.as_dict(), sched

    def comp_fcn_frozen(self, x, sched, out=None):
        """forward year on the accepted steps `sched` recorded by comp_fcn(..., record=True) on this engine under
        the same options: no decisions, nothing read back, the recorded year's own inner tolerance.  Returns
        (fx, stats dict)."""
        out = self.new_vec() if out is None else out
        stats = _lib.Stats()
        sched = sched_rows(sched)
        self._chk(self._lib.nk2d_comp_fcn_frozen(self._ctx, x.ptr, out.ptr, ctypes.byref(stats), _dp(sched),
                                                 sched.shape[0]))
        return out, stats.as_dict()

    def last_schedule(self):
        """accepted steps of the most recent free-running year of this engine, [n, SCHED_WIDTH]"""
        n = ctypes.c_int64(0)
        self._chk(self._lib.nk2d_last_schedule(self._ctx, None, 0, ctypes.byref(n)))
        out = np.zeros((n.value, _lib.SCHED_WIDTH))
        if n.value:
            self._chk(self._lib.nk2d_last_schedule(self._ctx, _dp(out), n.value, ctypes.byref(n)))
        return out

    def frozen_fallbacks(self):
        """how many frozen years of this engine were rejected by their Newton check so far"""
        n = ctypes.c_int64(0)
        self._chk(self._lib.nk2d_frozen_fallbacks(self._ctx, ctypes.byref(n)))
        return n.value

    def frozen_resumes(self):
        """how many frozen years of this engine were resumed from a checkpoint with one more Newton iteration so far"""
        n = ctypes.c_int64(0)
        self._chk(self._lib.nk2d_frozen_resumes(self._ctx, ctypes.byref(n)))
        return n.value

    def counter(self, name):
        """library counters by name: frozen_persistent_years, frozen_cache_builds, frozen_fallbacks, frozen_resumes"""
        n = ctypes.c_int64(0)
        self._chk(self._lib.nk2d_get_counter(self._ctx, name.encode(), ctypes.byref(n)))
        return n.value

    def cache_pending(self):
        """the slab of a large schedule cache is being allocated (by a thread of the library): frozen years run launch by launch meanwhile"""
        return self.counter("frozen_cache_pending") != 0

    def schedule_fingerprint(self):
        """what the steps this engine records now are stamped with (grid, module, tolerances, controller options, build)"""
        out = ctypes.c_double(0.0)
        self._chk(self._lib.nk2d_schedule_fingerprint(self._ctx, ctypes.byref(out)))
        return out.value

    def set_frozen_schedule(self, sched):
        """the schedule the perturbed years of jvp / gmres_solve repeat from now on (None: free-running years)"""
        if sched is None:
            self._chk(self._lib.nk2d_set_frozen_schedule(self._ctx, None, 0))
            return
        sched = sched_rows(sched)
        self._chk(self._lib.nk2d_set_frozen_schedule(self._ctx, _dp(sched), sched.shape[0]))

    # ---- sampled timing of the dominant kernel ----------------------------------------
    def profile_reset(self, every_n):
        self._chk(self._lib.nk2d_profile_reset(self._ctx, int(every_n)))

    def profile_read(self):
        """timing windows of the dominant kernel since profile_reset: `avg_us` per launch (net of
        the event overhead), `bytes` algorithmic bytes of the `samples` launches in the windows"""
        avg = ctypes.c_double()
        samples, launches, windows = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        nbytes, ovh = ctypes.c_double(), ctypes.c_double()
        self._chk(self._lib.nk2d_profile_read(self._ctx, ctypes.byref(avg), ctypes.byref(samples),
                                              ctypes.byref(launches), ctypes.byref(nbytes),
                                              ctypes.byref(ovh), ctypes.byref(windows)))
        return {"avg_us": avg.value, "samples": samples.value, "launches": launches.value,
                "bytes": nbytes.value, "event_overhead_us": ovh.value, "windows": windows.value}

    def profile_totals(self):
        """launches of the dominant kernel since profile_reset (timed or not) and their algorithmic bytes"""
        launches, nbytes = ctypes.c_int64(), ctypes.c_double()
        self._chk(self._lib.nk2d_profile_totals(self._ctx, ctypes.byref(launches), ctypes.byref(nbytes)))
        return {"launches": launches.value, "bytes": nbytes.value}

    def profile_shapes(self):
        """non-factorising launches of the dominant kernel since profile_reset by shape (stage + sweep + update,
        stage + first sweep, last sweep + update, middle sweep): counts and algorithmic bytes"""
        counts = (ctypes.c_int64 * 4)()
        nbytes = (ctypes.c_double * 4)()
        self._chk(self._lib.nk2d_profile_shapes(self._ctx, counts, nbytes))
        return {"counts": list(counts), "bytes": list(nbytes)}

    def profile_replay(self, shape, n=200):
        """n back-to-back launches of one shape of the dominant kernel inside one event pair (state of the last
        forward year, updates to scratch): microseconds per launch and algorithmic bytes per launch"""
        avg, nbytes = ctypes.c_double(), ctypes.c_double()
        self._chk(self._lib.nk2d_profile_replay(self._ctx, int(shape), int(n), ctypes.byref(avg), ctypes.byref(nbytes)))
        return {"avg_us": avg.value, "bytes": nbytes.value}

    def timer_begin(self):
        """first event of a HIP event pair on the context's own stream"""
        self._chk(self._lib.nk2d_timer_begin(self._ctx))

    def timer_end(self):
        """second event of the pair; waits for it and returns the elapsed milliseconds"""
        ms = ctypes.c_double()
        self._chk(self._lib.nk2d_timer_end(self._ctx, ctypes.byref(ms)))
        return ms.value

    # ---- the Krylov loop on the device (C entry points of SURVEY.md section 8(b)) -----------------
    def jvp(self, x, fx, v, out=None, perturb_fcn=None, sched=None):
        """finite-difference Jacobian-vector product in one call (model_state_base.py:492-527):
        returns (w, sigma [nreg], stats of the perturbed year).  sched: the accepted steps of the year that
        produced fx -- the perturbed year repeats them (otherwise whatever set_frozen_schedule installed, or a
        free-running year)"""
        out = self.new_vec() if out is None else out
        sigma = np.empty(self.nreg)
        stats = _lib.Stats()
        if sched is not None:
            self.set_frozen_schedule(sched)
        try:
            self._chk(self._lib.nk2d_jvp(self._ctx, x.ptr, fx.ptr, v.ptr, out.ptr,
                                         perturb_fcn.ptr if perturb_fcn is not None else None,
                                         _dp(sigma), ctypes.byref(stats)))
        finally:
            if sched is not None:
                self.set_frozen_schedule(None)
        return out, sigma, stats.as_dict()

    def gmres_solve(self, x, fx, rel_tol, min_iter, max_iter, out=None, sched=None):
        """KrylovSolver.solve for this one module, all on the device (krylov_solver.py:85-165).  Returns
        (increment, dict(beta [nreg], h_mat [iters+1, iters, nreg], resid_norm [iters, nreg],
        coeff [iters, nreg], iters))"""
        out = self.new_vec() if out is None else out
        beta = np.zeros(self.nreg)
        h_mat = np.zeros((max_iter + 1, max_iter, self.nreg))
        resid = np.zeros((max_iter, self.nreg))
        coeff = np.zeros((max_iter, self.nreg))
        iters = ctypes.c_int32()
        if not self._precond_ready and self.module_kind != 1:
            self.precond_setup()
        if sched is not None:
            self.set_frozen_schedule(sched)
        try:
            self._chk(self._lib.nk2d_gmres_solve(self._ctx, x.ptr, fx.ptr, float(rel_tol), int(min_iter),
                                                 int(max_iter), out.ptr, _dp(beta), _dp(h_mat), _dp(resid),
                                                 _dp(coeff), ctypes.byref(iters)))
        finally:
            if sched is not None:
                self.set_frozen_schedule(None)
        k = iters.value
        return out, {"beta": beta, "h_mat": h_mat[: k + 1, :k].copy(), "resid_norm": resid[:k].copy(),
                     "coeff": coeff[:k].copy(), "iters": k}

    def multi_dot(self, w, basis):
        """all region-weighted dots <w, basis[i]> in one launch: (n, nreg)"""
        out = np.empty((len(basis), self.nreg))
        ptrs = (ctypes.c_void_p * len(basis))(*[v.ptr for v in basis])
        self._chk(self._lib.nk2d_multi_dot(self._ctx, w.ptr, len(basis), ptrs, _dp(out)))
        return out

    def multi_axpy(self, w, basis, h, fill=1.0):
        """w -= sum_i bcast(h[i]) basis[i], in place, one launch; `fill` = broadcast value where the
        region mask is <= 0 (1.0: the reference's, 0.0: leave those cells alone)"""
        h = np.ascontiguousarray(h, dtype=np.float64).reshape(len(basis), self.nreg)
        ptrs = (ctypes.c_void_p * len(basis))(*[v.ptr for v in basis])
        self._chk(self._lib.nk2d_multi_axpy(self._ctx, w.ptr, len(basis), ptrs, _dp(h), float(fill)))
        return w

    def set_norm_hook(self, fcn, global_n, vector=False):
        """couple the integrator's scalar norms with other contexts holding tracers of the same module
        (dist.TracerShardedModule): `fcn(local_sum_of_squares) -> global sum`; None removes the hook.
        vector=True: `fcn(ndarray of local sums) -> ndarray of global sums` -- the controller then pairs the norm of a
        Newton iteration with that of the iteration (or error estimate) queued behind it in one call"""
        if fcn is None:
            self._norm_hook = None
            self._chk(self._lib.nk2d_set_norm_hook(self._ctx, None, None, 0.0))
            return
        if vector:
            def thunk(user, ptr, n):
                vals = np.ctypeslib.as_array(ptr, shape=(n,))
                vals[:] = fcn(vals.copy())

            self._norm_hook = _lib.NORM_HOOK_VEC(thunk)     # keep the thunk alive
            self._chk(self._lib.nk2d_set_norm_hook_vec(
                self._ctx, ctypes.cast(self._norm_hook, ctypes.c_void_p), None, float(global_n)))
            return
        self._norm_hook = _lib.NORM_HOOK(lambda user, val: float(fcn(val)))   # keep the thunk alive
        self._chk(self._lib.nk2d_set_norm_hook(
            self._ctx, ctypes.cast(self._norm_hook, ctypes.c_void_p), None, float(global_n)))

    def comp_fcn_hist(self, x, t_eval, out=None):
        """forward year with dense output: returns (fx, stats, hist [len(t_eval), tc, nz, ny])"""
        out = self.new_vec() if out is None else out
        t_eval = np.ascontiguousarray(t_eval, dtype=np.float64)
        hist = np.empty((len(t_eval),) + self.shape)
        stats = _lib.Stats()
        self._chk(self._lib.nk2d_comp_fcn_hist(self._ctx, x.ptr, out.ptr, ctypes.byref(stats),
                                               len(t_eval), _dp(t_eval), _dp(hist)))
        return out, stats.as_dict(), hist

    # ---- preconditioner -----------------------------------------------------------
    def precond_setup(self):
        self._chk(self._lib.nk2d_precond_setup(self._ctx))
        self._precond_ready = True

    def precond_setup_states(self, states):
        """preconditioner of a forced module whose Jacobian depends on the state: `states` = the
        tracer (device vectors) at the end of each third of the year (forced.py:222-236)"""
        ptrs = (ctypes.c_void_p * 3)(*[v.ptr for v in states])
        self._chk(self._lib.nk2d_precond_setup_states(self._ctx, ptrs))
        self._precond_ready = True

    def shift_factor(self, t, scale, shifts):
        """factorise scale * J(t, lin_state) - shift * I for every shift (block elimination)"""
        arr = np.ascontiguousarray(shifts, dtype=np.float64)
        self._chk(self._lib.nk2d_shift_factor(self._ctx, float(t), float(scale), len(arr), _dp(arr)))
        self._precond_ready = False

    def shift_solve(self, i, v, out=None):
        out = self.new_vec() if out is None else out
        self._chk(self._lib.nk2d_shift_solve(self._ctx, int(i), v.ptr, out.ptr))
        return out

    def po4_uptake(self, po4):
        """host evaluation of the uptake history variable (phosphorus.py:90-95)"""
        prm = self.phos
        return prm["max_uptake_rate"] * self.light_lim * (po4 / (po4 + prm["po4_halfsat"]))

    def precond_setup_state(self, po4, time_range=(0.0, YEAR)):
        """state dependent preconditioner (phosphorus): linearise about the given po4 field"""
        from .phosphorus import PhosphorusPrecond

        # the eigenvectors move little between Newton iterations: start from the last ones
        last = getattr(self, "_state_precond", None)
        self._state_precond = PhosphorusPrecond(self, po4, time_range,
                                                start=None if last is None else last.restart)
        return self._state_precond

    def precond_apply(self, v, out=None):
        if self.module_kind == 1:
            if getattr(self, "_state_precond", None) is None:
                raise Nk2dError("phosphorus preconditioner: call precond_setup_state(po4) first")
            return self._state_precond.apply(v, out=out)
        if not self._precond_ready:
            self.precond_setup()
        out = self.new_vec() if out is None else out
        self._chk(self._lib.nk2d_precond_apply(self._ctx, v.ptr, out.ptr))
        return out

    # ---- region-weighted algebra -----------------------------------------------------
    def _reg(self, vals):
        arr = np.ascontiguousarray(np.broadcast_to(np.asarray(vals, dtype=np.float64), (self.nreg,)))
        return arr

    def dot(self, a, b):
        out = np.empty(self.nreg)
        self._chk(self._lib.nk2d_dot(self._ctx, a.ptr, b.ptr, _dp(out)))
        return out

    def axpby(self, a, x, b, y, out=None):
        """out = bcast(a) x + bcast(b) y"""
        out = self.new_vec() if out is None else out
        ca, cb = self._reg(a), self._reg(b)
        self._chk(self._lib.nk2d_axpby(self._ctx, out.ptr, _dp(ca), x.ptr, _dp(cb), y.ptr))
        return out

    def scale(self, x, s, out=None):
        out = self.new_vec() if out is None else out
        cs = self._reg(s)
        self._chk(self._lib.nk2d_scale(self._ctx, out.ptr, x.ptr, _dp(cs)))
        return out

    def diff_scale(self, x, y, s, out=None):
        """out = (x - y) bcast(s)"""
        out = self.new_vec() if out is None else out
        cs = self._reg(s)
        self._chk(self._lib.nk2d_diff_scale(self._ctx, out.ptr, x.ptr, y.ptr, _dp(cs)))
        return out

    def lin_comb(self, vecs, coef, out=None):
        """out = sum_i bcast(coef[i]) vecs[i], accumulated in order; coef (n, nreg)"""
        out = self.new_vec() if out is None else out
        coef = np.ascontiguousarray(coef, dtype=np.float64).reshape(len(vecs), self.nreg)
        ptrs = (ctypes.c_void_p * len(vecs))(*[v.ptr for v in vecs])
        self._chk(self._lib.nk2d_lin_comb(self._ctx, out.ptr, len(vecs), ptrs, _dp(coef)))
        return out

    def mgs(self, w, basis):
        """in-place modified Gram-Schmidt of w against basis; returns h (n, nreg)"""
        h = np.empty((len(basis), self.nreg))
        ptrs = (ctypes.c_void_p * len(basis))(*[v.ptr for v in basis])
        self._chk(self._lib.nk2d_mgs(self._ctx, w.ptr, len(basis), ptrs, _dp(h)))
        return h

    def apply_region_mask(self, v):
        self._chk(self._lib.nk2d_apply_region_mask(self._ctx, v.ptr))
        return v


def iage_engine(grid, device_id=0, **kwargs):
    """engine of the `iage` tracer module (two tracers; surface restoring at
    24/day over 10 m and 100x slower; unit ageing source), iage.py:12-41"""
    rate = 24.0 / 86400.0 * 10.0 / grid.depth.delta[0]
    slow = 0.01
    return ModuleEngine(grid, tc=2, surf_rate=(rate, slow * rate),
                        const_src=1.0 / (365.0 * 86400.0), device_id=device_id, **kwargs)


def forced_engine(grid, modelinfo, device_id=0, **kwargs):
    """engine of a `forced_{suff}` tracer module (one tracer, forced.py:57-153): surface
    restoring none / const / file, source-minus-sink none / const / decay / file (the file source
    optionally with a sink threshold).  The file fields are read and put on the model axes here
    (`forcing.load_forcing`); the device interpolates them in time."""
    from .forcing import load_forcing

    restore_opt = modelinfo["forced_surf_restore_opt"]
    sms_opt = modelinfo["forced_sms_opt"]
    if restore_opt not in ("none", "const", "file"):
        raise ValueError(f"unknown forced_surf_restore_opt={restore_opt}")
    if sms_opt not in ("none", "const", "decay", "file"):
        raise ValueError(f"unknown forced_sms_opt={sms_opt}")
    if restore_opt == "none" and sms_opt != "decay":
        raise ValueError("forced_sms_opt must be decay if forced_surf_restore_opt == none")
    surf_rate, surf_target = 0.0, 0.0
    if restore_opt != "none":
        rate_10m = _eval_number(modelinfo.get("forced_surf_restore_rate_10m", 24.0 / 86400.0))
        surf_rate = 10.0 / grid.depth.delta[0] * rate_10m
    if restore_opt == "const":
        surf_target = _eval_number(modelinfo["forced_surf_restore_const"])
    decay = _eval_number(modelinfo["forced_sms_decay_rate"]) if sms_opt == "decay" else 0.0
    const_src = _eval_number(modelinfo["forced_sms_const"]) if sms_opt == "const" else 0.0
    files = {}
    if restore_opt == "file":
        files["restore_series"] = load_forcing(
            modelinfo["forced_surf_restore_fname"], modelinfo["forced_surf_restore_varname"], [grid.ypos.mid])
    if sms_opt == "file":
        scalef = _eval_number(modelinfo["forced_sms_scalef"]) if "forced_sms_scalef" in modelinfo else 1.0
        files["sms_series"] = load_forcing(
            modelinfo["forced_sms_fname"], modelinfo["forced_sms_varname"], [grid.depth.mid, grid.ypos.mid], scalef)
        if "forced_sink_thres" in modelinfo:
            files["sink_thres"] = _eval_number(modelinfo["forced_sink_thres"])
    return ModuleEngine(grid, tc=1, surf_rate=(surf_rate,), surf_target=(surf_target,),
                        decay_rate=(decay,), const_src=const_src, device_id=device_id,
                        module_kind=2 if files else 0, **files, **kwargs)


PHOSPHORUS_PARAM_NAMES = ("po4_halfsat", "max_uptake_rate", "sigma", "dop_remin_rate",
                          "pop_remin_rate", "pop_sink_vel")


def phosphorus_params(overrides=None):
    """parameters of the phosphorus module with the reference's defaults (phosphorus.py:42-58)"""
    params = {
        "po4_halfsat": 0.5,
        "max_uptake_rate": 1.0 / (3.0 * 86400.0),
        "sigma": 0.67,
        "dop_remin_rate": 1.0 / (0.5 * 365.0 * 86400.0),
        "pop_remin_rate": 1.0 / (0.5 * 365.0 * 86400.0),
        "pop_sink_vel": 2.0 / 86400.0,
    }
    for name, val in (overrides or {}).items():
        if name not in params:
            raise ValueError(f"unknown phosphorus parameter {name}")
        params[name] = _eval_number(val)
    return params


def _eval_number(expr):
    """value of a number or of an arithmetic expression string such as "1.0 / (3.0 * 86400.0)"
    (the reference evaluates modelinfo parameters with utils.eval_expr)"""
    if not isinstance(expr, str):
        return float(expr)
    import ast
    import operator

    ops = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul,
           ast.Div: operator.truediv, ast.Pow: operator.pow}

    def walk(node):
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
            return float(node.value)
        if isinstance(node, ast.BinOp) and type(node.op) in ops:
            return ops[type(node.op)](walk(node.left), walk(node.right))
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            val = walk(node.operand)
            return -val if isinstance(node.op, ast.USub) else val
        raise ValueError(f"unsupported expression {expr!r}")

    return walk(ast.parse(expr, mode="eval").body)


def phosphorus_light_lim(grid):
    """light limitation: e-folding depth 25 m, Gaussian in ypos (phosphorus.py:26-29)"""
    return np.outer(np.exp((-1.0 / 25.0) * grid.depth.mid),
                    np.exp(-1.0 * ((grid.ypos.mid - 2.5e6) / 1.5e6) ** 2))


def phosphorus_engine(grid, params=None, device_id=0, **kwargs):
    """engine of the `phosphorus` tracer module (po4, dop, pop; phosphorus.py:17-172):
    nonlinear uptake couples the three tracers in every cell"""
    prm = phosphorus_params(params)
    return ModuleEngine(grid, tc=3, device_id=device_id, module_kind=1,
                        phos_params=[prm[name] for name in PHOSPHORUS_PARAM_NAMES],
                        light_lim=phosphorus_light_lim(grid), **kwargs)
