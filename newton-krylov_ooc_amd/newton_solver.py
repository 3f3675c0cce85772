"""Newton outer loop around the GPU Krylov solver.

Counterpart of the reference's `NewtonSolver` (`nk_ooc/newton_solver.py:14-334`) -- the caller
of the hot path (`SURVEY.md` section 8f, row f1).  Same constructor / `converged()` / `step()` /
`log()` surface, the same sequence of checkpointed actions, file names and `Newton_state.json`
step strings, so that a run can be resumed by either implementation:

    increment  = Krylov solve of  J dx = -F(x)          (krylov_NN/ work directory)
    limiter    : scale dx so that x + dx respects the tracer bounds, per region
    Armijo     : halve the factor per (module, region) until |F(x + a dx)| <= (1 - 1e-4 a) |F(x)|
    fixed point: `post_newton_fp_iter` iterations x <- x + F(x) after the Newton update

Every F evaluation is one forward model year per tracer module on its GPU, here with the
61-sample history file the reference writes (dense output of the device integrator).

Layout of this file: `_Trail` names the files of the current Newton iteration, `_Ledger` wraps the
step log / saved values of `Newton_state.json`; `NewtonSolver.step` is then the three phases
`_newton_update` (increment, limiter, line search), `_enter_fixed_point` and `_fixed_point_tail`.
"""

import logging
import os

import numpy as np

from .krylov_solver import KrylovSolver
from .solver_base import SolverBase

ARMIJO_ALPHA = 1.0e-4
ARMIJO_MAX_HALVINGS = 10

# strings of the step log (they are part of the on-disk format shared with the reference)
MARK_ITERATE_0 = "Newton iterate 0 written"
MARK_KRYLOV_MADE = "KrylovSolver instantiated"
MARK_INCREMENT_DONE = "_comp_increment complete"
MARK_ARMIJO_INIT = "NewtonSolver._armijo_init"
MARK_LINE_SEARCH_DONE = "_comp_next_iterate complete"
MARK_FP_STARTED = "fp iterations started"


def _mark_fp_updated(k):
    return f"prov updated for fp iteration {k:02}"


class _Trail:
    """file names of the quantities of the current Newton iteration"""

    def __init__(self, name_fcn):
        self._name = name_fcn

    def __call__(self, quantity, index=None):
        return self._name(quantity if index is None else f"{quantity}_{index:02}")

    def line_search(self, ind):
        return (self("prov_Armijo", ind), self("prov_fcn_Armijo", ind), self("prov_hist_Armijo", ind))

    def fixed_point(self, k):
        return (self("prov_fp", k), self("prov_fcn_fp", k), self("prov_hist_fp", k))


class _Ledger:
    """step log and saved values of the solver state file"""

    def __init__(self, solver_state):
        self._st = solver_state

    def done(self, step, per_iteration=True):
        return self._st.step_logged(step, per_iteration=per_iteration)

    def mark(self, step, per_iteration=True):
        self._st.log_step(step, per_iteration=per_iteration)

    def rewound(self, step):
        return self._st.step_was_rewound(step)

    def recall(self, key):
        return self._st.get_value_saved_state(key=key)

    def remember(self, key, value):
        self._st.set_value_saved_state(key=key, value=value)


def _drop(fname, state_cls=None):
    """remove a history file; one that is still being written (the state class's background writer) first completes"""
    writer = getattr(state_cls, "hist_writer", None)
    if writer is not None:
        writer.remove(fname)
    elif os.path.exists(fname):
        os.remove(fname)


def _rename(src, dst, state_cls=None):
    writer = getattr(state_cls, "hist_writer", None)
    if writer is not None:
        writer.rename(src, dst)         # (behind the write of src, if that is still on its way)
    elif os.path.exists(src):
        os.rename(src, dst)


def _stats_entry(category, long_name, units, **extra):
    entry = {"category": category, "dimensions": ("iteration", "region"),
             "attrs": {"long_name": long_name, "units": units}}
    entry.update(extra)
    return entry


class NewtonSolver(SolverBase):
    """Newton's method for F(x) = 0 with F = one model year minus identity"""

    krylov_solver_class = KrylovSolver

    def __init__(self, model_state_class, solverinfo, resume, rewind):
        region_cnt = model_state_class.model_config_obj.region_cnt
        super().__init__("Newton", solverinfo, region_cnt, resume, rewind)
        self._who = f"{type(self).__module__}.{type(self).__name__}"
        self._krylov_info = dict(solverinfo)
        self._files = _Trail(self._fname)
        self._book = _Ledger(self._solver_state)
        files, book = self._files, self._book
        if book.done(MARK_ITERATE_0, per_iteration=False):
            start = model_state_class(files("iterate"))
        else:
            start = model_state_class(solverinfo["init_iterate_fname"])
            start.copy_real_tracers_to_shadow_tracers().dump(files("iterate"), f"{self._who}.__init__")
            book.mark(MARK_ITERATE_0, per_iteration=False)
        self._iterate = start
        self._def_solver_stats_vars(self.gen_stats_vars_metadata(), start.tracer_modules)
        self._fcn = start.comp_fcn(files("fcn"), self._solver_state, files("hist"))
        self._record(iterate=self._iterate, fcn=self._fcn)
        hist = files("hist")
        start.def_stats_vars(self._stats_file, hist, solver_state=self._solver_state)
        start.put_stats_vars_iteration_invariant(self._stats_file, hist, solver_state=self._solver_state)
        start.put_stats_vars(self._stats_file, hist, solver_state=self._solver_state)

    # ---- stats file ----------------------------------------------------------------------------
    @staticmethod
    def gen_stats_vars_metadata():
        meta = {}
        for what in ("iterate", "fcn", "increment"):
            meta[what] = _stats_entry("model_state", "{method} of {tracer_module_name} Newton " + what,
                                      "{tracer_module_units}")
        for name, purpose in (("increment_scalef", "satisfy bounds"),
                              ("Armijo_factor", "satisfy Armijo condition")):
            meta[name] = _stats_entry(
                "per_tracer_module", "factor applied to {tracer_module_name} Newton increment to " + purpose, "1")
        meta["Krylov_iterations"] = _stats_entry(
            "tracer_module_independent", "number of iterations in Krylov solver", "1", datatype="i4",
            dimensions=("iteration",))
        return meta

    def _record(self, **values):
        self._put_solver_stats_vars(**values)

    # ---- reporting --------------------------------------------------------------------------------
    def log(self, iterate=None, fcn=None, msg=None):
        prefix = f"iteration={self.get_iteration():02}"
        if msg is not None:
            prefix = f"{prefix},{msg}"
        (iterate if iterate is not None else self._iterate).log(f"{prefix},iterate")
        (fcn if fcn is not None else self._fcn).log(f"{prefix},fcn")

    def converged(self):
        """|F| < newton_rel_tol |x| for every (module, region), once min_iter iterations ran"""
        small = self._fcn.norm() < self._get_rel_tol() * self._iterate.norm()
        return (self.get_iteration() >= self._get_min_iter()) & small

    @property
    def iterate(self):
        return self._iterate

    @property
    def fcn(self):
        return self._fcn

    # ---- phase 1: Newton direction, limiter, Armijo line search ------------------------------------
    def _solve_for_increment(self):
        files, book = self._files, self._book
        state_cls = type(self._iterate)
        if book.done(MARK_INCREMENT_DONE):
            return state_cls(files("increment"))
        iteration = self.get_iteration()
        self._krylov_info["krylov_workdir"] = os.path.join(self._get_workdir(), f"krylov_{iteration:02}")
        rewind = book.rewound(MARK_KRYLOV_MADE)
        resume = rewind or book.done(MARK_KRYLOV_MADE)
        if not resume:
            self.log()
        krylov = self.krylov_solver_class(self._iterate, self._krylov_info, resume, rewind, files("hist"))
        book.mark(MARK_KRYLOV_MADE)
        increment = krylov.solve(files("increment"), self._fcn)
        self._record(Krylov_iterations=krylov.get_iteration(), increment=increment)
        book.mark(MARK_INCREMENT_DONE)
        increment.log(f"Newton increment {iteration:02}")
        return increment

    def _line_search(self, increment):
        """Armijo back-tracking per (module, region); returns (candidate, F(candidate))"""
        logger = logging.getLogger(__name__)
        files, book = self._files, self._book
        state_cls = type(self._iterate)
        if not book.done(MARK_ARMIJO_INIT):
            book.remember("armijo_ind", 0)
            book.remember("armijo_factor", np.where(self.converged(), 0.0, 1.0))
            book.mark(MARK_ARMIJO_INIT)
        ind, factor = book.recall("armijo_ind"), book.recall("armijo_factor")
        if book.done(MARK_LINE_SEARCH_DONE):
            cand_fname, fcn_fname, _ = files.line_search(ind)
            return state_cls(cand_fname), state_cls(fcn_fname)
        fcn_norm = self._fcn.norm()
        while ind <= ARMIJO_MAX_HALVINGS:
            cand_fname, fcn_fname, hist_fname = files.line_search(ind)
            cand = self._iterate + factor * increment
            cand.dump(cand_fname, f"{self._who}._comp_next_iterate")
            cand_fcn = cand.comp_fcn(fcn_fname, self._solver_state, hist_fname)
            if ind > 0:
                _drop(files.line_search(ind - 1)[2], state_cls)      # only the latest line-search history is kept
            logger.info("Armijo_ind=%d", ind)
            cand_norm = cand_fcn.norm()
            increment.log_vals(["ArmijoFactor", "fcn_norm", "prov_fcn_norm"],
                               np.stack((factor, fcn_norm, cand_norm)))
            accept = (factor == 0.0) | (cand_norm <= (1.0 - ARMIJO_ALPHA * factor) * fcn_norm)
            if accept.all():
                logger.info("Armijo condition satisfied")
                book.mark(MARK_LINE_SEARCH_DONE)
                self._record(Armijo_factor=factor)
                return cand, cand_fcn
            logger.info("Armijo condition not satisfied")
            factor = np.where(accept, factor, 0.5 * factor)
            ind += 1
            book.remember("armijo_ind", ind)
            book.remember("armijo_factor", factor)
        raise RuntimeError("Armijo_ind exceeds limit")

    def _newton_update(self):
        increment = self._solve_for_increment()
        self._record(increment_scalef=increment.apply_limiter(self._iterate))
        return self._line_search(increment)

    # ---- phase 2: hand the accepted candidate to the fixed-point iterations ---------------------------
    def _enter_fixed_point(self, cand, cand_fcn):
        files, book = self._files, self._book
        caller = f"{self._who}.step"
        book.remember("fp_iter", 0)
        cand.copy_shadow_tracers_to_real_tracers()
        fp_fname, fp_fcn_fname, fp_hist_fname = files.fixed_point(0)
        cand.dump(fp_fname, caller)
        line_hist = files.line_search(book.recall("armijo_ind"))[2]
        if cand.shadow_tracers_on():
            cand_fcn = cand.comp_fcn(fp_fcn_fname, self._solver_state, fp_hist_fname)
            _drop(line_hist, type(cand))
        else:
            # the accepted line-search evaluation IS the first fixed-point evaluation
            cand_fcn.dump(fp_fcn_fname, caller)
            _rename(line_hist, fp_hist_fname, type(cand))
        book.mark(MARK_FP_STARTED)
        return cand, cand_fcn

    # ---- phase 3: post-Newton fixed-point iterations; the last one becomes the next iterate ---------------
    def _fixed_point_tail(self, cand, cand_fcn, fp_iter):
        files, book = self._files, self._book
        state_cls = type(self._iterate)
        caller = f"{self._who}.step"
        total = int(self._solverinfo["post_newton_fp_iter"])
        while fp_iter < total:
            nxt = fp_iter + 1
            nxt_fname, nxt_fcn_fname, nxt_hist_fname = files.fixed_point(nxt)
            if book.done(_mark_fp_updated(fp_iter)):
                cand = state_cls(nxt_fname)
            else:
                if fp_iter == 0:
                    self.log(cand, cand_fcn, "pre-fp_iter")
                cand += cand_fcn
                cand.copy_shadow_tracers_to_real_tracers()
                cand.dump(nxt_fname, caller)
                book.mark(_mark_fp_updated(fp_iter))
            if nxt == total:
                self._solver_state.inc_iteration()
                cand.dump(files("iterate"), caller)
                nxt_fcn_fname, nxt_hist_fname = files("fcn"), files("hist")
            cand_fcn = cand.comp_fcn(nxt_fcn_fname, self._solver_state, nxt_hist_fname)
            fp_iter = nxt
            book.remember("fp_iter", fp_iter)
            self.log(cand, cand_fcn, f"fp_iter={fp_iter:02}")
        return cand, cand_fcn

    # ---- one Newton iteration ------------------------------------------------------------------------
    def step(self):
        files, book = self._files, self._book
        if self.get_iteration() >= int(self._solverinfo["newton_max_iter"]):
            self.log()
            raise RuntimeError("number of maximum Newton iterations exceeded")
        if book.done(MARK_FP_STARTED):
            fp_iter = book.recall("fp_iter")
            state_cls = type(self._iterate)
            fp_fname, fp_fcn_fname, _ = files.fixed_point(fp_iter)
            cand, cand_fcn = state_cls(fp_fname), state_cls(fp_fcn_fname)
        else:
            cand, cand_fcn = self._enter_fixed_point(*self._newton_update())
            fp_iter = 0
        self._iterate, self._fcn = self._fixed_point_tail(cand, cand_fcn, fp_iter)
        self._record(iterate=self._iterate, fcn=self._fcn)
        self._iterate.put_stats_vars(self._stats_file, hist_fname=files("hist"), solver_state=self._solver_state)
