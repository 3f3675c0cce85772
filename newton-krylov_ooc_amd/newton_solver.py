"""Newton outer loop around the GPU Krylov solver.

Counterpart of the reference's `NewtonSolver` (`nk_ooc/newton_solver.py:14-334`) -- the caller
of the hot path (`SURVEY.md` section 8f, row f1).  Same constructor / `converged()` / `step()` /
`log()` surface, same sequence of checkpointed actions and file names
(`iterate_NN`, `fcn_NN`, `hist_NN`, `increment_NN`, `prov_Armijo_AA_NN`, `prov_fcn_Armijo_AA_NN`,
`prov_fp_FF_NN`, ...) and the same `Newton_state.json` step strings, so that a run can be
resumed by either implementation:

    increment  = Krylov solve of  J dx = -F(x)          (krylov_NN/ work directory)
    limiter    : scale dx so that x + dx respects the tracer bounds, per region
    Armijo     : halve the factor per (module, region) until |F(x + a dx)| <= (1 - 1e-4 a) |F(x)|
    fixed point: `post_newton_fp_iter` iterations x <- x + F(x) after the Newton update

Every F evaluation is one forward model year per tracer module on its GPU, here with the
61-sample history file the reference writes (dense output of the device integrator).
"""

import logging
import os

import numpy as np

from .krylov_solver import KrylovSolver
from .solver_base import SolverBase

ARMIJO_ALPHA = 1.0e-4
ARMIJO_MAX_HALVINGS = 10


def _model_state_var(state_name):
    return {
        "category": "model_state",
        "dimensions": ("iteration", "region"),
        "attrs": {"long_name": "{method} of {tracer_module_name} Newton " + state_name,
                  "units": "{tracer_module_units}"},
    }


def _factor_var(purpose):
    return {
        "category": "per_tracer_module",
        "dimensions": ("iteration", "region"),
        "attrs": {"long_name": "factor applied to {tracer_module_name} Newton increment to " + purpose,
                  "units": "1"},
    }


class NewtonSolver(SolverBase):
    """Newton's method for F(x) = 0 with F = one model year minus identity"""

    krylov_solver_class = KrylovSolver

    def __init__(self, model_state_class, solverinfo, resume, rewind):
        super().__init__("Newton", solverinfo, model_state_class.model_config_obj.region_cnt,
                         resume, rewind)
        self._tag = f"{type(self).__module__}.{type(self).__name__}"
        self._krylov_info = dict(solverinfo)
        state = self._solver_state
        first = "Newton iterate 0 written"
        if state.step_logged(first, per_iteration=False):
            self._iterate = model_state_class(self._fname("iterate"))
        else:
            self._iterate = model_state_class(solverinfo["init_iterate_fname"])
            self._iterate.copy_real_tracers_to_shadow_tracers().dump(
                self._fname("iterate"), f"{self._tag}.__init__")
            state.log_step(first, per_iteration=False)
        self._def_solver_stats_vars(self.gen_stats_vars_metadata(), self._iterate.tracer_modules)
        self._fcn = self._iterate.comp_fcn(self._fname("fcn"), state, self._fname("hist"))
        self._put_solver_stats_vars(iterate=self._iterate, fcn=self._fcn)
        hist = self._fname("hist")
        self._iterate.def_stats_vars(self._stats_file, hist, solver_state=state)
        self._iterate.put_stats_vars_iteration_invariant(self._stats_file, hist, solver_state=state)
        self._iterate.put_stats_vars(self._stats_file, hist, solver_state=state)

    @staticmethod
    def gen_stats_vars_metadata():
        meta = {name: _model_state_var(name) for name in ("iterate", "fcn", "increment")}
        meta["increment_scalef"] = _factor_var("satisfy bounds")
        meta["Armijo_factor"] = _factor_var("satisfy Armijo condition")
        meta["Krylov_iterations"] = {
            "category": "tracer_module_independent",
            "datatype": "i4",
            "dimensions": ("iteration",),
            "attrs": {"long_name": "number of iterations in Krylov solver", "units": "1"},
        }
        return meta

    # ---- reporting -----------------------------------------------------------------------
    def log(self, iterate=None, fcn=None, msg=None):
        head = f"iteration={self.get_iteration():02}" + ("" if msg is None else f",{msg}")
        (self._iterate if iterate is None else iterate).log(f"{head},iterate")
        (self._fcn if fcn is None else fcn).log(f"{head},fcn")

    def converged(self):
        """|F| < newton_rel_tol |x| for every (module, region)"""
        enough = self.get_iteration() >= self._get_min_iter()
        return enough & (self._fcn.norm() < self._get_rel_tol() * self._iterate.norm())

    # ---- the Newton direction ----------------------------------------------------------------
    def _increment(self):
        state = self._solver_state
        done = "_comp_increment complete"
        if state.step_logged(done):
            return type(self._iterate)(self._fname("increment"))
        self._krylov_info["krylov_workdir"] = os.path.join(
            self._get_workdir(), f"krylov_{self.get_iteration():02}")
        mark = "KrylovSolver instantiated"
        rewind = state.step_was_rewound(mark)
        resume = rewind or state.step_logged(mark)
        if not resume:
            self.log()
        krylov = self.krylov_solver_class(self._iterate, self._krylov_info, resume, rewind,
                                          self._fname("hist"))
        state.log_step(mark)
        increment = krylov.solve(self._fname("increment"), self._fcn)
        self._put_solver_stats_vars(Krylov_iterations=krylov.get_iteration(), increment=increment)
        state.log_step(done)
        increment.log(f"Newton increment {self.get_iteration():02}")
        return increment

    # ---- line search -----------------------------------------------------------------------------
    def _line_search(self, increment):
        """Armijo back-tracking per (module, region); returns (candidate, F(candidate))"""
        logger = logging.getLogger(__name__)
        state = self._solver_state
        if not state.step_logged("NewtonSolver._armijo_init"):
            state.set_value_saved_state(key="armijo_ind", value=0)
            state.set_value_saved_state(key="armijo_factor",
                                        value=np.where(self.converged(), 0.0, 1.0))
            state.log_step("NewtonSolver._armijo_init")
        ind = state.get_value_saved_state(key="armijo_ind")
        factor = state.get_value_saved_state(key="armijo_factor")
        done = "_comp_next_iterate complete"
        cls = type(self._iterate)
        if state.step_logged(done):
            return (cls(self._fname(f"prov_Armijo_{ind:02}")),
                    cls(self._fname(f"prov_fcn_Armijo_{ind:02}")))
        caller = f"{self._tag}._comp_next_iterate"
        fcn_norm = self._fcn.norm()
        while True:
            cand = self._iterate + factor * increment
            cand.dump(self._fname(f"prov_Armijo_{ind:02}"), caller)
            cand_fcn = cand.comp_fcn(self._fname(f"prov_fcn_Armijo_{ind:02}"), state,
                                     self._fname(f"prov_hist_Armijo_{ind:02}"))
            if ind > 0:  # only the latest line-search history is kept
                stale = self._fname(f"prov_hist_Armijo_{(ind - 1):02}")
                if os.path.exists(stale):
                    os.remove(stale)
            logger.info("Armijo_ind=%d", ind)
            cand_norm = cand_fcn.norm()
            increment.log_vals(["ArmijoFactor", "fcn_norm", "prov_fcn_norm"],
                               np.stack((factor, fcn_norm, cand_norm)))
            ok = (factor == 0.0) | (cand_norm <= (1.0 - ARMIJO_ALPHA * factor) * fcn_norm)
            if ok.all():
                logger.info("Armijo condition satisfied")
                state.log_step(done)
                self._put_solver_stats_vars(Armijo_factor=factor)
                return cand, cand_fcn
            logger.info("Armijo condition not satisfied")
            factor = np.where(ok, factor, 0.5 * factor)
            ind += 1
            state.set_value_saved_state(key="armijo_ind", value=ind)
            state.set_value_saved_state(key="armijo_factor", value=factor)
            if ind > ARMIJO_MAX_HALVINGS:
                raise RuntimeError("Armijo_ind exceeds limit")

    # ---- one Newton iteration ------------------------------------------------------------------------
    def step(self):
        state = self._solver_state
        info = self._solverinfo
        if self.get_iteration() >= int(info["newton_max_iter"]):
            self.log()
            raise RuntimeError("number of maximum Newton iterations exceeded")
        caller = f"{self._tag}.step"
        cls = type(self._iterate)
        n_fp = int(info["post_newton_fp_iter"])
        started = "fp iterations started"
        if not state.step_logged(started):
            increment = self._increment()
            self._put_solver_stats_vars(increment_scalef=increment.apply_limiter(self._iterate))
            cand, cand_fcn = self._line_search(increment)
            fp_iter = 0
            state.set_value_saved_state(key="fp_iter", value=fp_iter)
            cand.copy_shadow_tracers_to_real_tracers()
            cand.dump(self._fname(f"prov_fp_{fp_iter:02}"), caller)
            ind = state.get_value_saved_state(key="armijo_ind")
            line_hist = self._fname(f"prov_hist_Armijo_{ind:02}")
            if cand.shadow_tracers_on():
                cand_fcn = cand.comp_fcn(self._fname(f"prov_fcn_fp_{fp_iter:02}"), state,
                                         self._fname(f"prov_hist_fp_{fp_iter:02}"))
                if os.path.exists(line_hist):
                    os.remove(line_hist)
            else:
                # the accepted line-search evaluation IS the first fixed-point evaluation
                cand_fcn.dump(self._fname(f"prov_fcn_fp_{fp_iter:02}"), caller)
                if os.path.exists(line_hist):
                    os.rename(line_hist, self._fname(f"prov_hist_fp_{fp_iter:02}"))
            state.log_step(started)
        else:
            fp_iter = state.get_value_saved_state(key="fp_iter")
            cand = cls(self._fname(f"prov_fp_{fp_iter:02}"))
            cand_fcn = cls(self._fname(f"prov_fcn_fp_{fp_iter:02}"))

        while fp_iter < n_fp:
            mark = f"prov updated for fp iteration {fp_iter:02}"
            if not state.step_logged(mark):
                if fp_iter == 0:
                    self.log(cand, cand_fcn, "pre-fp_iter")
                cand += cand_fcn
                cand.copy_shadow_tracers_to_real_tracers()
                cand.dump(self._fname(f"prov_fp_{(fp_iter + 1):02}"), caller)
                state.log_step(mark)
            else:
                cand = cls(self._fname(f"prov_fp_{(fp_iter + 1):02}"))
            if fp_iter + 1 < n_fp:
                res_fname = self._fname(f"prov_fcn_fp_{(fp_iter + 1):02}")
                hist_fname = self._fname(f"prov_hist_fp_{(fp_iter + 1):02}")
            else:
                state.inc_iteration()
                cand.dump(self._fname("iterate"), caller)
                res_fname, hist_fname = self._fname("fcn"), self._fname("hist")
            cand_fcn = cand.comp_fcn(res_fname, state, hist_fname)
            fp_iter += 1
            state.set_value_saved_state(key="fp_iter", value=fp_iter)
            self.log(cand, cand_fcn, f"fp_iter={fp_iter:02}")

        self._iterate, self._fcn = cand, cand_fcn
        self._put_solver_stats_vars(iterate=self._iterate, fcn=self._fcn)
        self._iterate.put_stats_vars(self._stats_file, hist_fname=self._fname("hist"),
                                     solver_state=state)

    @property
    def iterate(self):
        return self._iterate

    @property
    def fcn(self):
        return self._fcn
