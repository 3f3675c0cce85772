"""Left-preconditioned GMRES on the device-resident model state.

Mirror of the reference's `KrylovSolver` (`nk_ooc/krylov_solver.py:13-181`): same
constructor and `solve(res_fname, fcn)` surface, same algorithm (Saad alg. 9.4, x0 = 0,
no restart, one Hessenberg per (tracer module, region), least squares by
`np.linalg.lstsq`), same checkpoint trail (`Krylov_state.json` with `beta` and `h_mat`,
step strings, `<quantity>_NN.nc` files) and the same log lines.  The vectors never
leave HBM between iterations: re-opens of `basis_NN` / `w_NN` / `precond_fcn_00` are
served from device snapshots, the j+1 Gram-Schmidt projections and each `lin_comb` run
as fused launches (`nk2d_mgs`, `nk2d_lin_comb`).
"""

import logging

import numpy as np

from . import model_state
from .solver_base import SolverBase
from .solver_state import action_step_log_wrap


class KrylovSolver(SolverBase):
    """approximate solution of  A x = -fcn,  A = Jacobian of comp_fcn at iterate"""

    def __init__(self, iterate, solverinfo, resume, rewind, hist_fname):
        super().__init__("Krylov", solverinfo, iterate.model_config_obj.region_cnt, resume, rewind)
        self._iterate = iterate
        self._def_solver_stats_vars(self.gen_stats_vars_metadata(), self._iterate.tracer_modules)
        iterate.gen_precond_jacobian(
            hist_fname, precond_fname=self._fname("precond", iteration=0),
            solver_state=self._solver_state)
        self.max_iter = (int(self._solverinfo["krylov_max_iter"])
                         if "krylov_max_iter" in self._solverinfo else None)

    @staticmethod
    def gen_stats_vars_metadata():
        return {
            "precond_rhs_norm": {
                "category": "per_tracer_module",
                "dimensions": ("region",),
                "attrs": {"long_name": "norm of {tracer_module_name} preconditioned rhs",
                          "units": "{tracer_module_units}"},
            },
            "precond_resid_norm": {
                "category": "per_tracer_module",
                "dimensions": ("iteration", "region"),
                "attrs": {"long_name": "norm of {tracer_module_name} preconditioned residual",
                          "units": "{tracer_module_units}"},
            },
        }

    def converged(self, beta, precond_resid_norm):
        rel_tol = self._get_rel_tol()
        return (self.get_iteration() >= self._get_min_iter()) & (precond_resid_norm < rel_tol * beta)

    @action_step_log_wrap(step="KrylovSolver._solve0", per_iteration=False)
    def _solve0(self, fcn, solver_state):
        """r0 = M^-1 (rhs - A x0) = -M^-1 fcn; v0 = r0 / beta"""
        precond_fcn = fcn.apply_precond_jacobian(
            self._fname("precond", 0), self._fname("precond_fcn"), self._solver_state)
        beta = precond_fcn.norm()
        fcn.log_vals("beta", beta)
        self._put_solver_stats_vars_iteration_independent(precond_rhs_norm=beta)
        caller = f"{type(self).__module__}.{type(self).__name__}._solve0"
        (-precond_fcn / beta).dump(self._fname("basis"), caller)
        self._solver_state.set_value_saved_state("beta", beta)

    def solve(self, res_fname, fcn):
        logger = logging.getLogger(__name__)
        self._solve0(fcn, solver_state=self._solver_state)
        caller = f"{type(self).__module__}.{type(self).__name__}.solve"
        state_type = type(self._iterate)
        while True:
            j_val = self.get_iteration()
            h_mat = np.zeros((len(fcn.tracer_modules), j_val + 2, j_val + 1,
                              fcn.model_config_obj.region_cnt))
            if j_val > 0:
                h_mat[:, :-1, :-1, :] = self._solver_state.get_value_saved_state("h_mat")
            basis_j = state_type(self._fname("basis"))
            w_raw = self._iterate.comp_jacobian_fcn_state_prod(
                fcn, basis_j, self._fname("w_raw"), self._solver_state)
            w_j = w_raw.apply_precond_jacobian(
                self._fname("precond", 0), self._fname("w"), self._solver_state)
            h_mat[:, :-1, -1, :] = w_j.mod_gram_schmidt(j_val + 1, self._fname, "basis")
            h_mat[:, -1, -1, :] = w_j.norm()
            w_j /= h_mat[:, -1, -1, :]
            self._solver_state.set_value_saved_state("h_mat", h_mat)

            beta = self._solver_state.get_value_saved_state("beta")
            coeff = comp_krylov_basis_coeffs(beta, h_mat)
            self._iterate.log_vals("KrylovCoeff", coeff)

            res = model_state.lin_comb(state_type, coeff, self._fname, "basis")
            res.dump(self._fname("krylov_res", j_val), caller)

            precond_resid = model_state.lin_comb(state_type, coeff, self._fname, "w")
            precond_resid += state_type(self._fname("precond_fcn", 0))
            precond_resid_norm = precond_resid.norm()
            self._iterate.log_vals("precond_resid", precond_resid_norm)
            self._put_solver_stats_vars(precond_resid_norm=precond_resid_norm)

            self._solver_state.inc_iteration()

            if self.converged(beta, precond_resid_norm).all():
                logger.info("Krylov convergence criterion satisfied")
                break
            if self.max_iter is not None and self.get_iteration() >= self.max_iter:
                logger.info("Krylov iteration limit reached")
                break

            w_j.dump(self._fname("basis"), caller)

        return res.dump(res_fname, caller)


def comp_krylov_basis_coeffs(beta, h_mat):
    """argmin_c || beta e_1 - H c ||_2 for every (tracer module, region)"""
    ntm, nrow, ncol, nreg = h_mat.shape
    coeff = np.zeros((ntm, ncol, nreg))
    rhs = np.zeros(nrow)
    for module_ind in range(ntm):
        for region_ind in range(nreg):
            rhs[0] = beta[module_ind, region_ind]
            coeff[module_ind, :, region_ind] = np.linalg.lstsq(
                h_mat[module_ind, :, :, region_ind], rhs, rcond=None)[0]
    return coeff
