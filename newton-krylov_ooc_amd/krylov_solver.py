"""GMRES inner solver with an HBM-resident Krylov space.

Drop-in for the reference's `KrylovSolver` (`nk_ooc/krylov_solver.py:13-181`): same
constructor and `solve(res_fname, fcn)` surface, same mathematics (left-preconditioned
GMRES, Saad alg. 9.4, zero initial guess, no restart, an independent Hessenberg per
(tracer module, region), coefficients from `np.linalg.lstsq`), same stopping rule, same
checkpoint trail: `Krylov_state.json` (`beta`, `h_mat`, step strings), one NetCDF3 file per
vector (`precond_fcn_00`, `basis_NN`, `w_raw_NN`, `w_NN`, `perturb_fcn_w_raw_NN`,
`krylov_res_NN`), `Krylov_stats.nc`, and the `beta` / `KrylovCoeff` / `precond_resid` log lines.

Where it differs is the data flow.  The reference is out-of-core: every use of a basis
vector re-opens its file (j+1 reads in Gram-Schmidt, j+1 and j+2 more in the two
`lin_comb`s of each iteration).  Here the Arnoldi vectors V_0..V_j and the preconditioned
products W_0..W_j are kept as device-resident `ModelState`s for the life of the solve;
files are only written (at the reference's points and in the reference's order -- on the
checkpoint trail's writer thread where the driver switched that on, `trail.py`), and read back
only when a solve is resumed from its checkpoint.  Each iteration is then

    w_raw = J v_j        one perturbed forward year per tracer module (nk2d_comp_fcn)
    w     = M^-1 w_raw   streamed block-Thomas apply             (nk2d_precond_apply)
    h     = MGS(w; V)    j+1 fused dot/axpy pairs, no host sync between them (nk2d_mgs)
    x_j, r_j             two single-launch linear combinations    (nk2d_lin_comb)
"""

import logging

import numpy as np

from . import trail
from .solver_base import SolverBase


def least_squares_coeffs(beta, hess):
    """c = argmin || beta e_1 - H c ||_2 for every (tracer module, region).
    hess has shape [ntm, j+2, j+1, nreg] (the reference's `h_mat` layout)."""
    ntm, nrow, ncol, nreg = hess.shape
    coeff = np.zeros((ntm, ncol, nreg))
    for im in range(ntm):
        for ir in range(nreg):
            target = np.zeros(nrow)
            target[0] = beta[im, ir]
            coeff[im, :, ir] = np.linalg.lstsq(hess[im, :, :, ir], target, rcond=None)[0]
    return coeff


# name used by the reference's callers / tests
comp_krylov_basis_coeffs = least_squares_coeffs


class KrylovSolver(SolverBase):
    """approximate solve of  J dx = -fcn  with J the Jacobian of comp_fcn at `iterate`"""

    STATS_VARS = {
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

    def __init__(self, iterate, solverinfo, resume, rewind, hist_fname):
        super().__init__("Krylov", solverinfo, iterate.model_config_obj.region_cnt, resume, rewind)
        self._iterate = iterate
        self._state_cls = type(iterate)
        self._tag = f"{type(self).__module__}.{type(self).__name__}"
        self._V = {}   # Arnoldi vectors, resident
        self._W = {}   # preconditioned Jacobian-vector products (before orthogonalisation)
        self._r0 = None  # M^-1 fcn
        self._def_solver_stats_vars(self.gen_stats_vars_metadata(), iterate.tracer_modules)
        iterate.gen_precond_jacobian(hist_fname, precond_fname=self._fname("precond", iteration=0),
                                     solver_state=self._solver_state)
        self.max_iter = (int(self._solverinfo["krylov_max_iter"])
                         if "krylov_max_iter" in self._solverinfo else None)

    @classmethod
    def gen_stats_vars_metadata(cls):
        return {key: dict(val, attrs=dict(val["attrs"])) for key, val in cls.STATS_VARS.items()}

    # ---- resident Krylov space (files are the fallback after a resume) ----------------
    def _resident(self, store, quantity, index):
        if index not in store:
            store[index] = self._state_cls(self._fname(quantity, index))
        return store[index]

    def _basis(self, index):
        try:
            return self._resident(self._V, "basis", index)
        except FileNotFoundError:
            if index < 1:
                raise
            return self._rebuild_basis(index)

    def _rebuild_basis(self, index):
        """a resumed run whose predecessor died between `inc_iteration` and the dump of the next Arnoldi vector (the
        reference's own window, krylov_solver.py:167-181: the vector is written only when the loop goes on): v_index again
        from what IS on disk -- the preconditioned product `w_{index-1}` orthogonalised against v_0 .. v_{index-1} and
        scaled by the last sub-diagonal entry of the saved Hessenberg matrix --, the same operations on the same values"""
        logging.getLogger(__name__).warning("basis vector %d is not on disk: rebuilt from w_%02d and the saved Hessenberg matrix",
                                            index, index - 1)
        hess = self._solver_state.get_value_saved_state("h_mat")
        if hess.shape[2] != index:
            raise FileNotFoundError(f"{self._fname('basis', index)}: not on disk, and the saved Hessenberg matrix is not "
                                    f"that of iteration {index - 1}")
        vec = self._state_cls(self._fname("w", index - 1))
        vec.mgs_against([self._basis(i) for i in range(index)])
        vec /= hess[:, -1, -1, :]
        self._V[index] = vec.dump(self._fname("basis", index), f"{self._tag}._rebuild_basis")
        return self._V[index]

    def _prod(self, index):
        return self._resident(self._W, "w", index)

    def _precond_fcn(self):
        if self._r0 is None:
            self._r0 = self._state_cls(self._fname("precond_fcn", 0))
        return self._r0

    # ---- pieces of one solve ---------------------------------------------------------------
    def converged(self, beta, precond_resid_norm):
        """elementwise over (tracer module, region); the caller requires all of them"""
        enough = self.get_iteration() >= self._get_min_iter()
        return enough & (precond_resid_norm < self._get_rel_tol() * beta)

    def _start(self, fcn):
        """first Arnoldi vector v_0 = -M^-1 fcn / beta (once per solve, step-logged)"""
        state = self._solver_state
        if state.step_logged("KrylovSolver._solve0", per_iteration=False):
            return
        r0 = fcn.apply_precond_jacobian(self._fname("precond", 0), self._fname("precond_fcn"), state)
        beta = r0.norm()
        fcn.log_vals("beta", beta)
        self._put_solver_stats_vars_iteration_independent(precond_rhs_norm=beta)
        v0 = (-r0 / beta).dump(self._fname("basis"), f"{self._tag}._solve0")
        self._r0, self._V[0] = r0, v0
        state.set_value_saved_state("beta", beta)
        state.log_step("KrylovSolver._solve0", per_iteration=False)

    def _hessenberg(self, j, ntm, nreg):
        hess = np.zeros((ntm, j + 2, j + 1, nreg))
        if j > 0:
            # (the leading block: a resumed run whose predecessor died between saving this iteration's matrix and
            # `inc_iteration` finds the matrix of iteration j already there -- its first j columns are those of iteration j - 1)
            hess[:, :-1, :-1, :] = self._solver_state.get_value_saved_state("h_mat")[:, :j + 1, :j, :]
        return hess

    def _arnoldi_step(self, fcn, j):
        """extend the space by one vector; returns (h_mat, normalised new direction)"""
        state = self._solver_state
        w_raw = self._iterate.comp_jacobian_fcn_state_prod(fcn, self._basis(j), self._fname("w_raw"), state)
        w = w_raw.apply_precond_jacobian(self._fname("precond", 0), self._fname("w"), state)
        self._W[j] = w.copy()  # the residual below needs w before orthogonalisation
        hess = self._hessenberg(j, len(fcn.tracer_modules), fcn.model_config_obj.region_cnt)
        hess[:, :-1, -1, :] = w.mgs_against([self._basis(i) for i in range(j + 1)])
        hess[:, -1, -1, :] = w.norm()
        w /= hess[:, -1, -1, :]
        state.set_value_saved_state("h_mat", hess)
        return hess, w

    def solve(self, res_fname, fcn):
        logger = logging.getLogger(__name__)
        state = self._solver_state
        self._start(fcn)
        caller = f"{self._tag}.solve"
        while True:
            j = self.get_iteration()
            hess, v_next = self._arnoldi_step(fcn, j)
            beta = state.get_value_saved_state("beta")
            coeff = least_squares_coeffs(beta, hess)
            self._iterate.log_vals("KrylovCoeff", coeff)

            # iterate x_j = V c and its preconditioned residual  W c + M^-1 fcn
            approx = self._state_cls.lin_comb_of(coeff, [self._basis(i) for i in range(j + 1)])
            approx.dump(self._fname("krylov_res", j), caller)
            resid = self._state_cls.lin_comb_of(coeff, [self._prod(i) for i in range(j + 1)])
            resid += self._precond_fcn()
            resid_norm = resid.norm()
            self._iterate.log_vals("precond_resid", resid_norm)
            self._put_solver_stats_vars(precond_resid_norm=resid_norm)

            state.inc_iteration()
            if self.converged(beta, resid_norm).all():
                logger.info("Krylov convergence criterion satisfied")
                break
            if self.max_iter is not None and self.get_iteration() >= self.max_iter:
                logger.info("Krylov iteration limit reached")
                break
            self._V[j + 1] = v_next.dump(self._fname("basis"), caller)
        approx.dump(res_fname, caller)
        # the solve's trail is complete on disk when it returns (inside a solve the files follow on the trail's writer
        # thread while the next perturbed year runs: trail.py)
        trail.flush()
        return approx
