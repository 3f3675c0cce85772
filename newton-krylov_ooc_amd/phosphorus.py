"""Preconditioner of the `phosphorus` tracer module on the GPU.

Reference: `phosphorus.apply_precond_jacobian` (`nk_ooc/py_driver_2d/phosphorus.py:197-274`).
With `mat = T J(T/2, po4)` (one backward-Euler step over the whole year, po4 from the end of the
iterate's history, `T` = one year) the reference computes, for every application,

    e_vals, e_vects = eigs(mat, k=5, sigma=0.0)     # null vector, second smallest eigenvalue
    shift = 0.5 * e_vals[1].real
    x = 2 (mat - shift/2 I)^-1 v - (mat - shift I)^-1 v        # extrapolation to zero shift
    x -= mean(x) * null_vect / mean(null_vect)                  # total P is conserved
    result = x - v

Here the two shifted matrices are factorised ONCE per preconditioner (block elimination over
the ypos columns with the three tracers of a column in one block, `nk2d_shift_factor`) and every
application is two block substitutions (`nk2d_shift_solve`) plus region-weighted algebra, all on
the device.  The eigen-pair comes from shift-invert Arnoldi with the same block solver at a
small POSITIVE shift, where `-(mat - mu I)` is a non-singular M-matrix.

The reference's `sigma=0.0` asks ARPACK to invert the exactly singular `mat`; its second
eigenvalue then depends on ARPACK's random start vector (real part scattered by several per
cent, `tests/test_oracle_phosphorus.py`), and with it `shift` and the preconditioner.  The
eigenvalue used here is the converged one (it agrees with a dense eigen-decomposition).
"""

import logging
import time

import numpy as np


class PhosphorusPrecond:
    """factorised phosphorus preconditioner for one linearisation state"""

    def __init__(self, eng, po4, time_range, mu=0.02, tol=1.0e-10, max_solves=80, start=None):
        """`start`: start vector of the Arnoldi process, e.g. `restart` of the previous Newton
        iteration's preconditioner (the eigenvectors move little between iterations)"""
        wall0 = time.time()
        self.eng = eng
        t0, t1 = float(time_range[0]), float(time_range[1])
        self.t_mid = t0 + 0.5 * (t1 - t0)        # time_n = 1 (phosphorus.py:208-229)
        self.scale = t1 - t0
        ylin = np.zeros(eng.shape)
        ylin[0] = po4                             # only po4 enters the Jacobian (phosphorus.py:213-216)
        eng.set_lin_state(eng.upload(ylin))
        self.ones = eng.upload(np.ones(eng.shape))
        self.e_vals, null_vect, self.eig_solves = self._smallest_eigs(mu, tol, max_solves, start)
        wall1 = time.time()
        self.null_vect = null_vect
        self.shift = 0.5 * self.e_vals[1].real
        eng.shift_factor(self.t_mid, self.scale, [self.shift, 0.5 * self.shift])
        e_vect = eng.upload(null_vect.reshape(eng.shape))
        self.e_hat = eng.scale(e_vect, 1.0 / eng.dot(e_vect, self.ones))
        logging.getLogger(__name__).info(
            "phosphorus preconditioner: %d Arnoldi solves (factor %.2f s, solves + orthogonalisation %.2f s, host %.2f s), "
            "shift %.6e, factorisation of the two shifted systems %.2f s",
            self.eig_solves, self.clock["factor"], self.clock["solve"], self.clock["host"], self.shift,
            time.time() - wall1)

    def _smallest_eigs(self, mu, tol, max_solves, start):
        """eigenvalues of mat closest to zero and the null vector by shift-invert Arnoldi: the
        Krylov space of (mat - mu I)^-1 is built with the device block solver (one solve per
        basis vector, ~20 in all) and orthogonalised on the device with the engine's Gram-Schmidt
        kernels under a uniform inner product (one region, unit weights -- any inner product gives
        the same Hessenberg eigenvalues); only the small Hessenberg eigenproblem runs on the host.
        A Ritz pair (theta, s) of the inverse has residual |h[j+1,j] s[j]|."""
        eng = self.eng
        n = int(np.prod(eng.shape))
        clock = {"factor": -time.time(), "solve": 0.0, "host": 0.0}
        eng.shift_factor(self.t_mid, self.scale, [mu])
        eng.sync()
        clock["factor"] += time.time()
        if start is None or start.shape != (n,):
            start = np.random.default_rng(0).standard_normal(n)
        region = getattr(eng, "_region", None)
        plane = (eng.nz, eng.ny)
        eng.set_region(np.ones(plane, dtype=np.int32), np.ones(plane))
        try:
            first = eng.upload(start.reshape(eng.shape))
            basis = [eng.scale(first, 1.0 / np.sqrt(eng.dot(first, first)))]
            hess = np.zeros((max_solves + 1, max_solves))
            for col in range(max_solves):
                tick = time.time()
                work = eng.shift_solve(0, basis[col])
                # modified Gram-Schmidt, repeated only when cancellation was severe (DGKS criterion)
                before = np.sqrt(eng.dot(work, work)[0])
                for _ in range(2):
                    hess[:col + 1, col] += eng.mgs(work, basis)[:, 0]
                    after = np.sqrt(eng.dot(work, work)[0])
                    if after > 0.7 * before:
                        break
                    before = after
                hess[col + 1, col] = after
                basis.append(eng.scale(work, 1.0 / after))
                clock["solve"] += time.time() - tick
                used = col + 1
                if used % 4 and used != max_solves:
                    continue
                tick = time.time()
                theta, ritz = np.linalg.eig(hess[:used, :used])       # theta ~ 1 / (lambda - mu)
                lam = mu + 1.0 / theta
                order = np.argsort(np.abs(lam))
                lam, ritz, theta = lam[order], ritz[:, order], theta[order]
                resid = abs(hess[used, used - 1]) * np.abs(ritz[used - 1, :3]) / np.abs(theta[:3])
                clock["host"] += time.time() - tick
                if used >= 8 and np.all(resid <= tol):
                    break
            if not np.all(resid <= tol):
                logging.getLogger(__name__).warning(
                    "phosphorus preconditioner: Arnoldi stopped after %d solves with Ritz residuals %s", used, resid)
            if abs(lam[0]) > 1.0e-6 * abs(lam[1]):
                # mat = T J has an exact null vector (total P is conserved); without it the mean-preserving
                # projection of the reference is undefined
                raise RuntimeError(f"smallest eigenvalue {lam[0]} is not a null eigenvalue")
            lead = ritz[:, :3]
            null_coef = lead[:, 0]
            if np.max(np.abs(null_coef.imag)) > 1.0e-10 * np.max(np.abs(null_coef.real)):
                raise RuntimeError("1st eigenvector has non-trivial imaginary part")
            null_vect = eng.download(eng.lin_comb(basis[:used], null_coef.real)).reshape(-1)
            # start vector of the next preconditioner: the invariant subspace found here
            restart_coef = lead[:, 0].real + lead[:, 1].real + lead[:, 1].imag
            self.restart = eng.download(eng.lin_comb(basis[:used], restart_coef)).reshape(-1)
        finally:
            if region is not None:
                eng.set_region(*region)
        self.clock = clock
        return lam, np.ascontiguousarray(null_vect), used

    def apply(self, v, out=None):
        eng = self.eng
        sol_full = eng.shift_solve(0, v)
        sol_half = eng.shift_solve(1, v)
        sol = eng.axpby(2.0, sol_half, -1.0, sol_full, out=sol_half)
        sol = eng.axpby(1.0, sol, -eng.dot(sol, self.ones), self.e_hat, out=sol)
        return eng.axpby(1.0, sol, -1.0, v, out=out)
