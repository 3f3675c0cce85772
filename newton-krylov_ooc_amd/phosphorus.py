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
the device.  The eigen-pair comes from subspace inverse iteration with the same block solver
at a small POSITIVE shift, where `-(mat - mu I)` is a non-singular M-matrix.

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

    def __init__(self, eng, po4, time_range, nvec=6, mu=0.02, tol=1.0e-10, max_iter=300, start=None):
        """`start`: (n, nvec) basis to start the subspace iteration from, e.g. the converged basis of
        the previous Newton iteration's preconditioner (`self.basis`)"""
        wall0 = time.time()
        self.eng = eng
        t0, t1 = float(time_range[0]), float(time_range[1])
        self.t_mid = t0 + 0.5 * (t1 - t0)        # time_n = 1 (phosphorus.py:208-229)
        self.scale = t1 - t0
        ylin = np.zeros(eng.shape)
        ylin[0] = po4                             # only po4 enters the Jacobian (phosphorus.py:213-216)
        eng.set_lin_state(eng.upload(ylin))
        self.ones = eng.upload(np.ones(eng.shape))
        self.e_vals, null_vect, self.eig_iters = self._smallest_eigs(nvec, mu, tol, max_iter, start)
        wall1 = time.time()
        self.null_vect = null_vect
        self.shift = 0.5 * self.e_vals[1].real
        eng.shift_factor(self.t_mid, self.scale, [self.shift, 0.5 * self.shift])
        e_vect = eng.upload(null_vect.reshape(eng.shape))
        self.e_hat = eng.scale(e_vect, 1.0 / eng.dot(e_vect, self.ones))
        logging.getLogger(__name__).info(
            "phosphorus preconditioner: %d subspace iterations (%.2f s), shift %.6e, factorisations %.2f s",
            self.eig_iters, wall1 - wall0, self.shift, time.time() - wall1)

    def _smallest_eigs(self, nvec, mu, tol, max_iter, start):
        """eigenvalues of mat closest to zero and the null vector, by subspace inverse
        iteration with Rayleigh-Ritz extraction; the solves run on the device, the
        (n x nvec) dense algebra on the host"""
        eng = self.eng
        n = int(np.prod(eng.shape))
        eng.shift_factor(self.t_mid, self.scale, [mu])
        if start is not None and start.shape == (n, nvec):
            basis, _ = np.linalg.qr(start)
        else:
            basis, _ = np.linalg.qr(np.random.default_rng(0).standard_normal((n, nvec)))
        prev = None
        for it in range(max_iter):
            work = np.empty((n, nvec))
            for col in range(nvec):
                sol = eng.shift_solve(0, eng.upload(basis[:, col].reshape(eng.shape)))
                work[:, col] = eng.download(sol).reshape(-1)
            theta, ritz = np.linalg.eig(basis.T @ work)       # theta ~ 1 / (lambda - mu)
            lam = mu + 1.0 / theta
            order = np.argsort(np.abs(lam))
            lam, ritz = lam[order], ritz[:, order]
            lead = lam[:3]
            if prev is not None and np.all(np.abs(lead[1:] - prev[1:]) <= tol * np.abs(lead[1:])):
                break
            prev = lead
            basis, _ = np.linalg.qr(work)
        self.basis = basis
        null_comp = basis @ ritz[:, 0]
        if np.max(np.abs(null_comp.imag)) > 1.0e-10 * np.max(np.abs(null_comp.real)):
            raise RuntimeError("1st eigenvector has non-trivial imaginary part")
        return lam, np.ascontiguousarray(null_comp.real), it + 1

    def apply(self, v, out=None):
        eng = self.eng
        sol_full = eng.shift_solve(0, v)
        sol_half = eng.shift_solve(1, v)
        sol = eng.axpby(2.0, sol_half, -1.0, sol_full, out=sol_half)
        sol = eng.axpby(1.0, sol, -eng.dot(sol, self.ones), self.e_hat, out=sol)
        return eng.axpby(1.0, sol, -1.0, v, out=out)
