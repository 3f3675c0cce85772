"""Oracle: Radau IIA(5) time integration exactly as SciPy drives it for the reference.

TEST INFRASTRUCTURE ONLY (see `oracle/__init__.py`).

The reference's `comp_fcn` (`nk_ooc/py_driver_2d/model_state.py:95-121`) calls
`scipy.integrate.solve_ivp(tend, (0, T), y0, "Radau", t_eval=[0, T],
max_step=0.01 T, atol=rtol=1e-6, jac=analytic sparse)` and returns `y(T) - y0`.
The integrator is third-party (SciPy, reference pin 1.9.1; this image 1.15.3,
`scipy/integrate/_ivp/radau.py`, `common.py:63-134`, `base.py:181-208`,
`ivp.py:660-723`).  This module restates that published algorithm -- constants,
initial-step heuristic, simplified-Newton collocation solve, error estimate and
filter, step-size controller, Jacobian/LU reuse rules, cubic dense output and
the final evaluation of the interpolant at T -- with the same floating-point
operation order, so that it reproduces `solve_ivp` bit for bit
(`tests/test_oracle_radau.py`).  On top of that it

* records the accepted-step schedule `(t, h, n_newton, t_jac)` so that the HIP
  integrator can be run in step-replay mode against it, and
* can itself replay a recorded schedule (`replay=`), which is the smooth map the
  1e-10 parity tests use.

Linear algebra: SuperLU through `scipy.sparse.linalg.splu`, as in SciPy's Radau.
"""

import numpy as np
from scipy.sparse import csc_matrix, eye
from scipy.sparse.linalg import splu

EPS = np.finfo(float).eps
S6 = 6 ** 0.5

# Butcher nodes, error-estimate weights
C = np.array([(4 - S6) / 10, (4 + S6) / 10, 1])
E = np.array([-13 - 7 * S6, -13 + 7 * S6, -1]) / 3

# eigenvalues of the inverse Butcher matrix: one real, one complex pair
MU_REAL = 3 + 3 ** (2 / 3) - 3 ** (1 / 3)
MU_COMPLEX = (3 + 0.5 * (3 ** (1 / 3) - 3 ** (2 / 3))
              - 0.5j * (3 ** (5 / 6) + 3 ** (7 / 6)))

# similarity transform of the stage system and its inverse
T = np.array([
    [0.09443876248897524, -0.14125529502095421, 0.03002919410514742],
    [0.25021312296533332, 0.20412935229379994, -0.38294211275726192],
    [1, 1, 0]])
TI = np.array([
    [4.17871859155190428, 0.32768282076106237, 0.52337644549944951],
    [-4.17871859155190428, -0.32768282076106237, 0.47662355450055044],
    [0.50287263494578682, -2.57192694985560522, 0.59603920482822492]])
TI_REAL = TI[0]
TI_COMPLEX = TI[1] + 1j * TI[2]

# dense-output (collocation polynomial) coefficients
P = np.array([
    [13 / 3 + 7 * S6 / 3, -23 / 3 - 22 * S6 / 3, 10 / 3 + 5 * S6],
    [13 / 3 - 7 * S6 / 3, -23 / 3 + 22 * S6 / 3, 10 / 3 - 5 * S6],
    [1 / 3, -8 / 3, 10 / 3]])

NEWTON_MAXITER = 6
MIN_FACTOR = 0.2
MAX_FACTOR = 10


def rms_norm(x):
    return np.linalg.norm(x) / x.size ** 0.5


def initial_step(fun, t0, y0, t_bound, max_step, f0, order, rtol, atol):
    """common.py:68-134 for forward integration"""
    interval = abs(t_bound - t0)
    if interval == 0.0:
        return 0.0
    scale = atol + np.abs(y0) * rtol
    d0 = rms_norm(y0 / scale)
    d1 = rms_norm(f0 / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    h0 = min(h0, interval)
    y1 = y0 + h0 * 1.0 * f0
    f1 = fun(t0 + h0 * 1.0, y1)
    d2 = rms_norm((f1 - f0) / scale) / h0
    if d1 <= 1e-15 and d2 <= 1e-15:
        h1 = max(1e-6, h0 * 1e-3)
    else:
        h1 = (0.01 / max(d1, d2)) ** (1 / (order + 1))
    return min(100 * h0, h1, interval, max_step)


def step_factor(h_abs, h_abs_old, err, err_old):
    """radau.py:139-176 (Gustafsson-type predictive controller)"""
    if err_old is None or h_abs_old is None or err == 0:
        mult = 1
    else:
        mult = h_abs / h_abs_old * (err_old / err) ** 0.25
    with np.errstate(divide="ignore"):
        return min(1, mult) * err ** -0.25


class Stats:
    def __init__(self):
        self.nfev = 0
        self.njev = 0
        self.nlu = 0
        self.nsteps = 0
        self.nrejected = 0
        self.nnewton = 0
        self.nsolve = 0


class RadauOracle:
    """integrate y' = fun(t, y) over [t0, t_bound] like solve_ivp(..., "Radau")"""

    def __init__(self, fun, jac, t0, y0, t_bound, max_step, rtol=1e-6, atol=1e-6):
        self.stats = Stats()
        self._fun = fun
        self._jac = jac
        self.t = t0
        self.y = np.array(y0, dtype=float)
        self.t_bound = t_bound
        self.n = self.y.size
        self.max_step = max_step
        self.rtol = rtol
        self.atol = atol
        self.f = self.fun(self.t, self.y)
        self.h_abs = initial_step(self.fun, self.t, self.y, t_bound, max_step, self.f,
                                  3, rtol, atol)
        self.h_abs_old = None
        self.err_old = None
        self.newton_tol = max(10 * EPS / rtol, min(0.03, rtol ** 0.5))
        self.J = csc_matrix(jac(t0, self.y))
        self.t_jac = t0
        self.stats.njev = 1
        self.I = eye(self.n, format="csc")
        self.current_jac = True
        self.LU_real = None
        self.LU_complex = None
        self.dense = None  # (t_old, h, y_old, Q)
        # accepted steps: (t, t_new, h, n_newton, t_jac, h_lu); h_lu is the step size
        # the LU factors in use were formed with (SciPy keeps factors across steps
        # and does not refactor when the last step is clipped to t_bound)
        self.schedule = []
        self.h_lu = None
        self.t_old = None

    # counted wrappers
    def fun(self, t, y):
        self.stats.nfev += 1
        return self._fun(t, y)

    def jac(self, t, y):
        self.stats.njev += 1
        self.t_jac = t
        return csc_matrix(self._jac(t, y), dtype=float)

    def lu(self, A):
        self.stats.nlu += 1
        return splu(A)

    def solve(self, LU, b):
        self.stats.nsolve += 1
        return LU.solve(b)

    def _dense_eval(self, times):
        """RadauDenseOutput._call_impl for a 1-d array of times"""
        t_old, h, y_old, Q = self.dense
        x = (times - t_old) / h
        p = np.tile(x, (3, 1))
        p = np.cumprod(p, axis=0)
        y = np.dot(Q, p)
        y += y_old[:, None]
        return y

    def newton(self, t, y, h, Z0, scale, LU_real, LU_complex, force_iters=None):
        """solve_collocation_system (radau.py:48-136); with `force_iters` the
        convergence tests are skipped and exactly that many iterations are run"""
        n = y.shape[0]
        M_real = MU_REAL / h
        M_complex = MU_COMPLEX / h
        W = TI.dot(Z0)
        Z = Z0
        F = np.empty((3, n))
        ch = h * C
        dW_norm_old = None
        dW = np.empty_like(W)
        converged = False
        rate = None
        tol = self.newton_tol
        kmax = NEWTON_MAXITER if force_iters is None else force_iters
        k = -1
        for k in range(kmax):
            for i in range(3):
                F[i] = self.fun(t + ch[i], y + Z[i])
            if not np.all(np.isfinite(F)):
                break
            f_real = F.T.dot(TI_REAL) - M_real * W[0]
            f_complex = F.T.dot(TI_COMPLEX) - M_complex * (W[1] + 1j * W[2])
            dW_real = self.solve(LU_real, f_real)
            dW_complex = self.solve(LU_complex, f_complex)
            dW[0] = dW_real
            dW[1] = dW_complex.real
            dW[2] = dW_complex.imag
            dW_norm = rms_norm(dW / scale)
            if dW_norm_old is not None:
                rate = dW_norm / dW_norm_old
            if force_iters is None and (rate is not None and (
                    rate >= 1
                    or rate ** (NEWTON_MAXITER - k) / (1 - rate) * dW_norm > tol)):
                break
            W += dW
            Z = T.dot(W)
            self.stats.nnewton += 1
            if force_iters is None and (
                    dW_norm == 0
                    or rate is not None and rate / (1 - rate) * dW_norm < tol):
                converged = True
                break
            dW_norm_old = dW_norm
        if force_iters is not None:
            converged = True
        return converged, k + 1, Z, rate

    def step(self):
        """one accepted step, radau.py:399-539"""
        t, y, f = self.t, self.y, self.f
        max_step, atol, rtol = self.max_step, self.atol, self.rtol
        min_step = 10 * np.abs(np.nextafter(t, np.inf) - t)
        if self.h_abs > max_step:
            h_abs, h_abs_old, err_old = max_step, None, None
        elif self.h_abs < min_step:
            h_abs, h_abs_old, err_old = min_step, None, None
        else:
            h_abs, h_abs_old, err_old = self.h_abs, self.h_abs_old, self.err_old
        J = self.J
        LU_real, LU_complex = self.LU_real, self.LU_complex
        current_jac = self.current_jac
        rejected = False
        accepted = False
        while not accepted:
            if h_abs < min_step:
                raise RuntimeError("Radau: step size too small")
            h = h_abs
            t_new = t + h
            if t_new - self.t_bound > 0:
                t_new = self.t_bound
            h = t_new - t
            h_abs = np.abs(h)
            if self.dense is None:
                Z0 = np.zeros((3, y.shape[0]))
            else:
                Z0 = self._dense_eval(t + h * C).T - y
            scale = atol + np.abs(y) * rtol
            converged = False
            while not converged:
                if LU_real is None or LU_complex is None:
                    LU_real = self.lu(MU_REAL / h * self.I - J)
                    LU_complex = self.lu(MU_COMPLEX / h * self.I - J)
                    self.h_lu = h
                converged, n_iter, Z, rate = self.newton(
                    t, y, h, Z0, scale, LU_real, LU_complex)
                if not converged:
                    if current_jac:
                        break
                    J = self.jac(t, y)
                    current_jac = True
                    LU_real = None
                    LU_complex = None
            if not converged:
                h_abs *= 0.5
                LU_real = None
                LU_complex = None
                continue
            y_new = y + Z[-1]
            ZE = Z.T.dot(E) / h
            error = self.solve(LU_real, f + ZE)
            scale = atol + np.maximum(np.abs(y), np.abs(y_new)) * rtol
            err = rms_norm(error / scale)
            safety = 0.9 * (2 * NEWTON_MAXITER + 1) / (2 * NEWTON_MAXITER + n_iter)
            if rejected and err > 1:
                error = self.solve(LU_real, self.fun(t, y + error) + ZE)
                err = rms_norm(error / scale)
            if err > 1:
                factor = step_factor(h_abs, h_abs_old, err, err_old)
                h_abs *= max(MIN_FACTOR, safety * factor)
                LU_real = None
                LU_complex = None
                rejected = True
                self.stats.nrejected += 1
            else:
                accepted = True
        recompute_jac = n_iter > 2 and rate > 1e-3
        factor = step_factor(h_abs, h_abs_old, err, err_old)
        factor = min(MAX_FACTOR, safety * factor)
        if not recompute_jac and factor < 1.2:
            factor = 1
        else:
            LU_real = None
            LU_complex = None
        self.schedule.append((t, t_new, h, n_iter, self.t_jac, self.h_lu))
        f_new = self.fun(t_new, y_new)
        if recompute_jac:
            J = self.jac(t_new, y_new)
            current_jac = True
        else:
            current_jac = False
        self.h_abs_old = self.h_abs
        self.err_old = err
        self.h_abs = h_abs * factor
        self.dense = (t, t_new - t, y, np.dot(Z.T, P))
        self.t_old = t
        self.t, self.y, self.f = t_new, y_new, f_new
        self.LU_real, self.LU_complex = LU_real, LU_complex
        self.current_jac = current_jac
        self.J = J
        self.stats.nsteps += 1

    def run(self):
        """step to t_bound and return y(t_bound) as solve_ivp(t_eval=[t0, t_bound])
        does: the dense-output polynomial of the last step evaluated at t_bound
        (ivp.py:707-723 with radau.py:557-570)"""
        while self.t < self.t_bound:
            self.step()
        return self._dense_eval(np.array([self.t_bound]))[:, -1]

    def run_replay(self, schedule):
        """consume a recorded accepted-step schedule: no error control, no
        rejected attempts; Newton runs the recorded number of iterations with the
        Jacobian evaluated at the recorded time"""
        t_jac_cur = self.t_jac
        h_lu_cur = None
        for (t, t_new, h, n_iter, t_jac, h_lu) in schedule:
            assert t == self.t
            y = self.y
            if t_jac != t_jac_cur:
                # SciPy only ever evaluates a Jacobian at the start point of a step; the schedules of the HIP library's
                # production mode (option "jac_stage") take it at a stage time of the step, t + c_i h, for modules
                # whose Jacobian is a function of time alone -- evaluated here with the state at the step start, which
                # such a Jacobian does not read
                self.J = csc_matrix(self._jac(t_jac, y), dtype=float)
                t_jac_cur = t_jac
                h_lu_cur = None
            if h_lu != h_lu_cur:
                self.LU_real = self.lu(MU_REAL / h_lu * self.I - self.J)
                self.LU_complex = self.lu(MU_COMPLEX / h_lu * self.I - self.J)
                h_lu_cur = h_lu
            if self.dense is None:
                Z0 = np.zeros((3, y.shape[0]))
            else:
                Z0 = self._dense_eval(t + h * C).T - y
            scale = self.atol + np.abs(y) * self.rtol
            _, _, Z, _ = self.newton(t, y, h, Z0, scale, self.LU_real, self.LU_complex,
                                     force_iters=n_iter)
            self.dense = (t, t_new - t, y, np.dot(Z.T, P))
            self.t, self.y = t_new, y + Z[-1]
            self.stats.nsteps += 1
        return self._dense_eval(np.array([self.t]))[:, -1]


def comp_fcn(module, x, time_range=(0.0, 365.0 * 86400.0), replay=None,
             return_solver=False):
    """F(x) = y(T) - x for one tracer module (py_driver_2d/model_state.py:95-121)"""
    y0 = np.asarray(x, dtype=float).reshape(-1)
    solver = RadauOracle(module.comp_tend, module.comp_jacobian, time_range[0], y0,
                         time_range[1],
                         max_step=(time_range[1] - time_range[0]) * 0.01,
                         rtol=1.0e-6, atol=1.0e-6)
    y_end = solver.run() if replay is None else solver.run_replay(replay)
    res = y_end - y0
    return (res, solver) if return_solver else res
