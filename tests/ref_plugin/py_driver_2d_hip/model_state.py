"""ModelState of the py_driver_2d_hip plugin: the reference's py_driver_2d ModelState with the forward
year taken from the HIP library.  The reference's method body (history file, step log, postprocessing,
re-invocation) runs unchanged; only its call of `integrate.solve_ivp` is answered by the backend."""
import contextlib
import types

from nk_ooc.py_driver_2d import model_state as ref_model_state

from . import _backend


_FROZEN = None     # {module name: schedule} while a finite-difference product computes its perturbed year


@contextlib.contextmanager
def _forward_year_from(backend):
    """while active, `integrate.solve_ivp(...)` inside nk_ooc/py_driver_2d/model_state.py:102-114 returns the
    backend's year: the call passes the tracer module's bound `comp_tend`, the time range, the initial
    values, "Radau" and t_eval -- everything the device integrator needs is there"""

    def solve_ivp(fun, t_span, y0, method, t_eval, **kwargs):
        if method != "Radau":
            raise NotImplementedError(f"py_driver_2d_hip: integrator {method}")
        if _FROZEN and len(t_eval) == 2:
            times, vals = backend.forward_year(fun.__self__, y0, t_eval, frozen=_FROZEN)
        else:
            times, vals = backend.forward_year(fun.__self__, y0, t_eval)
        return types.SimpleNamespace(t=times, y=vals, success=True)

    saved = ref_model_state.integrate
    ref_model_state.integrate = types.SimpleNamespace(solve_ivp=solve_ivp)
    try:
        yield
    finally:
        ref_model_state.integrate = saved


class ModelState(ref_model_state.ModelState):
    """discovered by get_model_state_class("py_driver_2d_hip") (nk_ooc/model_state_base.py:627-646)"""

    def __init__(self, fname):
        # the driver configures the class it discovered (nk_driver.py:56); the parent's constructor checks
        # its own class attribute (py_driver_2d/model_state.py:31-32)
        ref_model_state.ModelState.model_config_obj = type(self).model_config_obj
        super().__init__(fname)
        _backend.bind(type(self))

    def comp_fcn(self, res_fname, solver_state, hist_fname=None):
        backend = _backend.backend()
        with _forward_year_from(backend):
            res = super().comp_fcn(res_fname, solver_state, hist_fname)
        # the accepted Radau steps of the years just run travel with the result (in this process): the products around
        # it repeat them.  A backend without recorded steps (the CPU stand-in of the test harness) has none.
        if not _FROZEN and hasattr(backend, "last_schedules"):
            res._nk2d_sched = backend.last_schedules()
        return res

    def comp_jacobian_fcn_state_prod(self, fcn, direction, res_fname, solver_state):
        """the reference's finite-difference product (model_state_base.py:492-527), its perturbed year on the steps of
        the year that produced `fcn` (internal numerical differentiation; this repository's DESIGN.md section 3c)"""
        global _FROZEN
        _FROZEN = getattr(fcn, "_nk2d_sched", None)
        try:
            return super().comp_jacobian_fcn_state_prod(fcn, direction, res_fname, solver_state)
        finally:
            _FROZEN = None
