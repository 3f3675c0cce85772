"""iage tracer module of the py_driver_2d_hip plugin: the reference's class with the preconditioner apply
(`nk_ooc/py_driver_2d/iage.py:66-93`: three sparse products and a SuperLU solve per call) routed to
nk2d_precond_apply (factorised once, streamed from HBM)."""
from nk_ooc.py_driver_2d import iage as ref_iage

from . import _backend


class iage(ref_iage.iage):  # the reference looks tracer module classes up by this name
    def apply_precond_jacobian(self, time_range, res_tms, processes):
        vals = self.get_tracer_vals_all()
        out = _backend.backend().precond_apply(self, vals)
        res_tms.set_tracer_vals_all(out.reshape(vals.shape))
