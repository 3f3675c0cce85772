"""Binding of the plugin to libnk2d.so (C ABI of include/nk2d.h through this repository's ctypes layer).

`HipBackend` is the product path: it needs the built library and a GPU and fails loudly otherwise.  A test
harness without a GPU may install a stand-in with `set_backend` (tests/test_ref_dropin.py checks the plugin's
plumbing under the reference's real driver that way); nothing in this package falls back by itself."""
import numpy as np

_BACKEND = None
_MODEL_STATE_CLS = None


def set_backend(obj):
    global _BACKEND
    _BACKEND = obj


def bind(model_state_cls):
    """called by the plugin's ModelState once its class variables (axes, model configuration) exist"""
    global _MODEL_STATE_CLS
    _MODEL_STATE_CLS = model_state_cls


def backend():
    global _BACKEND
    if _BACKEND is None:
        if _MODEL_STATE_CLS is None:
            raise RuntimeError("py_driver_2d_hip: no ModelState has been constructed yet")
        _BACKEND = HipBackend(_MODEL_STATE_CLS)
    return _BACKEND


class HipBackend:
    """one nk2d context (HIP stream) per tracer module, created on first use"""

    def __init__(self, model_state_cls):
        from nk_ooc_amd.grid import Grid2d, SpatialAxis

        modelinfo = model_state_cls.model_config_obj.modelinfo
        depth = SpatialAxis(modelinfo["depth_axisname"], np.array(model_state_cls.depth.edges),
                            modelinfo.get("depth_units"))
        ypos = SpatialAxis(modelinfo["ypos_axisname"], np.array(model_state_cls.ypos.edges),
                           modelinfo.get("ypos_units"))
        self.grid = Grid2d(depth, ypos, float(modelinfo["max_abs_vvel"]), float(modelinfo["horiz_mix_coeff"]))
        self.config = model_state_cls.model_config_obj
        self.engines = {}

    def engine(self, tracer_module):
        from nk_ooc_amd.engine import iage_engine

        name = tracer_module.name
        if name not in self.engines:
            if name != "iage":
                raise NotImplementedError(f"py_driver_2d_hip: tracer module {name} is not routed to the HIP library yet")
            eng = iage_engine(self.grid)
            tracer_name = next(iter(tracer_module._tracer_module_def["tracers"]))
            grid_vars = tracer_module.get_grid_vars(tracer_name)
            eng.set_region(grid_vars["region_mask"], grid_vars["grid_weight"])
            self.engines[name] = eng
        return self.engines[name]

    def last_schedules(self):
        """accepted Radau steps of the most recent free-running year of every module: what the plugin's ModelState
        attaches to a comp_fcn result, for the products around it"""
        return {name: eng.last_schedule() for name, eng in self.engines.items()}

    def forward_year(self, tracer_module, y0, t_eval, frozen=None):
        """what solve_ivp returned to the reference: times and the solution at them, shape (N, len(t_eval)).
        frozen: {module name: schedule} of the year whose steps this one repeats (the perturbed year of a
        finite-difference product); a state the recorded Newton counts do not converge for gets a free-running year"""
        from nk_ooc_amd.engine import Nk2dFrozenMismatch

        eng = self.engine(tracer_module)
        x = eng.upload(np.asarray(y0).reshape(eng.shape))
        if len(t_eval) == 2:
            fx = None
            sched = (frozen or {}).get(tracer_module.name)
            if sched is not None and len(sched) > 0:
                try:
                    fx, _ = eng.comp_fcn_frozen(x, sched)
                except Nk2dFrozenMismatch:
                    fx = None
            if fx is None:
                fx, _, _ = eng.comp_fcn(x)
            y_end = np.asarray(y0).reshape(-1) + eng.download(fx).reshape(-1)
            return np.asarray(t_eval), np.stack([np.asarray(y0).reshape(-1), y_end], axis=1)
        _, _, hist = eng.comp_fcn_hist(x, np.asarray(t_eval))
        return np.asarray(t_eval), hist.reshape(len(t_eval), -1).T.copy()

    def precond_apply(self, tracer_module, vals):
        eng = self.engine(tracer_module)
        return eng.download(eng.precond_apply(eng.upload(np.asarray(vals).reshape(eng.shape))))
