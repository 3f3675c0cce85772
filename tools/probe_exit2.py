"""which part of a two-module run makes the HIP runtime's tear-down crash at exit?  modes: fcn, fcn_serial, solve, solve_serial, fcn_noreset"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
if "serial" in mode:
    os.environ["NK2D_SERIAL_MODULES"] = "1"
from nk_ooc_amd.krylov_solver import KrylovSolver  # noqa: E402
from nk_ooc_amd.model_config import ModelConfig  # noqa: E402
from nk_ooc_amd.model_state import ModelState  # noqa: E402
from nk_ooc_amd.setup_solver import gen_grid_vars_file, make_config  # noqa: E402

opts = [kv.split("=") for kv in os.environ.get("PROBE_OPTS", "").split(",") if kv]
if opts:
    from nk_ooc_amd import engine as _engine

    _init = _engine.ModuleEngine.__init__

    def _patched(self, *args, **kwargs):
        _init(self, *args, **kwargs)
        for key, val in opts:
            self.set_option(key, float(val))

    _engine.ModuleEngine.__init__ = _patched
tmp = tempfile.mkdtemp()
names = sys.argv[2] if len(sys.argv) > 2 else "iage,forced_{suff}:dye"
extra = {"forced_surf_restore_opt": "none", "forced_sms_opt": "decay", "forced_sms_decay_rate": "1.0e-8"}
cfg = make_config(tmp, 22, 9, tracer_module_names=names, extra_modelinfo=extra,
                  extra_solverinfo={"krylov_max_iter": "2", "krylov_rel_tol": "1e-9"})
gen_grid_vars_file(cfg["modelinfo"])
ModelState.reset_class()
ModelState.model_config_obj = ModelConfig(cfg["modelinfo"])
ModelState.write_files = True
iterate = ModelState("gen_init_iterate")
fcn = iterate.comp_fcn(os.path.join(tmp, "fcn_00.nc"), None)
if mode.startswith("solve"):
    solverinfo = dict(cfg["solverinfo"], krylov_workdir=os.path.join(tmp, "krylov_00"))
    solver = KrylovSolver(iterate, solverinfo, False, False, None)
    solver.solve(os.path.join(tmp, "increment_00.nc"), fcn)
if "noreset" not in mode:
    ModelState.reset_class()
print(mode, names, os.environ.get("PROBE_OPTS", ""), "done", flush=True)
