"""probe: forward year of three tracer modules (iage, phosphorus, forced decay) on ONE GPU,
module years one after the other (NK2D_SERIAL_MODULES=1) or concurrently (default)"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ooc_amd.model_config import ModelConfig  # noqa: E402
from nk_ooc_amd.model_state import ModelState  # noqa: E402
from nk_ooc_amd.setup_solver import gen_grid_vars_file, make_config  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
work = tempfile.mkdtemp()
extra = {"forced_surf_restore_opt": "none", "forced_sms_opt": "decay", "forced_sms_decay_rate": "1.0e-8"}
cfg = make_config(work, n, n, tracer_module_names="iage,phosphorus,forced_{suff}:dye", extra_modelinfo=extra)
gen_grid_vars_file(cfg["modelinfo"])
ModelState.model_config_obj = ModelConfig(cfg["modelinfo"])
ModelState.write_files = False
iterate = ModelState("gen_init_iterate")
for rep in range(2):
    t0 = time.time()
    fcn = iterate.comp_fcn(os.path.join(work, f"fcn_{rep}.nc"), None)
    print(f"n={n} three modules, forward years: {time.time() - t0:.3f} s  " +
          " ".join(f"{st['seconds']:.2f}" for st in ModelState.last_stats), flush=True)
