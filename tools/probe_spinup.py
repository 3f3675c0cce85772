"""probe: a complete Newton-Krylov spin-up of the iage module through the driver mirror (set-up with one fixed-point year,
then nk_driver.run with the reference's newton_krylov.cfg defaults) -- wall time, Newton and Krylov iterations, forward
years by kind

    python tools/probe_spinup.py [n]           NK2D_JVP_FROZEN=0 for free-running products
"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ooc_amd import nk_driver  # noqa: E402
from nk_ooc_amd.model_state import ModelState  # noqa: E402
from nk_ooc_amd.setup_solver import make_config, setup  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
modules = sys.argv[2] if len(sys.argv) > 2 else "iage"          # e.g. phosphorus, or iage,phosphorus
work = tempfile.mkdtemp(prefix="nk2d_spinup_")
cfg = make_config(work, n, n, tracer_module_names=modules)
ModelState.reset_class()
ModelState.write_files = True
t0 = time.time()
setup(cfg, fp_cnt=1)
t1 = time.time()
solver = nk_driver.run(cfg)
t2 = time.time()
kry = []
it = 0
while os.path.isdir(os.path.join(work, f"krylov_{it:02}")):
    state = json.load(open(os.path.join(work, f"krylov_{it:02}", "Krylov_state.json")))
    kry.append(state["iteration"])
    it += 1
rejected = {name: eng.frozen_fallbacks() for name, eng in ModelState._engines.items()}
print(f"n={n} modules {modules} frozen products: {os.environ.get('NK2D_JVP_FROZEN', '1') != '0'}; set-up {t1 - t0:.2f} s (grid, one fixed-point year, "
      f"F, preconditioner factors); Newton-Krylov solve {t2 - t1:.2f} s: converged {bool(solver.converged().all())}, "
      f"{solver.get_iteration()} Newton iterations, Krylov iterations {kry}, frozen years rejected {rejected}", flush=True)
