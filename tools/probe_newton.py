"""probe: Newton-Krylov to convergence (NK2D_MODULES, default phosphorus), per-iteration norms and timings"""
import logging
import os
import sys
import tempfile
import time


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ooc_amd import nk_driver  # noqa: E402
from nk_ooc_amd.model_state import ModelState  # noqa: E402
from nk_ooc_amd.setup_solver import make_config, setup  # noqa: E402

nz, ny = int(sys.argv[1]), int(sys.argv[2])
logging.basicConfig(level=logging.INFO, format="%(message)s", stream=sys.stdout)
work = tempfile.mkdtemp()
cfg = make_config(work, nz, ny, tracer_module_names=os.environ.get("NK2D_MODULES", "phosphorus"),
                  extra_solverinfo={"newton_max_iter": sys.argv[3] if len(sys.argv) > 3 else "6"})
ModelState.write_files = True
t0 = time.time()
setup(cfg, fp_cnt=1)
print("setup %.1fs" % (time.time() - t0), flush=True)
t0 = time.time()
solver = nk_driver.run(cfg)
print("newton iterations", solver.get_iteration(), "converged", solver.converged(), "%.1fs" % (time.time() - t0))
x = solver.iterate.tracer_modules[0]
print("mean total P", x.mean())
