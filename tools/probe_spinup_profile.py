"""probe: where the wall time of a Newton-Krylov spin-up goes (cProfile of the driver mirror, cumulative top entries)"""
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ooc_amd import nk_driver  # noqa: E402
from nk_ooc_amd.model_state import ModelState  # noqa: E402
from nk_ooc_amd.setup_solver import make_config, setup  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
work = tempfile.mkdtemp(prefix="nk2d_spinup_")
cfg = make_config(work, n, n)
ModelState.reset_class()
ModelState.write_files = True
setup(cfg, fp_cnt=1)
prof = cProfile.Profile()
t0 = time.time()
prof.enable()
solver = nk_driver.run(cfg)
prof.disable()
print(f"n={n}: Newton-Krylov solve {time.time() - t0:.2f} s, {solver.get_iteration()} Newton iterations")
out = io.StringIO()
pstats.Stats(prof, stream=out).sort_stats("cumulative").print_stats(45)
print(out.getvalue())
