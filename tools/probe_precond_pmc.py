"""probe (torch-free) for rocprofv3 passes over the preconditioner: nk2d_precond_setup (block
elimination) + nk2d_precond_apply (2 ny dense mat-vecs streamed from HBM) of iage at n x n.

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_pc_f -- python tools/probe_precond_pmc.py 416

Round 1 recorded a host SIGSEGV with nk2d_precond_setup on the stack under `rocprofv3 --pmc -- python
bench.py` (gpurun_out/pmc_write.err).  To resolve such a trace this probe writes /proc/self/maps next to
its log right before the call, and enables faulthandler."""
import faulthandler
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

faulthandler.enable()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
napply = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tag = sys.argv[3] if len(sys.argv) > 3 else "run"
grid = Grid2d.default(n, n)
eng = iage_engine(grid)
rng = np.random.default_rng(0)
v = eng.upload(rng.standard_normal((2, n, n)))
out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
with open("/proc/self/maps") as src, open(os.path.join(out_dir, f"precond_probe_maps_{tag}.txt"), "w") as dst:
    dst.write(src.read())
print("maps written; calling precond_setup", flush=True)
t0 = time.perf_counter()
eng.precond_setup()
eng.sync()
print(f"precond_setup {time.perf_counter() - t0:.3f} s", flush=True)
for i in range(napply):
    t0 = time.perf_counter()
    pv = eng.precond_apply(v)
    eng.sync()
    print(f"precond_apply {1e3 * (time.perf_counter() - t0):.3f} ms", flush=True)
print("checksum", float(np.abs(eng.download(pv)).sum()), flush=True)
