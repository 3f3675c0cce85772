"""where does the host time of a Krylov iteration go?  bench.py's Workload (iage n x n, checkpoint trail on) under cProfile
    python tools/probe_krylov_profile.py [n] [iterations]"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
device = torch.device("cuda", 0)
from nk_ooc_amd import trail  # noqa: E402

trail.set_enabled(os.environ.get("NK2D_ASYNC_TRAIL", "1") != "0")      # (as bench.py; the profile is the main thread's)
wl = bench.Workload(n, 0, "prof", write_files=True)
wl.krylov(1, "krylov_warm", device)
wl.eng.sync()
prof = cProfile.Profile()
t0 = time.perf_counter()
prof.enable()
wl.krylov(iters, "krylov_prof", device)
wl.eng.sync()
prof.disable()
el = time.perf_counter() - t0
print(f"n={n}: {iters} Krylov iterations in {el:.3f} s = {1000 * el / iters:.1f} ms each ({iters / el:.3f} JVPs/s)")
buf = io.StringIO()
pstats.Stats(prof, stream=buf).sort_stats("tottime").print_stats(28)
print(buf.getvalue())
wl.close()
