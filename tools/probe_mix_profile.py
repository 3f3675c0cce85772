"""cProfile of three Krylov iterations of the three-module mix at 416 x 416, modules back to back"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402

device = torch.device("cuda", 0)
os.environ["NK2D_SERIAL_MODULES"] = "1"
wl = bench.MixWorkload(416, bench.MIX_NAMES, 0, "prof")
try:
    wl.krylov(bench.WARM_ITERS, "warm", device)
    bench.warm_until_cached(lambda: wl.krylov(1, "warm_more", device), wl.engines())
    wl.sync()
    pr = cProfile.Profile()
    pr.enable()
    wl.krylov(3, "timed", device)
    wl.sync()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
finally:
    wl.close()
