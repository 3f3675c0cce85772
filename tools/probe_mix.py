"""config4_mix: where a Krylov iteration of iage + phosphorus + forced (decay) at 416 x 416 goes -- the perturbed year of each
module, side by side and back to back (NK2D_SERIAL_MODULES)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=416)
args = ap.parse_args()
import torch  # noqa: E402

device = torch.device("cuda", 0)
from nk_ooc_amd.model_state import ModelState  # noqa: E402

for serial in ("1", "0"):
    os.environ["NK2D_SERIAL_MODULES"] = serial
    wl = bench.MixWorkload(args.grid, bench.MIX_NAMES, 0, f"probe{serial}")
    try:
        wl.krylov(bench.WARM_ITERS, "warm", device)
        bench.warm_until_cached(lambda: wl.krylov(1, "warm_more", device), wl.engines())
        wl.sync()
        t0 = time.perf_counter()
        wl.krylov(3, "timed", device)
        wl.sync()
        el = (time.perf_counter() - t0) / 3
        years = {tms.name: (round(st["seconds"], 4), st["nlaunch"]) for tms, st in zip(wl.iterate.tracer_modules, ModelState.last_stats)}
        one = {tms.name: tms.eng.counter("frozen_persistent_years") for tms in wl.iterate.tracer_modules}
        t1 = time.perf_counter()
        for tms in wl.iterate.tracer_modules:
            v = tms.eng.precond_apply(tms.vec)
        wl.sync()
        pc = time.perf_counter() - t1
        print(f"serial={serial}: {1e3 * el:.1f} ms per Krylov iteration; perturbed years (s, launches) {years}; one-launch years so far {one}; "
              f"three preconditioner applies {1e3 * pc:.1f} ms", flush=True)
    finally:
        wl.close()
