"""free-running forward year by launches and as a command stream (nk2d_stream.hip): seconds, launches, commands"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [416]
for n in sizes:
    eng = iage_engine(Grid2d.default(n, n))
    eng.set_option("stream_years", 0)       # (by launches)
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
    res = {}
    for mode in (0, 1):
        eng.set_option("stream_years", mode)
        best = None
        for rep in range(3):
            c0 = eng.counter("stream_commands")
            l0 = eng.counter("stream_launches")
            t0 = time.perf_counter()
            fx, st, sched = eng.comp_fcn(x, record=True)
            wall = time.perf_counter() - t0
            if best is None or wall < best[0]:
                best = (wall, st, eng.counter("stream_commands") - c0, eng.counter("stream_launches") - l0)
        res[mode] = (best, eng.download(fx), sched)
        wall, st, cmds, kl = best
        print(f"{n}^2 stream_years={mode}: {wall:.4f} s  steps {st['nsteps']} rejected {st['nrejected']} newton {st['nnewton']} "
              f"launches {st['nlaunch']} commands {cmds} kernel starts {kl} timeouts {eng.counter('stream_timeouts')}", flush=True)
    pr = [eng.counter(f"stream_prof_{i}") for i in range(12)]
    print(f"{n}^2 per workgroup over the stream years: waiting for commands {pr[0] / 1e3:.1f} ms, executing {pr[1] / 1e3:.1f} ms, "
          f"waiting for neighbours {pr[2] / 1e3:.1f} ms, {pr[3]} commands", flush=True)
    for k, name in enumerate(("SETUP", "NEWTON", "ERR", "BOUNDARY")):
        if pr[8 + k]:
            print(f"    {name}: {pr[8 + k]} commands, {pr[4 + k] / max(pr[8 + k], 1):.2f} us each", flush=True)
    print("    host controller: " + ", ".join(f"{k} {eng.counter(k)}" for k in ("spec_launches_dropped", "spec_front_launches_dropped",
                                                                               "err_estimates_queued", "err_estimates_dropped")), flush=True)
    same = np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    print(f"{n}^2 bit-identical: {same}; speed-up {res[0][0][0] / res[1][0][0]:.2f}x", flush=True)
    eng.close()
