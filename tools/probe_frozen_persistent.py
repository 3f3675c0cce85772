"""probe: frozen years of the small grids launch by launch (option "frozen_persistent" 0) and as one launch on the schedule
cache (1), there with a wave or a four-wave team per column: seconds per year, cache build time, bit-identity"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

for n in [int(v) for v in (sys.argv[1:] or ["26", "52", "104", "208"])]:
    eng = iage_engine(Grid2d.default(n, n))
    eng.set_option("frozen_persistent_max_e", 8)
    eng.set_option("frozen_cache_gb", 128.0)
    eng.set_option("frozen_cache_after", 0)
    eng.set_option("frozen_err_check", 0)     # (the perturbed state below is white noise: not a state the recorded steps control)
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
    x = eng.axpby(1.0, x, 1.0, eng.comp_fcn(x)[0])
    fx, st, sched = eng.comp_fcn(x, record=True)
    want = eng.download(fx)
    rng = np.random.default_rng(0)
    xp = eng.upload(eng.download(x) * (1.0 + 1.0e-4 * rng.standard_normal(eng.shape)))
    ref = None
    for flag, team, xcd, nbs in ((0, 0, 1, 0), (1, 0, 1, 0), (1, 1, 1, 0), (1, 1, 1, 1), (1, 0, 0, 0), (1, 0, 1, 0), (1, 1, 1, 0), (1, 1, 1, 1)):
        eng.set_option("frozen_persistent", flag)
        eng.set_option("frozen_team", team)
        eng.set_option("frozen_xcd", xcd)
        eng.set_option("frozen_nbsync", nbs)
        builds = eng.counter("frozen_cache_builds")
        t0 = time.perf_counter()
        fx2, st2 = eng.comp_fcn_frozen(x, sched)
        t_first = time.perf_counter() - t0
        same = np.array_equal(eng.download(fx2), want)
        fx3, st3 = eng.comp_fcn_frozen(xp, sched)
        got = eng.download(fx3)
        ref = got if ref is None else ref
        print(f"n={n} frozen_persistent={flag} team={team} xcd={xcd} nbsync={nbs} (team years {eng.counter('frozen_team_years')}, XCD years {eng.counter('frozen_xcd_years')}): year {st2['seconds']*1e3:.2f} ms / {st3['seconds']*1e3:.2f} ms "
              f"(first call {t_first*1e3:.2f} ms, cache builds {eng.counter('frozen_cache_builds') - builds}), "
              f"{st3['nlaunch']} launches, {st3['nnewton']} Newton iterations in {len(sched)} steps; "
              f"one-launch years so far {eng.counter('frozen_persistent_years')}; free-running year {st['seconds']*1e3:.1f} ms; "
              f"recorded state bit-identical: {same}; perturbed identical to launch path: {np.array_equal(got, ref)}", flush=True)
    eng.close()
