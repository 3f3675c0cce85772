"""probe: frozen forward years with the next step's line factorisation computed by the launch that ends the step
(option "prefactor" 1, the default) against the factorising first launch of every step (0): time, launches, and
whether either year of the recorded state is the recorded year, bit for bit"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

for n in [int(v) for v in (sys.argv[1:] or ["416"])]:
    eng = iage_engine(Grid2d.default(n, n))
    if os.environ.get("PROBE_TEAM") is not None:
        eng.set_option("team", float(os.environ["PROBE_TEAM"]))
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
    x = eng.axpby(1.0, x, 1.0, eng.comp_fcn(x)[0])
    fx, st, sched = eng.comp_fcn(x, record=True)
    want = eng.download(fx)
    rng = np.random.default_rng(0)
    xp = eng.upload(eng.download(x) * (1.0 + 1.0e-4 * rng.standard_normal(eng.shape)))
    ref = None
    only = os.environ.get("PROBE_ONLY")
    for pre in ((0, 1, 0, 1, 0, 1) if only is None else (int(only),) * 4):
        eng.set_option("prefactor", pre)
        fx2, st2 = eng.comp_fcn_frozen(x, sched)
        same = np.array_equal(eng.download(fx2), want)
        fx3, st3 = eng.comp_fcn_frozen(xp, sched)
        got = eng.download(fx3)
        ref = got if ref is None else ref
        print(f"n={n} prefactor={pre}: frozen year {st2['seconds']:.4f} s / {st3['seconds']:.4f} s, {st2['nlaunch']} launches, "
              f"steps {st2['nsteps']}, Newton {st2['nnewton']}; free-running year {st['seconds']:.4f} s; "
              f"recorded state bit-identical: {same}; perturbed state identical to prefactor=0: {np.array_equal(got, ref)}",
              flush=True)
    eng.close()
