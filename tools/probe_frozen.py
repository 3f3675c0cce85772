"""probe: a forward year frozen on the steps of a free-running one (the perturbed year of a finite-difference product)
with a step boundary launch of its own (option "final_fuse" 0) and with every step ending in the launch of its last
Newton iteration (1, the default): time, launches, and whether the year of the recorded state is the recorded year"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

for n in [int(v) for v in (sys.argv[1:] or ["26", "104", "416"])]:
    eng = iage_engine(Grid2d.default(n, n))
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
    x = eng.axpby(1.0, x, 1.0, eng.comp_fcn(x)[0])
    fx, st, sched = eng.comp_fcn(x, record=True)
    want = eng.download(fx)
    for fuse in (0, 1, 0, 1):
        eng.set_option("final_fuse", fuse)
        fx2, st2 = eng.comp_fcn_frozen(x, sched)
        same = np.array_equal(eng.download(fx2), want)
        print(f"n={n} final_fuse={fuse}: frozen year {st2['seconds']:.4f} s, {st2['nlaunch']} launches, steps {st2['nsteps']}, "
              f"Newton {st2['nnewton']}; free-running year {st['seconds']:.4f} s, {st['nlaunch']} launches; "
              f"bit-identical to the recorded year: {same}", flush=True)
    eng.close()
