"""probe: how noisy is the finite-difference Jacobian-vector product (F(x + sigma v) - F(x)) / sigma, sigma = 1e-4 |x|,
when both years run freely under the adaptive controller -- by integrator mode.  Reference: the same product from years
integrated 10^4 times tighter.  Both the relative error of the product and of <v, w> / <v, v> (what the first Hessenberg
entry of a Krylov solve sees) are printed, for the first Krylov direction (preconditioned residual) and a random one.

    python tools/probe_jvp_noise.py [n ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

MODES = [("SciPy decisions (jac_fresh 0)", 0, -1), ("J at step start", 1, -1), ("J at stage 0", 1, 0),
         ("J at stage 1", 1, 1), ("J at stage 2", 1, 2)]

for n in [int(v) for v in (sys.argv[1:] or ["26", "52", "104"])]:
    grid = Grid2d.default(n, n)
    eng = iage_engine(grid)
    eng.set_option("stream_years", 0)       # (by launches)
    tight = iage_engine(grid, rtol=1.0e-10, atol=1.0e-10, lin_tol=1.0e-10)
    tight.set_option("stream_years", 0)       # (by launches)
    tight.set_option("jac_fresh", 0)
    rng = np.random.default_rng(n)
    col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x0 = np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy()
    x = eng.upload(x0)
    x = eng.axpby(1.0, x, 1.0, eng.comp_fcn(x)[0])
    xh = eng.download(x)
    xt = tight.upload(xh)
    fxt, _, _ = tight.comp_fcn(xt)
    eng.precond_setup()
    v1 = eng.download(eng.precond_apply(eng.upload(tight.download(fxt))))
    v1 /= np.sqrt(np.sum(v1 * v1))
    smooth = np.cumsum(np.cumsum(rng.standard_normal(xh.shape), axis=1), axis=2)
    v2 = smooth / np.sqrt(np.sum(smooth * smooth))
    for vname, v in (("preconditioned residual", v1), ("random smooth", v2)):
        w_ref = tight.download(tight.jvp(xt, fxt, tight.upload(v))[0])
        q_ref = np.sum(v * w_ref)
        print(f"n={n}, direction: {vname}; |w_ref| = {np.sqrt(np.sum(w_ref ** 2)):.3e}, <v, w_ref> = {q_ref:.6f}", flush=True)
        for name, fresh, stage in MODES:
            eng.set_option("jac_fresh", fresh)
            eng.set_option("jac_stage", stage)
            fx, st, _ = eng.comp_fcn(x)
            w, _, stp = eng.jvp(x, fx, eng.upload(v))
            w = eng.download(w)
            err = np.sqrt(np.sum((w - w_ref) ** 2) / np.sum(w_ref ** 2))
            dq = (np.sum(v * w) - q_ref) / abs(q_ref)
            same = st["nsteps"] == stp["nsteps"] and st["nrejected"] == stp["nrejected"]
            print(f"    {name:32s}: |w - w_ref| / |w_ref| = {err:.3e}, d<v,w>/<v,w> = {dq:+.3e}; steps {st['nsteps']} / {stp['nsteps']}, "
                  f"rejected {st['nrejected']} / {stp['nrejected']}{'' if same else '  (sequences differ in length)'}", flush=True)
        # frozen controller: the perturbed year repeats the accepted steps of the year that produced F(x)
        for name, fresh, stage in MODES[:1] + MODES[3:4]:
            eng.set_option("jac_fresh", fresh)
            eng.set_option("jac_stage", stage)
            fx, st, sched = eng.comp_fcn(x, record=True)
            fx2, st2 = eng.comp_fcn_frozen(x, sched)
            same = np.array_equal(eng.download(fx2), eng.download(fx))
            eng.set_frozen_schedule(sched)
            w, _, stp = eng.jvp(x, fx, eng.upload(v))
            eng.set_frozen_schedule(None)
            w = eng.download(w)
            err = np.sqrt(np.sum((w - w_ref) ** 2) / np.sum(w_ref ** 2))
            dq = (np.sum(v * w) - q_ref) / abs(q_ref)
            print(f"    FROZEN, {name:24s}: |w - w_ref| / |w_ref| = {err:.3e}, d<v,w>/<v,w> = {dq:+.3e}; free year {st['seconds']:.4f} s, "
                  f"frozen year {stp['seconds']:.4f} s ({stp['nlaunch']} launches); frozen year of x itself bit-identical: {same}", flush=True)
    eng.close()
    tight.close()
