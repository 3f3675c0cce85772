"""shared helpers of the test-suite (oracle-side set-up)"""
import numpy as np

from oracle.grid import default_axes
from oracle.model import Iage, Py2dModel


def oracle_iage(nz, ny, max_abs_vvel=0.1, horiz_mix_coeff=1000.0):
    depth, ypos = default_axes(nz, ny)
    model = Py2dModel(depth, ypos, max_abs_vvel, horiz_mix_coeff)
    return model, Iage(model)


def rel_err(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


def free_years(eng, x, **kw):
    """one free-running forward year with SciPy's controller decision for decision (Jacobian reuse of
    radau.py:509-517, no memory of Newton failures: the mode whose counters are comparable with solve_ivp's)
    and one in the engine's default mode (Jacobian re-evaluated for every step attempt, at its second stage time
    after a Newton failure): ((fx, stats, sched), (fx, stats, sched))"""
    from nk_ooc_amd.engine import DEFAULT_GROWTH_CAP, DEFAULT_JAC_FRESH, DEFAULT_JAC_STAGE

    eng.set_option("jac_fresh", 0)
    eng.set_option("growth_cap", 0)
    eng.set_option("jac_stage", -1)
    try:
        faithful = eng.comp_fcn(x, **kw)
    finally:
        eng.set_option("jac_fresh", DEFAULT_JAC_FRESH)
        eng.set_option("growth_cap", DEFAULT_GROWTH_CAP)
        eng.set_option("jac_stage", DEFAULT_JAC_STAGE)
    return faithful, eng.comp_fcn(x, **kw)


def oracle_year_job(args):
    """one CPU year of the oracle on one BLAS thread, for worker processes of the parity tests (spawned: they never touch
    the GPU): args = (nz, ny, x, rows) -- a replay of the accepted steps `rows` from x, or (rows None) a free-running year"""
    from threadpoolctl import threadpool_limits

    from oracle import radau

    nz, ny, x, rows = args
    _, tm = oracle_iage(nz, ny)
    with threadpool_limits(limits=1):
        if rows is None:
            return radau.comp_fcn(tm, x)
        return radau.comp_fcn(tm, x, replay=rows)
