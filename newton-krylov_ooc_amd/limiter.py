"""Per-region limiter of a Newton increment (host side, once per Newton iteration).

Reference: `utils.min_by_region`, `utils.comp_scalef_lob`, `utils.comp_scalef_upb`
(`nk_ooc/utils.py:540-600`) as used by `TracerModuleStateBase.apply_limiter`
(`nk_ooc/tracer_module_state_base.py:112-151`): the largest factor in (0, 1], region by region, such
that `base + factor * increment` stays on the allowed side of a bound.
"""

import numpy as np


def region_min(region_cnt, region_mask, vals):
    """minimum of vals over the cells of every region (mask values 1..region_cnt); a region without
    cells gives +inf, as np.amin(..., initial=inf, where=...) does in the reference"""
    res = np.full(region_cnt, np.inf)
    for reg in range(region_cnt):
        sel = region_mask == reg + 1
        if sel.any():
            res[reg] = vals[sel].min()
    return res


def scalef_for_bound(region_cnt, region_mask, base, increment, bound, upper):
    """factor per region keeping base + factor * increment >= bound (upper=False) or <= bound (upper=True).
    Ones when there is no bound or nothing crosses it; ValueError when base itself is on the wrong side."""
    if bound is None:
        return np.ones(region_cnt)
    sign = 1.0 if upper else -1.0
    crossing = sign * (base + increment - bound) > 0.0
    if not crossing.any():
        return np.ones(region_cnt)
    if (sign * (base - bound) > 0.0).any():
        raise ValueError("base > upb" if upper else "base < lob")
    ratio = np.ones(np.shape(base))
    np.divide(bound - base, increment, out=ratio, where=crossing)
    return region_min(region_cnt, region_mask, ratio)
