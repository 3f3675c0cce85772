"""The oracle's Radau restatement reproduces scipy.integrate.solve_ivp(method="Radau")
driven with the reference's own tendency / Jacobian functions BIT FOR BIT (golden
`comp_fcn_*.npz`), including SciPy's nfev / njev / nlu counters; and step replay of the
recorded schedule is the identical map."""
import numpy as np
import pytest

from helpers import oracle_iage
from oracle import radau

CASES = [("20x3_columns", 20, 3, 0.0, 0.0), ("26x26", 26, 26, 0.1, 1000.0)]


@pytest.mark.parametrize("tag,nz,ny,vv,kh", CASES)
def test_radau_bitwise_and_replay(golden_dir, tag, nz, ny, vv, kh):
    g = np.load(f"{golden_dir}/comp_fcn_{tag}.npz")
    _, tm = oracle_iage(nz, ny, vv, kh)
    res, solver = radau.comp_fcn(tm, g["y0"], return_solver=True)
    assert np.array_equal(res, g["fcn"])
    st = solver.stats
    assert (st.nfev, st.njev, st.nlu) == (int(g["nfev"]), int(g["njev"]), int(g["nlu"]))
    if nz * ny < 100:
        replayed = radau.comp_fcn(tm, g["y0"], replay=solver.schedule)
        assert np.array_equal(replayed, res)


def test_radau_constants_match_scipy():
    from scipy.integrate._ivp import radau as sp

    for name in ("C", "E", "T", "TI", "P"):
        assert np.array_equal(getattr(radau, name), getattr(sp, name)), name
    assert radau.MU_REAL == sp.MU_REAL and radau.MU_COMPLEX == sp.MU_COMPLEX
