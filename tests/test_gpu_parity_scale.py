"""Parity of the forward year at the sizes the benchmark runs at, in BOTH controller modes.

* 52 x 52 and 104 x 104 against fixtures made by `scipy.integrate.solve_ivp` on the genuine reference
  functions (tests/golden/gen_golden.py `large`; the oracle's Radau restatement reproduced those two
  runs bit for bit and supplied the accepted-step schedules): step replay <= 1e-10, free-running year at
  the reference CI tolerance (atol 1e-6, rtol 1e-3, scripts/ci_py_driver_2d_iage.sh:38) with SciPy's
  decision sequence ("faithful": jac_fresh 0) and in the engines' default mode (Jacobian re-evaluated at
  every step start, inner tolerance 3e-2), counters of the faithful mode within 10 % of solve_ivp's.
* 416 x 416 (no CPU run finishes there: one Radau attempt costs the oracle 15 s): both modes against a
  GPU year integrated 1000 times tighter (rtol = atol = 1e-9, inner solves to 1e-10, SciPy's Jacobian
  reuse), i.e. against the converged solution of the same ODE, again at the CI tolerance.

Every comparison also records its margin -- max |got - ref| / (atol + rtol |ref|), 1.0 = the CI
tolerance -- into gpurun_out/r02_parity_margins.json (DESIGN.md section 5 quotes them)."""
import json
import os

import numpy as np
import pytest

from helpers import free_years, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MARGINS = os.path.join(ROOT, "gpurun_out", "r02_parity_margins.json")


def margin(got, ref, atol=1.0e-6, rtol=1.0e-3):
    got, ref = np.asarray(got).reshape(-1), np.asarray(ref).reshape(-1)
    return float(np.max(np.abs(got - ref) / (atol + rtol * np.abs(ref))))


def record(key, value):
    os.makedirs(os.path.dirname(MARGINS), exist_ok=True)
    data = json.load(open(MARGINS)) if os.path.exists(MARGINS) else {}
    data[key] = value
    with open(MARGINS, "w") as fptr:
        json.dump(data, fptr, indent=1, sort_keys=True)


def make_engine(n, **kw):
    from nk_ooc_amd.engine import iage_engine
    from nk_ooc_amd.grid import Grid2d

    return iage_engine(Grid2d.default(n, n, 0.1, 1000.0), **kw)


@pytest.mark.parametrize("n", [52, 104])
def test_replay_of_reference_schedule(golden_dir, n):
    g = np.load(f"{golden_dir}/comp_fcn_{n}x{n}.npz")
    eng = make_engine(n)
    fx, stats, _ = eng.comp_fcn(eng.upload(g["y0"]), replay=g["schedule"])
    got = eng.download(fx).reshape(-1)
    assert stats["nsteps"] == len(g["schedule"])
    err = rel_err(got, g["fcn"])
    record(f"replay_rel_err_{n}", err)
    assert err < 1e-10, err


@pytest.mark.parametrize("n", [52, 104])
def test_free_running_year_both_modes(golden_dir, n):
    g = np.load(f"{golden_dir}/comp_fcn_{n}x{n}.npz")
    eng = make_engine(n)
    (fx, stats, _), (fx_def, stats_def, _) = free_years(eng, eng.upload(g["y0"]))
    m_faithful = margin(eng.download(fx), g["fcn"])
    m_default = margin(eng.download(fx_def), g["fcn"])
    record(f"free_margin_{n}", {"faithful": m_faithful, "default": m_default,
                                "nfev": {"solve_ivp": int(g["nfev"]), "faithful": stats["nfev"],
                                         "default": stats_def["nfev"]},
                                "nsteps": {"faithful": stats["nsteps"], "default": stats_def["nsteps"]}})
    assert m_faithful < 1.0, m_faithful
    assert m_default < 1.0, m_default
    for key in ("nfev", "njev", "nlu"):
        assert abs(stats[key] - int(g[key])) <= 0.1 * int(g[key]) + 5, (key, stats[key], int(g[key]))


def test_tight_year_locates_the_reference_at_104(golden_dir):
    """how far solve_ivp's own tolerance-1e-6 year is from the converged solution of the ODE, in units of
    the CI tolerance: the yardstick for the GPU modes' margins"""
    g = np.load(f"{golden_dir}/comp_fcn_104x104.npz")
    eng = make_engine(104, rtol=1.0e-9, atol=1.0e-9, lin_tol=1.0e-10)
    eng.set_option("jac_fresh", 0)
    fx, stats, _ = eng.comp_fcn(eng.upload(g["y0"]))
    m_ref = margin(g["fcn"], eng.download(fx))
    record("tight_vs_solve_ivp_104", {"margin": m_ref, "nsteps_tight": stats["nsteps"]})
    assert m_ref < 1.0, m_ref


def test_416_modes_against_converged_year():
    """the size the benchmark is quoted on.  State = gen_init_iterate + one forward year (bench.py's
    iterate); reference = the same engine integrating 1000 times tighter with SciPy's Jacobian reuse."""
    n = 416
    eng = make_engine(n)
    col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
    x = eng.axpby(1.0, x, 1.0, eng.comp_fcn(x)[0])
    (fx, stats, _), (fx_def, stats_def, _) = free_years(eng, x)
    x_host = eng.download(x)
    faithful, default = eng.download(fx), eng.download(fx_def)
    eng.close()
    tight = make_engine(n, rtol=1.0e-9, atol=1.0e-9, lin_tol=1.0e-10)
    tight.set_option("jac_fresh", 0)
    fx_t, stats_t, _ = tight.comp_fcn(tight.upload(x_host))
    ref = tight.download(fx_t)
    m_faithful, m_default = margin(faithful, ref), margin(default, ref)
    record("free_margin_416_vs_tight", {
        "faithful": m_faithful, "default": m_default, "faithful_vs_default": margin(default, faithful),
        "nsteps": {"tight": stats_t["nsteps"], "faithful": stats["nsteps"], "default": stats_def["nsteps"]},
        "seconds": {"tight": stats_t["seconds"], "faithful": stats["seconds"], "default": stats_def["seconds"]}})
    assert m_faithful < 1.0, m_faithful
    assert m_default < 1.0, m_default
