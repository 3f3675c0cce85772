"""CPU oracle for the py_driver_2d Krylov/JVP hot path.

TEST INFRASTRUCTURE ONLY.  This package is a NumPy/SciPy restatement of the
reference algorithm (klindsay28/Newton-Krylov_OOC, `nk_ooc/`), written so that
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
can check the HIP path against it.  Nothing under `newton-krylov_ooc_amd/`
imports it, and the product path never falls back to it.

Parity pin: every function here is checked against golden vectors produced by
importing the genuine reference numerics in the build container
(`tests/golden/gen_golden.py`, fixtures in `tests/golden/*.npz`) and against
the reference's own committed baselines (`tests/golden/ref_baselines/`).

Third-party arithmetic: the reference delegates the time integration to
SciPy's Radau IIA (`scipy.integrate.solve_ivp(method="Radau")`, reference pin
scipy=1.9.1, this image scipy 1.15.3) and SuperLU (`scipy.sparse.linalg`).
`oracle.radau` restates that controller step for step and is pinned bit for
bit against `scipy.integrate.solve_ivp` itself in `tests/test_oracle_radau.py`.
"""
