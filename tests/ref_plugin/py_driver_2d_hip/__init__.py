"""`nk_ooc.py_driver_2d_hip` -- the reference-side plugin of INTEGRATION.md, variant B.

A model plugin of klindsay28/Newton-Krylov_OOC is a package `nk_ooc.<model_name>` found by name
(`nk_ooc/model_state_base.py:627-667`).  This one puts the MI355X library under the reference's own
`nk_driver.py` / `NewtonSolver` / `KrylovSolver` / `SolverState`, which stay untouched:

    python -m nk_ooc.nk_driver --model_name py_driver_2d_hip --cfg_fnames ...,override.cfg

with `[DEFAULT] model_name = py_driver_2d_hip` in the override.  Everything the reference's py_driver_2d
model does on the host stays the reference's code (it is subclassed, not restated); the two numerical
kernels are re-routed:

* the `scipy.integrate.solve_ivp(..., "Radau", ...)` call of `ModelState.comp_fcn`
  (`nk_ooc/py_driver_2d/model_state.py:102-114`)            -> `nk2d_comp_fcn` / `nk2d_comp_fcn_hist`
* `iage.apply_precond_jacobian` (`nk_ooc/py_driver_2d/iage.py:66-93`)   -> `nk2d_precond_apply`

The package is installed by making it visible on `nk_ooc.__path__` (or by copying it into the reference
tree).  `tests/test_ref_dropin.py` runs the reference's real driver against it in the build container.
"""
