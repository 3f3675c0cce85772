"""MI355X-native Krylov / finite-difference-JVP hot path of the py_driver_2d model.

Host side (Python) mirrors the reference plugin surface of
klindsay28/Newton-Krylov_OOC (`ModelState`, `KrylovSolver`, `SolverState`); the
numerics run in hand-written HIP kernels for gfx950 behind the C ABI declared in
`include/nk2d.h` (`csrc/libnk2d.so`, loaded with ctypes).  There is no CPU
fallback: importing the device layer without the built library raises.
"""

__version__ = "0.2.0"
