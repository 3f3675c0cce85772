"""Importable alias of the `newton-krylov_ooc_amd/` package directory.

The package directory name mandated for this repository contains a hyphen, which
Python's import statement cannot spell; this shim makes `import nk_ooc_amd`
(and `nk_ooc_amd.<submodule>`) resolve to the modules that live there.
"""
import os as _os

_real = _os.path.join(
    _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "newton-krylov_ooc_amd"
)
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
