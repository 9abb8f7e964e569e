"""Import shim: the product package lives in ``uncertainty-nerf-gs_amd/`` (the name the build
contract asks for, not a valid Python identifier).  This package forwards its ``__path__``
there so that ``import uncertainty_nerf_gs_amd.ops`` resolves to
``uncertainty-nerf-gs_amd/ops.py``.  No code lives here."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "uncertainty-nerf-gs_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
