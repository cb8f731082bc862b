"""Import alias for the ``fgs-nerf_amd/`` source directory (a hyphen cannot appear in a module name).

``import fgs_nerf_amd.grid`` loads ``fgs-nerf_amd/grid.py``: this package only redirects ``__path__``.
"""
import os as _os

_src = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "fgs-nerf_amd")
__path__ = [_src]
with open(_os.path.join(_src, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_src, "__init__.py"), "exec"))
