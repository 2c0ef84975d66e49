"""Import shim: ``import nerfart_amd`` resolves to the sources in ``nerf-art_amd/``.

The product directory carries the reference's name (``nerf-art_amd``), which is not a
valid Python identifier; this package simply points its ``__path__`` there.
"""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
_src = _os.path.join(_os.path.dirname(_here), "nerf-art_amd")
__path__ = [_src]
with open(_os.path.join(_src, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_src, "__init__.py"), "exec"))
del _f
