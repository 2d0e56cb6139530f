"""Import alias for the hyphenated package directory ``siam-mot_amd/``.

``siam-mot_amd`` is not a legal Python identifier, so this one-file module turns
itself into a package whose submodules resolve from that directory:
``import siammot_amd.emm`` loads ``siam-mot_amd/emm.py``.
"""
import os as _os

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "siam-mot_amd")
__path__ = [_pkg_dir]
__package__ = "siammot_amd"
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__

with open(_os.path.join(_pkg_dir, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_pkg_dir, "__init__.py"), "exec"))
del _f
