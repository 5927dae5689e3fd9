"""Import shim: the package directory is ``dpcr-agb_amd/`` (hyphenated, not a Python identifier);
``import dpcr_agb_amd`` resolves to it by pointing this module's ``__path__`` at that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "dpcr-agb_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _os, _f
