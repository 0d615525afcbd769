"""Import shim: the package directory is ``cer-mvs_amd/`` (a hyphen is not a valid
Python identifier), so ``import cer_mvs_amd`` resolves here and is re-pointed at that
directory.  Nothing else lives in this file."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "cer-mvs_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__, "r") as _f:
    exec(compile(_f.read(), __file__, "exec"))
