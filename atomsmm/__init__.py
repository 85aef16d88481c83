"""`import atomsmm` -- the reference's import name (src/atomsmm/__init__.py:3-43), served by the MI355X-native
package `atomsmm_amd`: same classes, same sub-modules (`atomsmm.forces`, `atomsmm.propagators`, ...).  Put the
repository root on `sys.path` and a script written for AtomsMM runs unchanged; nothing is re-implemented here."""
import sys

import atomsmm_amd as _impl
from atomsmm_amd import *  # noqa: F401,F403
from atomsmm_amd import __all__, __version__  # noqa: F401

for _name in ('forces', 'integrators', 'propagators', 'systems', 'utils', 'computers'):
    _module = getattr(_impl, _name, None) or __import__('atomsmm_amd.' + _name, fromlist=[_name])
    sys.modules[__name__ + '.' + _name] = _module
    globals()[_name] = _module
InputError = _impl.InputError
