"""Import shim: the package directory is named ``sbm-bp_amd`` (not a valid Python identifier), so
``import sbm_bp_amd`` resolves here and this module turns itself into that package."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "sbm-bp_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
