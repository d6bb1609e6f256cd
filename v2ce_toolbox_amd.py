"""Import shim: makes the directory ``v2ce-toolbox_amd/`` (not a valid Python identifier)
importable as the package ``v2ce_toolbox_amd``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "v2ce-toolbox_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f
