"""Import alias: ``import ntmtrack`` resolves to the ``ntm-tracker_amd/`` package
directory (whose name is not a valid Python identifier)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "ntm-tracker_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
