"""Import alias: the product package lives in ``anomaly-detection-super-resolution_amd/`` (the
layout the build contract names); a hyphenated directory is not a Python identifier, so this
shim makes it importable as ``srad_amd`` by pointing ``__path__`` at it and running its
``__init__``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "anomaly-detection-super-resolution_amd")
__path__ = [_real]
_init = _os.path.join(_real, "__init__.py")
with open(_init) as _f:
    exec(compile(_f.read(), _init, "exec"))
del _os, _f, _init
