"""Import alias: the package lives in the directory `wfcrl-env_amd/` (not a valid Python identifier).

`import wfcrl_env_amd` resolves sub-modules from that directory.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "wfcrl-env_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
