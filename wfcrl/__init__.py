"""Drop-in import name of the reference package: `from wfcrl import environments as envs`,
`from wfcrl.rewards import StepPercentage`, `from wfcrl.interface import FlorisInterface` resolve to this build's
modules in `wfcrl-env_amd/` (HIP backend).  Nothing of the reference lives here — it is an alias, like
`wfcrl_env_amd`.  (The reference's own `wfcrl/__init__.py:1-13` only holds two stale gym registrations pointing at a
non-existent module; they are not replicated)"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "wfcrl-env_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
