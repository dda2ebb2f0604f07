"""Drop-in import name of the reference package: `from wfcrl import environments as envs`,
`from wfcrl.rewards import StepPercentage`, `from wfcrl.interface import FlorisInterface` resolve to this build's
modules in `wfcrl-env_amd/` (HIP backend).  Nothing of the reference lives here — it is an alias of `wfcrl_env_amd`:
ONE module tree, so `wfcrl.rewards.StepPercentage is wfcrl_env_amd.rewards.StepPercentage` and isinstance checks hold
across the two names.  (The reference's own `wfcrl/__init__.py:1-13` only holds two stale gym registrations pointing at
a non-existent module; they are not replicated)"""
import importlib
import importlib.abc
import importlib.util
import sys

import wfcrl_env_amd as _real


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """`wfcrl.x.y` -> the module object of `wfcrl_env_amd.x.y` (imported once, under its canonical name)."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname == "wfcrl" or fullname.startswith("wfcrl."):
            return importlib.util.spec_from_loader(fullname, self, origin="wfcrl_env_amd" + fullname[len("wfcrl"):])
        return None

    def create_module(self, spec):
        return importlib.import_module(spec.origin)

    def exec_module(self, module):
        return None


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
sys.modules[__name__] = _real
