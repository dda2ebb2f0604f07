"""Test-only stand-ins.  `OracleStep` has the call surface of wfcrl_env_amd.backend.WfStep but evaluates
the float64 C oracle on the CPU, so the host-side env surface (interface, mdp, envs, wrappers, registry)
can be exercised in the CPU suite.  It lives under tests/ — the product has no CPU path."""
import numpy as np

from oracle import c_oracle
from wfcrl_env_amd.interface import HipFlorisInterface


class OracleStep:
    def __init__(self, xcoords, ycoords, env_batch=1, device_id=0, model=None):
        self.x, self.y = np.asarray(xcoords, float), np.asarray(ycoords, float)
        self.num_turbines, self.env_batch = len(self.x), env_batch
        self.ws = self.wd = None
        self.calls = 0
        self.wind_sets = 0

    def set_wind(self, ws, wd):
        self.ws, self.wd = np.atleast_1d(np.asarray(ws, float)), np.atleast_1d(np.asarray(wd, float))
        self.wind_sets += 1

    def step(self, yaw, out=None):
        self.calls += 1
        yaw = np.asarray(yaw, np.float32).reshape(self.env_batch, self.num_turbines).astype(np.float64)
        r = c_oracle.farm_step_batch(self.x, self.y, self.ws, self.wd, yaw, nthreads=1)
        return {k: v.astype(np.float32) for k, v in r.items()}

    def sync(self):
        pass

    def close(self):
        pass


class OracleFlorisInterface(HipFlorisInterface):
    def _make_backend(self, xcoords, ycoords, device_id, model):
        return OracleStep(xcoords, ycoords)


class OracleStep64(OracleStep):
    """The same stand-in handing back float64 (no float32 device surface in between): what the reference's
    FLORIS object hands its interface.  Used where host-side arithmetic is compared at 1e-12."""

    def step(self, yaw, out=None):
        self.calls += 1
        yaw = np.asarray(yaw, np.float32).reshape(self.env_batch, self.num_turbines).astype(np.float64)
        return dict(c_oracle.farm_step_batch(self.x, self.y, self.ws, self.wd, yaw, nthreads=1))


class OracleFlorisInterface64(HipFlorisInterface):
    def _make_backend(self, xcoords, ycoords, device_id, model):
        return OracleStep64(xcoords, ycoords)
