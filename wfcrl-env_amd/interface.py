"""`HipFlorisInterface` — drop-in for the reference's `FlorisInterface` (wfcrl/interface.py:444-671).

Same duck type, names, units and error behaviour; the FLORIS object (`self.fi`) is replaced by a
`WfStep` handle on libwfstep.so (HIP kernels, C ABI include/wfstep.h) evaluating ONE farm instance
(env_batch = 1).  The batched path for B >= 1 is `vec_env.VecWindFarmEnv`, which drives the same handle.
"""
from __future__ import annotations

import os
import warnings
from abc import ABC
from typing import List, Union

import numpy as np

from .backend import WfStep


class BaseInterface(ABC):
    """The reference's simulator 'plugin API' (wfcrl/interface.py:25-50): duck-typed, all no-ops."""

    def __init__(self):
        self.num_turbines = None

    @property
    def wind_speed(self):
        return None

    @property
    def wind_dir(self):
        return None

    def set_yaw_angles(self, yaws: List):
        return None

    def get_yaw_angles(self) -> List:
        return None

    def avg_powers(self) -> List:
        return None

    def init(self):
        return None

    def next_wind(self):
        return None


def _load_time_series(ts):
    """CSV path (first line is a header, as pandas.read_csv) or ndarray -> (T, 2) float array
    [speed, direction] (wfcrl/interface.py:513-515; ndarray accepted too)."""
    if isinstance(ts, str):
        try:
            import pandas as pd

            ts = pd.read_csv(ts).values
        except ImportError:  # pragma: no cover
            ts = np.loadtxt(ts, delimiter=",", skiprows=1, ndmin=2)
    ts = np.asarray(ts)
    assert isinstance(ts, np.ndarray) and ts.ndim == 2 and ts.shape[1] >= 2
    return ts


class HipFlorisInterface(BaseInterface):
    CONTROL_SET = ["yaw"]

    # column of `current_measures` per measure (interface.py:448-454)
    DEFAULT_MEASURE_MAP = {
        "yaw": 0,
        "wind_speed": 1,
        "wind_direction": 2,
        "load": [3, 4, 5, 6],
        "freewind_measurements": None,
    }

    _REF_ARGS = ("simul_file", "max_iter", "log_file", "wind_speed", "wind_direction", "wind_time_series")
    _OWN_ARGS = ("xcoords", "ycoords", "max_iter", "log_file", "wind_speed", "wind_direction", "wind_time_series",
                 "device_id", "model", "seed", "risk_resolve")

    def __init__(self, num_turbines: int, *args, **kw):
        """Two call forms.

        Reference form (wfcrl/interface.py:462-471), selected when the second argument is a path:
            FlorisInterface(num_turbines, simul_file, max_iter=1e4, log_file=None, wind_speed=None,
                            wind_direction=None, wind_time_series=None)
        reads layout, wind and model constants from the FLORIS case.yaml (`simul_utils.load_case_yaml`); `wind_speed` /
        `wind_direction` None mean "what the file holds" (the reference passes the None on to `update_wind`, which
        raises TypeError at `None % 360`, interface.py:664 — its callers always go through `from_case` with explicit
        values); in the coordinate form None means the template's 8 m/s / 270 deg (case.yaml:31-36).

        Coordinate form of this build:
            HipFlorisInterface(num_turbines, xcoords, ycoords, max_iter=1e4, log_file=None, wind_speed=None,
                               wind_direction=None, wind_time_series=None, device_id=0, model=None, seed=None,
                               risk_resolve=True)
        risk_resolve (default on for this single-farm drop-in): a step the float32 kernel flags — a deficit within rounding
        of the overlap threshold, a turbine on the cut-in ramp or the cut-out drop — is solved again in float64 on the
        device (include/wfstep.h: wf_set_risk_resolve), as the reference computes every step (interface.py:564); costs three
        empty launches per step otherwise.
        """
        ref_form = (len(args) > 0 and isinstance(args[0], (str, os.PathLike))) or "simul_file" in kw
        names = self._REF_ARGS if ref_form else self._OWN_ARGS
        if len(args) > len(names):
            raise TypeError(f"{type(self).__name__}() takes at most {len(names) + 1} positional arguments")
        bound = dict(zip(names, args))
        for k, v in kw.items():
            if k in bound:
                raise TypeError(f"{type(self).__name__}() got multiple values for argument '{k}'")
            if k not in names and k not in self._OWN_ARGS[7:]:
                raise TypeError(f"{type(self).__name__}() got an unexpected keyword argument '{k}'")
            bound[k] = v
        if ref_form:
            from .simul_utils import load_case_yaml

            c = load_case_yaml(bound.pop("simul_file"))
            bound["xcoords"], bound["ycoords"] = c["xcoords"], c["ycoords"]
            bound.setdefault("model", c["model"])
            if bound.get("wind_speed") is None:
                bound["wind_speed"] = c["speed"]
            if bound.get("wind_direction") is None:
                bound["wind_direction"] = c["direction"]
        if bound.get("wind_time_series") is None:
            if bound.get("wind_speed") is None:
                bound["wind_speed"] = 8.0
            if bound.get("wind_direction") is None:
                bound["wind_direction"] = 270.0
        missing = [k for k in ("xcoords", "ycoords") if k not in bound]
        if missing:
            raise TypeError(f"{type(self).__name__}() missing required argument(s): {', '.join(missing)}")
        self._construct(num_turbines, **bound)

    def _construct(self, num_turbines: int, xcoords, ycoords, max_iter: int = int(1e4), log_file: str = None,
                   wind_speed: float = None, wind_direction: float = None,
                   wind_time_series: Union[str, np.ndarray] = None, device_id: int = 0, model: dict = None,
                   seed: int = None, risk_resolve: bool = True):
        BaseInterface.__init__(self)
        if len(xcoords) != num_turbines or len(ycoords) != num_turbines:
            raise ValueError("xcoords and ycoords layout coordinates must have num_turbines entries")
        self.num_turbines = num_turbines
        self.fi = self._make_backend(xcoords, ycoords, device_id, model)
        if hasattr(self.fi, "set_risk_resolve"):  # (on is also the default of the handle itself)
            self.fi.set_risk_resolve(1 if risk_resolve else 0)
        self.measure_map = self.DEFAULT_MEASURE_MAP
        self._num_measures = sum(len(v) if isinstance(v, list) else 1 for v in self.measure_map.values()) - 1
        self.dt = 60
        self.max_iter = max_iter
        self._logging = False
        self._ws = None  # what FLORIS' flow_field would hold
        self._wd = None
        self._wind_dirty = True
        self._rng = np.random.default_rng(seed) if seed is not None else None
        self._has_series = wind_time_series is not None
        self.wind_time_series = wind_time_series
        self.wind_generator = self._make_wind_generator(wind_speed, wind_direction, wind_time_series)
        ws0, wd0 = next(self.wind_generator)
        self.init(ws0, wd0)
        if log_file is not None:
            self._log_file = log_file
            self._logging = True

    # -- construction -----------------------------------------------------------------------------
    def _make_backend(self, xcoords, ycoords, device_id, model):
        """The device handle.  Always the HIP library: there is no CPU fallback in the product (the CPU
        test-suite substitutes an oracle-backed stand-in by overriding this hook in a tests/ subclass)."""
        return WfStep(xcoords, ycoords, env_batch=1, device_id=device_id, model=model)

    @classmethod
    def from_case(cls, case, log_file: str = None, output_dir: str = None, **kw):
        """interface.py:526-547.  No case.yaml is written on this path; pass
        `output_dir` to get one for inspection (`simul_utils.dump_case_yaml`)."""
        p = case.simul_params
        if output_dir is not None:
            from .simul_utils import dump_case_yaml

            dump_case_yaml(case.dict(), output_dir)
        return cls(case.num_turbines, xcoords=p["xcoords"], ycoords=p["ycoords"],
                   max_iter=case.max_iter, log_file=log_file, wind_speed=float(p["speed"]),
                   wind_direction=float(p["direction"]), wind_time_series=p["wind_time_series"], **kw)

    @classmethod
    def from_yaml(cls, simul_file, max_iter: int = int(1e4), log_file: str = None, wind_time_series=None, **kw):
        """The reference's constructor path `FlorisInterface(num_turbines, simul_file, ...)` (interface.py:462-479):
        layout, wind and model constants are read from a FLORIS v3 case.yaml (simul_utils.load_case_yaml)."""
        from .simul_utils import load_case_yaml

        c = load_case_yaml(simul_file)
        return cls(len(c["xcoords"]), xcoords=c["xcoords"], ycoords=c["ycoords"], max_iter=max_iter,
                   log_file=log_file, wind_speed=c["speed"], wind_direction=c["direction"],
                   wind_time_series=wind_time_series, model=c["model"], **kw)

    def _make_wind_generator(self, wind_speed=None, wind_direction=None, time_series=None):
        """Constant infinite generator, or a finite playback of the series rolled to a random start
        (interface.py:503-524: global np.random there; seedable here via `seed=`)."""
        if time_series is None:
            def gen():
                while True:
                    yield wind_speed, wind_direction
            return gen()
        ts = _load_time_series(time_series)
        start = int(self._rng.integers(0, ts.shape[0])) if self._rng is not None else np.random.randint(0, ts.shape[0])
        rolled = np.r_[ts[start:], ts[:start]]

        def gen():
            for row in rolled:
                yield row
        return gen()

    # -- state ------------------------------------------------------------------------------------
    @property
    def wind_speed(self):
        return self._ws

    @property
    def wind_dir(self):
        return self._wd

    def init(self, wind_speed: float = None, wind_direction: float = None):
        """interface.py:588-613."""
        if self._has_series and wind_speed is not None:
            warnings.warn(f"Wind speed = {wind_speed} requested, but wind_time_series mode is activated. "
                          "Request will be ignored.")
            wind_speed = None
        if self._has_series and wind_direction is not None:
            warnings.warn(f"Wind direction = {wind_direction} requested, but wind_time_series mode is activated. "
                          "Request will be ignored.")
            wind_direction = None
        self.wind_generator = self._make_wind_generator(wind_speed, wind_direction, self.wind_time_series)
        self.update_wind(*next(self.wind_generator))
        self._num_iter = 0
        self._current_yaw_command = np.zeros((1, 1, self.num_turbines))
        self.current_measures = np.full((self.num_turbines, self._num_measures), np.nan)
        self._powers = np.full(self.num_turbines, np.nan)

    def update_wind(self, wind_speed: float = None, wind_direction: float = None):
        """interface.py:663-671: wd % 360 (TypeError on None, as in the reference); a None speed keeps the
        current one (FLORIS `reinitialize(wind_speeds=None)`).  The device-side geometry (rotation +
        sort) is redone only when (ws, wd) actually changed."""
        wind_direction = wind_direction % 360
        if wind_speed is None:
            wind_speed = self._ws
        if wind_speed != self._ws or wind_direction != self._wd:
            self._ws, self._wd = float(wind_speed), float(wind_direction)
            self._wind_dirty = True

    # -- the step ---------------------------------------------------------------------------------
    def update_command(self, yaw: np.ndarray = None):
        """interface.py:557-586: set yaw, advance wind, solve, fill the (N, 7) measure matrix."""
        if yaw is not None:
            self._current_yaw_command[0, 0, :] = np.asarray(yaw).astype(np.double)
        self.update_wind(*next(self.wind_generator))  # StopIteration when a finite series is exhausted
        if self._wind_dirty:
            self.fi.set_wind(self._ws, self._wd)
            self._wind_dirty = False
        out = self.fi.step(self._current_yaw_command.reshape(1, -1).astype(np.float32))
        m = self.current_measures
        m[:, self.measure_map["yaw"]] = self._current_yaw_command[0, 0]
        m[:, self.measure_map["wind_speed"]] = out["wind_speed"][0]
        m[:, self.measure_map["wind_direction"]] = out["wind_direction"][0]
        # the reference stores load*1e7 and its consumer divides by 1e7 in place (interface.py:575-577,
        # mdp.py:281-283); same convention here so WindFarmMDP works unmodified
        m[:, self.measure_map["load"]] = out["load"][0].astype(np.float64) * 1e7
        self._powers = out["power"][0].astype(np.float64)
        self._num_iter += 1
        if self._logging:
            with open(self._log_file, "a") as fp:
                fp.write(f"Sent command YAW {self.get_yaw_command()} - ***********Received Power: {self.avg_powers()}"
                         f" Wind : {self.avg_wind()}\n")
        return self._num_iter == self.max_iter

    # -- accessors (interface.py:615-655) -----------------------------------------------------------
    def get_yaw_command(self):
        return self._current_yaw_command.copy().flatten()

    def avg_powers(self) -> np.ndarray:
        """Per-turbine power [W] of the last solve (fi.get_turbine_powers().flatten())."""
        return self._powers.copy()

    def avg_farm_power(self):
        return self.avg_powers().sum()

    def avg_wind(self) -> np.ndarray:
        return np.array([self.wind_speed, self.wind_dir]).squeeze()

    def local_load_proxies(self):
        l = self.current_measures[:, self.measure_map["load"]] / 1e7
        return l[:, 0], l[:, 1], l[:, 2], l[:, 3]

    def local_wind_measurements(self):
        return (self.current_measures[:, self.measure_map["wind_speed"]],
                self.current_measures[:, self.measure_map["wind_direction"]])

    def get_measure(self, measure: str):
        if measure not in self.measure_map:
            return None
        if measure == "freewind_measurements":
            return self.avg_wind()
        return self.current_measures[:, self.measure_map[measure]]  # "load": fancy index -> fresh array

    def get_parameters(self):
        return None

    def sample_parameters(self):
        return None


# The registry keeps the reference's simulator name working: "<Layout>_Floris" is served by the HIP backend.
FlorisInterface = HipFlorisInterface


class FastFarmInterface(BaseInterface):  # pragma: no cover - out of scope
    def __init__(self, *a, **k):
        raise NotImplementedError("the FAST.Farm/MPI backend of the reference is out of scope of this build")

    @classmethod
    def from_case(cls, *a, **k):
        raise NotImplementedError("the FAST.Farm/MPI backend of the reference is out of scope of this build")
