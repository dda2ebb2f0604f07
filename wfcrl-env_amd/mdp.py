"""`WindFarmMDP` — spaces, yaw state transition, reset wind sampling (reference wfcrl/mdp.py:19-319).

Host-side mirror of the reference's MDP so that the Gymnasium / PettingZoo surface behaves the same
on top of `HipFlorisInterface`.  The batched, device-resident restatement of the same transition is
`vec_env.VecWindFarmEnv`.
"""
from __future__ import annotations

import copy
from collections import OrderedDict
from collections.abc import Iterable
from warnings import warn

import numpy as np

from ._compat import spaces
from .interface import BaseInterface

# wind sampled at reset (mdp.py:237-258)
WEIBULL_SHAPE, WEIBULL_SCALE = 8, 8
WD_MEAN, WD_STD = 270, 20


def clip_to_dict_space(element: dict, space) -> dict:
    for key in element:
        element[key] = np.clip(element[key], space[key].low, space[key].high)
    return element


class WindFarmMDP:
    CONTROL_SET = ["yaw", "pitch", "torque"]
    POSSIBLE_STATE_ATTRIBUTES = ["freewind_measurements", "wind_speed", "wind_direction", "yaw", "pitch", "torque"]
    DEFAULT_BOUNDS = {
        "wind_speed": [3, 28],
        "wind_direction": [0, 360],
        "yaw": [-40, 40],
        "pitch": [0, 360],
        "torque": [-1e5, 1e5],
    }
    ACTUATORS_RATE = {"yaw": 0.3, "pitch": 8}  # deg/s

    def __init__(self, interface, farm_case, controls: dict, continuous_control: bool = True, start_iter: int = 0,
                 horizon: int = int(1e6)):
        farm_case = farm_case.clone() if hasattr(farm_case, "clone") else copy.copy(farm_case)
        farm_case.max_iter = horizon
        if isinstance(interface, BaseInterface):
            warn("Interface already instantiated. Simulation arguments from `Farm case` will be ignored.")
            self.interface = interface
        else:
            extra = farm_case.interface_kwargs.get("path_to_simulator", None)
            self.interface = interface.from_case(farm_case) if extra is None else interface.from_case(farm_case, extra)
        self.num_turbines = farm_case.num_turbines
        self.continuous_control = continuous_control
        self.horizon = horizon
        self.start_iter = start_iter
        self.farm_case = farm_case

        self._check_controls(controls)
        self.controls = controls
        self.num_controls = len(controls)
        # everything not controlled is measured, if the interface can measure it
        self.measures = [a for a in self.POSSIBLE_STATE_ATTRIBUTES
                         if a not in controls and a in self.interface.measure_map]
        self.state_attributes = list(controls) + self.measures

        n = self.num_turbines
        if continuous_control:
            self.action_space = spaces.Dict({name: spaces.Box(-spec[2], spec[2], shape=(n,))
                                             for name, spec in controls.items()})
        else:
            # 0 / 1 / 2 = down / hold / up
            self.action_space = spaces.Dict({name: spaces.MultiDiscrete([3] * n) for name in controls})

        ones = np.ones(n, dtype=np.float32)
        ws_lo, ws_hi = self.DEFAULT_BOUNDS["wind_speed"]
        wd_lo, wd_hi = self.DEFAULT_BOUNDS["wind_direction"]
        boxes = OrderedDict()
        for attr in self.state_attributes:
            if attr == "freewind_measurements":
                lo = np.array([ws_lo, wd_lo], dtype=np.float32)
                hi = np.array([ws_hi, wd_hi], dtype=np.float32)
            else:
                b = controls[attr] if attr in controls else self.DEFAULT_BOUNDS[attr]
                lo, hi = ones * b[0], ones * b[1]
            boxes[attr] = spaces.Box(lo, hi, shape=lo.shape)
        self.state_space = spaces.Dict(boxes)
        self.start_state = None
        self._actuation_accumulator = self._zero_accumulator()

    # -- helpers ------------------------------------------------------------------------------------
    def _zero_accumulator(self):
        return {c: np.zeros(self.num_turbines, dtype=np.float32) for c in self.controls}

    def get_state_powers(self):
        return self.interface.avg_powers()

    def get_accumulated_actions(self, agent=None):
        return self._actuation_accumulator.copy()

    def _cast_dict_array(self, state):
        return OrderedDict((k, v.astype(np.float32)) for k, v in state.items())

    def _check_controls(self, control_dict: dict):
        for name, spec in control_dict.items():
            if name not in self.CONTROL_SET:
                raise ValueError(f"Cannot control {name}. Allowed controls are {self.CONTROL_SET}")
            if name not in self.interface.CONTROL_SET:
                raise ValueError(f"Cannot control `{name}`. Interface {self.interface.__class__.__name__}"
                                 f" only allows for the following: {self.interface.CONTROL_SET}")
            if not (isinstance(spec, Iterable) and 2 <= len(spec) <= 3):
                raise TypeError(f"Wrong bounds for actuator {name}: Bounds on actuators must be an iterable of the type"
                                " [lower_bound, upper_bound] if control is continuous and"
                                " [lower_bound, upper_bound, step_size] otherwise")
            if not spec[0] < spec[1]:
                raise ValueError(f"Wrong bounds for actuator {name}: ensure that lower_bound < upper_bound")
            if len(spec) == 2:
                control_dict[name] = tuple(spec) + (1,)
                warn(f"No step size was provided for actuator {name}. Step size will default to 1.")
            elif not self.continuous_control and spec[2] <= 0:
                raise ValueError(f"Invalid step size provided for actuator {name} the step size must be stricly positive")

    def _check_state(self, state: dict):
        for attr, value in state.items():
            if attr not in self.state_attributes:
                raise ValueError(f"Unknwon attribute {attr} in state dict. Accepted attributed are: {self.state_attributes}")
            if not isinstance(value, np.ndarray):
                raise TypeError(f"State attribute {attr} must be a numpy array. Received {type(value)}")
            if attr != "freewind_measurements" and value.shape != (self.num_turbines,):
                raise TypeError(f"State attribute {attr} must be of shape (NUM_TURBINES,), but received {value.shape}."
                                f" NUM_TURBINES = {self.num_turbines})")

    # -- episode ------------------------------------------------------------------------------------
    def sample_wind(self, rng, options=None):
        """Draw order and clipping of mdp.py:237-258: weibull first, then normal; a draw happens only for
        a quantity that is neither given in `options`, nor pinned by the case, nor driven by a series."""
        fw = self.state_space["freewind_measurements"]
        series = bool(_truthy(self.farm_case.wind_time_series))
        ws = wd = None
        if options is not None and "wind_speed" in options:
            ws = options["wind_speed"]
        elif not (self.farm_case.set_wind_speed or series):
            ws = np.clip(WEIBULL_SCALE * rng.weibull(WEIBULL_SHAPE), fw.low[0], fw.high[0])
        if options is not None and "wind_direction" in options:
            wd = options["wind_direction"]
        elif not (self.farm_case.set_wind_direction or series):
            wd = np.clip(rng.normal(WD_MEAN, WD_STD) % 360, fw.low[1], fw.high[1])
        return ws, wd

    def reset(self, seed: int = None, options: dict = None):
        rng = np.random.default_rng(seed)
        ws, wd = self.sample_wind(rng, options)
        self.interface.init(ws, wd)
        for _ in range(self.start_iter + 1):  # at least one solve with yaw = 0 (mdp.py:261-262)
            self.interface.update_command()
        start = OrderedDict((a, self.interface.get_measure(a)) for a in self.state_attributes)
        self.start_state = clip_to_dict_space(start, self.state_space)  # only the START state is clipped (C9)
        self._actuation_accumulator = self._zero_accumulator()
        return self.start_state

    def step_interface(self, state: dict):
        command = OrderedDict((c, state[c]) for c in self.controls)
        done = self.interface.update_command(**command)
        powers = self.get_state_powers()
        for m in self.measures:
            state[m] = self.interface.get_measure(m)
        loads = self.interface.get_measure("load")
        if loads is not None:
            loads /= 1e7
        return state, powers / 1e6, loads, done

    def take_action(self, state: dict, joint_action: dict):
        return self.step_interface(self.get_controlled_state_transition(state, joint_action))

    def get_controlled_state_transition(self, state: dict, joint_action: dict):
        """Deterministic actuator update: clip the increment, accumulate, clip the setpoint (mdp.py:291-319)."""
        if not isinstance(joint_action, dict):
            raise TypeError("Joint action must be a dictionary")
        state = clip_to_dict_space(self._cast_dict_array(state), self.state_space)
        nxt = copy.deepcopy(state)
        for control, delta in joint_action.items():
            assert control in self.controls, f"Control of `{control}` is not activated"
            delta = np.array(delta, np.float32)
            if self.continuous_control:
                box = self.action_space[control]
                delta = np.clip(delta, box.low, box.high)
            else:
                delta = (delta - 1) * self.controls[control][-1]
            bounds = self.state_space[control]
            nxt[control] = np.clip(state[control] + delta, bounds.low, bounds.high)
            if control in self._actuation_accumulator:
                self._actuation_accumulator[control] += np.abs(delta)
        return nxt


def _truthy(x) -> bool:
    """`bool(wind_time_series)` of the reference breaks on ndarrays; accept both."""
    if x is None:
        return False
    if isinstance(x, np.ndarray):
        return x.size > 0
    return bool(x)
