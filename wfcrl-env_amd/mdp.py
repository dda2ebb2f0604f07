"""`WindFarmMDP` — the farm as a Markov decision process: action / state spaces, the deterministic actuator
transition, reset-time wind sampling, and the hand-off to the simulator interface.

Host-side mirror of reference wfcrl/mdp.py:19-319 (same class, attributes, units, error behaviour) so that the
Gymnasium / PettingZoo surface behaves identically on top of `HipFlorisInterface`.  The batched, device-resident
restatement of the same transition is the fused env step of the HIP kernel (`vec_env.VecWindFarmEnv`).
"""
from __future__ import annotations

import copy
from collections import OrderedDict
from collections.abc import Iterable
from warnings import warn

import numpy as np

from ._compat import spaces
from .interface import BaseInterface

# reset-time wind distribution (mdp.py:237-258): ws = clip(8 * Weibull(8)), wd = clip(N(270, 20) mod 360)
WEIBULL_SHAPE, WEIBULL_SCALE = 8, 8
WD_MEAN, WD_STD = 270, 20

FREEWIND = "freewind_measurements"


def clip_to_dict_space(element: dict, space) -> dict:
    """Clip every entry of `element` (in place) to the bounds of the matching sub-space."""
    for key, value in element.items():
        box = space[key]
        element[key] = np.clip(value, box.low, box.high)
    return element


def _truthy(x) -> bool:
    """`bool(wind_time_series)` of the reference breaks on ndarrays; accept both."""
    if x is None:
        return False
    return x.size > 0 if isinstance(x, np.ndarray) else bool(x)


def validate_controls(control_dict: dict, allowed, interface, continuous: bool) -> None:
    """Checks of reference mdp.py:174-211; fills in a default step of 1 (with a warning) when none is given."""
    for name, spec in control_dict.items():
        if name not in allowed:
            raise ValueError(f"Cannot control {name}. Allowed controls are {allowed}")
        if name not in interface.CONTROL_SET:
            raise ValueError(f"Cannot control `{name}`. Interface {interface.__class__.__name__}"
                             f" only allows for the following: {interface.CONTROL_SET}")
        well_formed = isinstance(spec, Iterable) and 2 <= len(spec) <= 3
        if not well_formed:
            raise TypeError(f"Wrong bounds for actuator {name}: Bounds on actuators must be an iterable of the type"
                            " [lower_bound, upper_bound] if control is continuous and"
                            " [lower_bound, upper_bound, step_size] otherwise")
        lower, upper = spec[0], spec[1]
        if not lower < upper:
            raise ValueError(f"Wrong bounds for actuator {name}: ensure that lower_bound < upper_bound")
        if len(spec) == 2:
            warn(f"No step size was provided for actuator {name}. Step size will default to 1.")
            control_dict[name] = tuple(spec) + (1,)
        elif not continuous and spec[2] <= 0:
            raise ValueError(f"Invalid step size provided for actuator {name} the step size must be stricly positive")


def build_action_space(controls: dict, n: int, continuous: bool):
    """Continuous: one increment in [-step, +step] per turbine; discrete: {0, 1, 2} = down / hold / up."""
    if continuous:
        return spaces.Dict({name: spaces.Box(-spec[2], spec[2], shape=(n,)) for name, spec in controls.items()})
    return spaces.Dict({name: spaces.MultiDiscrete(np.full(n, 3).tolist()) for name in controls})


def build_state_space(attributes, controls: dict, defaults: dict, n: int):
    """Ordered Dict space: controlled quantities first (their own bounds), then the measured ones."""
    per_turbine = np.ones(n, dtype=np.float32)
    out = OrderedDict()
    for attr in attributes:
        if attr == FREEWIND:
            lo = np.array([defaults["wind_speed"][0], defaults["wind_direction"][0]], dtype=np.float32)
            hi = np.array([defaults["wind_speed"][1], defaults["wind_direction"][1]], dtype=np.float32)
        else:
            bounds = controls[attr] if attr in controls else defaults[attr]
            lo, hi = per_turbine * bounds[0], per_turbine * bounds[1]
        out[attr] = spaces.Box(lo, hi, shape=lo.shape)
    return spaces.Dict(out)


class WindFarmMDP:
    CONTROL_SET = ["yaw", "pitch", "torque"]
    POSSIBLE_STATE_ATTRIBUTES = [FREEWIND, "wind_speed", "wind_direction", "yaw", "pitch", "torque"]
    DEFAULT_BOUNDS = {
        "wind_speed": [3, 28],
        "wind_direction": [0, 360],
        "yaw": [-40, 40],
        "pitch": [0, 360],
        "torque": [-1e5, 1e5],
    }
    ACTUATORS_RATE = {"yaw": 0.3, "pitch": 8}  # deg / s

    def __init__(self, interface, farm_case, controls: dict, continuous_control: bool = True, start_iter: int = 0,
                 horizon: int = int(1e6)):
        # the reference mutates the shared module-level case here; we work on a copy
        self.farm_case = farm_case.clone() if hasattr(farm_case, "clone") else copy.copy(farm_case)
        self.farm_case.max_iter = horizon
        self.interface = self._resolve_interface(interface)
        self.num_turbines = self.farm_case.num_turbines
        self.continuous_control, self.horizon, self.start_iter = continuous_control, horizon, start_iter

        self._check_controls(controls)
        self.controls, self.num_controls = controls, len(controls)
        self.measures = [attr for attr in self.POSSIBLE_STATE_ATTRIBUTES
                         if attr not in controls and attr in self.interface.measure_map]
        self.state_attributes = [*controls, *self.measures]
        self.action_space = build_action_space(controls, self.num_turbines, continuous_control)
        self.state_space = build_state_space(self.state_attributes, controls, self.DEFAULT_BOUNDS, self.num_turbines)
        self.start_state = None
        self._actuation_accumulator = self._fresh_accumulator()

    def _resolve_interface(self, interface):
        if isinstance(interface, BaseInterface):
            warn("Interface already instantiated. Simulation arguments from `Farm case` will be ignored.")
            return interface
        simulator_path = self.farm_case.interface_kwargs.get("path_to_simulator", None)
        if simulator_path is None:
            return interface.from_case(self.farm_case)
        return interface.from_case(self.farm_case, simulator_path)

    def _fresh_accumulator(self):
        return {name: np.zeros(self.num_turbines, dtype=np.float32) for name in self.controls}

    # -- small accessors -----------------------------------------------------------------------------
    def get_state_powers(self):
        return self.interface.avg_powers()

    def get_accumulated_actions(self, agent=None):
        return dict(self._actuation_accumulator)

    def _cast_dict_array(self, state):
        return OrderedDict((key, value.astype(np.float32)) for key, value in state.items())

    def _check_controls(self, control_dict: dict):
        validate_controls(control_dict, self.CONTROL_SET, self.interface, self.continuous_control)

    def _check_state(self, state: dict):
        expected_shape = (self.num_turbines,)
        for attr, value in state.items():
            if attr not in self.state_attributes:
                raise ValueError(f"Unknwon attribute {attr} in state dict. Accepted attributed are: {self.state_attributes}")
            if not isinstance(value, np.ndarray):
                raise TypeError(f"State attribute {attr} must be a numpy array. Received {type(value)}")
            if attr != FREEWIND and value.shape != expected_shape:
                raise TypeError(f"State attribute {attr} must be of shape (NUM_TURBINES,), but received {value.shape}."
                                f" NUM_TURBINES = {self.num_turbines})")

    # -- episode -------------------------------------------------------------------------------------
    def sample_wind(self, rng, options=None):
        """(ws, wd) for a reset.  A quantity is drawn only if it is neither given in `options`, nor pinned by the
        case, nor driven by a time series; the Weibull draw comes before the normal one (seed reproducibility)."""
        options = options or {}
        bounds = self.state_space[FREEWIND]
        driven = _truthy(self.farm_case.wind_time_series)

        def pick(key, pinned, draw, index):
            if key in options:
                return options[key]
            if pinned or driven:
                return None
            return np.clip(draw(), bounds.low[index], bounds.high[index])

        ws = pick("wind_speed", self.farm_case.set_wind_speed, lambda: WEIBULL_SCALE * rng.weibull(WEIBULL_SHAPE), 0)
        wd = pick("wind_direction", self.farm_case.set_wind_direction, lambda: rng.normal(WD_MEAN, WD_STD) % 360, 1)
        return ws, wd

    def reset(self, seed: int = None, options: dict = None):
        ws, wd = self.sample_wind(np.random.default_rng(seed), options)
        self.interface.init(ws, wd)
        for _ in range(self.start_iter + 1):  # at least one solve at yaw = 0 (mdp.py:261-262)
            self.interface.update_command()
        measured = OrderedDict((attr, self.interface.get_measure(attr)) for attr in self.state_attributes)
        self.start_state = clip_to_dict_space(measured, self.state_space)  # only the START state is clipped (C9)
        self._actuation_accumulator = self._fresh_accumulator()
        return self.start_state

    def get_controlled_state_transition(self, state: dict, joint_action: dict):
        """Clip the increment to the step, move the setpoint within its bounds, accumulate |increment|."""
        if not isinstance(joint_action, dict):
            raise TypeError("Joint action must be a dictionary")
        current = clip_to_dict_space(self._cast_dict_array(state), self.state_space)
        moved = copy.deepcopy(current)
        for control, raw in joint_action.items():
            assert control in self.controls, f"Control of `{control}` is not activated"
            increment = np.array(raw, np.float32)
            if self.continuous_control:
                limits = self.action_space[control]
                increment = np.clip(increment, limits.low, limits.high)
            else:
                increment = (increment - 1) * self.controls[control][-1]
            setpoint_bounds = self.state_space[control]
            moved[control] = np.clip(current[control] + increment, setpoint_bounds.low, setpoint_bounds.high)
            if control in self._actuation_accumulator:
                self._actuation_accumulator[control] += np.abs(increment)
        return moved

    def step_interface(self, state: dict):
        commands = OrderedDict((name, state[name]) for name in self.controls)
        done = self.interface.update_command(**commands)
        powers_w = self.get_state_powers()
        for attr in self.measures:
            state[attr] = self.interface.get_measure(attr)
        loads = self.interface.get_measure("load")
        if loads is not None:
            loads /= 1e7  # the interface stores load proxies x 1e7 (interface.py:575-577)
        return state, powers_w / 1e6, loads, done

    def take_action(self, state: dict, joint_action: dict):
        return self.step_interface(self.get_controlled_state_transition(state, joint_action))
