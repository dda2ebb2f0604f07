"""Env registry and factory: `make("<Layout>_Floris" | "Dec_<Layout>_Floris", ...)`
(reference wfcrl/environments/registration.py:17-117).  The `_Floris` suffix is served by the HIP
backend; `_Fastfarm` names are listed (same registry as the reference) but raise on make()."""
from __future__ import annotations

import math
import re
from itertools import product
from typing import Union

from ..interface import FastFarmInterface, HipFlorisInterface
from ..multiagent_env import MAWindFarmEnv
from ..simple_env import WindFarmEnv
from ..wrappers import AECLogWrapper, LogWrapper
from .data_cases import ALIASES, DefaultControl, FarmRowFastfarm, FarmRowFloris, named_cases_dictionary

env_pattern = r"(Dec_)*(\w+\d*_)(\w+)"
layout_pattern = r"Turb(\d+)_Row(\d+)"

registered_simulators = ["Fastfarm", "Floris"]
registered_layouts = list(named_cases_dictionary.keys()) + [f"Turb{n}_Row1_" for n in range(1, 13)]
control_types = ["", "Dec_"]
registered_envs = ["".join(p) for p in product(control_types, registered_layouts, registered_simulators)]
# build-defined aliases, kept apart from the reference's 88 names
alias_envs = ["".join(p) for p in product(control_types, ALIASES.keys(), ["Floris"])]


def get_default_control(controls):
    d = DefaultControl()
    return {name: getattr(d, name) for name in ("yaw", "pitch", "torque") if name in controls}


def get_case(name: str, simulator: str):
    k = registered_simulators.index(simulator)
    if name in named_cases_dictionary:
        return named_cases_dictionary[name][k]
    if name in ALIASES:
        return ALIASES[name][k]
    m = re.match(layout_pattern, name)
    n_turbines, n_rows = int(m.group(1)), int(m.group(2))
    assert n_rows == 1  # only single rows are generated procedurally
    return (FarmRowFastfarm if k == 0 else FarmRowFloris).build(n_turbines)


def validate_case(env_id, case):
    try:
        assert len(case.xcoords) == len(case.ycoords), \
            "xcoords and ycoords layout coordinates must have the same length"
    except Exception as e:
        raise ValueError(f"Invalid configuration for case {env_id}: {e}")


def make(env_id: str, controls: Union[dict, list] = ["yaw"], log=True, **env_kwargs):
    """Return a wind-farm benchmark environment.  Extra kwarg of this build: `env_batch=B` (B > 1)
    returns the batched, device-resident `VecWindFarmEnv` instead of the single-farm env."""
    if env_id not in registered_envs and env_id not in alias_envs:
        raise ValueError(f"{env_id} is not a registered WFCRL benchmark environment.")
    dec, name, simulator = re.match(env_pattern, env_id).groups()
    case = get_case(name, simulator).clone()
    validate_case(env_id, case)
    if "wind_time_series" in env_kwargs:
        case.wind_time_series = env_kwargs.pop("wind_time_series")
    if "path_to_simulator" in env_kwargs:
        case.path_to_simulator = env_kwargs.pop("path_to_simulator")
    if not isinstance(controls, dict):
        controls = get_default_control(controls)
    start_iter = math.ceil(case.t_init / case.dt)
    if simulator == "Fastfarm":
        FastFarmInterface.from_case(case)  # raises NotImplementedError: out of scope
    env_batch = env_kwargs.pop("env_batch", None)
    if env_batch is not None:
        from ..vec_env import VecWindFarmEnv

        return VecWindFarmEnv(case, controls, env_batch=env_batch, start_iter=start_iter, **env_kwargs)
    env_class = MAWindFarmEnv if dec == "Dec_" else WindFarmEnv
    env = env_class(interface=HipFlorisInterface, farm_case=case, controls=controls, start_iter=start_iter,
                    **env_kwargs)
    if log:
        env = (AECLogWrapper if dec == "Dec_" else LogWrapper)(env)
    return env


def list_envs():
    return registered_envs
