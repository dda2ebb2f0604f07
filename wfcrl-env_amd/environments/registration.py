"""Env registry and factory.

`make("<Layout>_Floris")` builds the centralised Gymnasium env, `make("Dec_<Layout>_Floris")` the per-turbine
PettingZoo AEC env; `<Layout>` is one of the named farms (layouts.json) or a procedural single row `Turb<N>_Row1_`
with N = 1..12.  Same name grammar, defaults and 88 registered names as reference
wfcrl/environments/registration.py:17-117; the `_Floris` suffix is served by the HIP backend, `_Fastfarm` names are
listed for compatibility but raise (that simulator is out of scope).  Extra keyword of this build: `env_batch=B`
returns the batched, device-resident `VecWindFarmEnv`.
"""
from __future__ import annotations

import math
import re
from typing import Union

from ..interface import FastFarmInterface, HipFlorisInterface
from ..multiagent_env import MAWindFarmEnv
from ..simple_env import WindFarmEnv
from ..wrappers import AECLogWrapper, LogWrapper
from .data_cases import ALIASES, DefaultControl, FarmRowFastfarm, FarmRowFloris, named_cases_dictionary

# "<Dec_>?<layout>_<simulator>"  and  "Turb<N>_Row<R>"
env_pattern = r"(Dec_)*(\w+\d*_)(\w+)"
layout_pattern = r"Turb(\d+)_Row(\d+)"

registered_simulators = ["Fastfarm", "Floris"]
control_types = ["", "Dec_"]
registered_layouts = [*named_cases_dictionary, *(f"Turb{n}_Row1_" for n in range(1, 13))]


def _enumerate(prefixes, layouts, simulators):
    return [p + layout + sim for p in prefixes for layout in layouts for sim in simulators]


registered_envs = _enumerate(control_types, registered_layouts, registered_simulators)
# build-defined aliases, kept apart from the reference's 88 names
alias_envs = _enumerate(control_types, list(ALIASES), ["Floris"])


def get_default_control(controls) -> dict:
    defaults = DefaultControl()
    return {name: getattr(defaults, name) for name in ("yaw", "pitch", "torque") if name in controls}


def get_case(name: str, simulator: str):
    column = registered_simulators.index(simulator)  # [FAST.Farm case, FLORIS case]
    for table in (named_cases_dictionary, ALIASES):
        if name in table:
            return table[name][column]
    n_turbines, n_rows = (int(g) for g in re.match(layout_pattern, name).groups())
    assert n_rows == 1  # only single rows are generated procedurally
    return (FarmRowFloris if simulator == "Floris" else FarmRowFastfarm).build(n_turbines)


def validate_case(env_id, case) -> None:
    if len(case.xcoords) != len(case.ycoords):
        raise ValueError(f"Invalid configuration for case {env_id}: "
                         "xcoords and ycoords layout coordinates must have the same length")


def _parse(env_id: str):
    if env_id not in registered_envs and env_id not in alias_envs:
        raise ValueError(f"{env_id} is not a registered WFCRL benchmark environment.")
    prefix, layout, simulator = re.match(env_pattern, env_id).groups()
    return prefix == "Dec_", layout, simulator


def make(env_id: str, controls: Union[dict, list] = ["yaw"], log=True, **env_kwargs):
    """Return a wind-farm benchmark environment (see the module docstring)."""
    decentralised, layout, simulator = _parse(env_id)
    case = get_case(layout, simulator).clone()
    validate_case(env_id, case)
    for key in ("wind_time_series", "path_to_simulator"):  # case-level options passed as env kwargs
        if key in env_kwargs:
            setattr(case, key, env_kwargs.pop(key))
    if simulator == "Fastfarm":
        FastFarmInterface.from_case(case)  # raises NotImplementedError: out of scope
    if not isinstance(controls, dict):
        controls = get_default_control(controls)
    first_control_iter = math.ceil(case.t_init / case.dt)

    batch = env_kwargs.pop("env_batch", None)
    if batch is not None:
        from ..vec_env import VecWindFarmEnv

        if decentralised:  # "Dec_<layout>_Floris" with env_batch: the batched AEC flavour (one launch per agent cycle)
            from ..vec_adapters import VecAECLogWrapper, VecAECWindFarmEnv

            # the per-agent budget of the AEC env replaces the joint one (multiagent_env.py:196-207); a logged env keeps
            # what it returned, so its inner env never reuses output buffers
            if env_kwargs.pop("actuation_budget", None) is not None:
                raise ValueError("actuation_budget is fixed by the AEC flavour (per-agent budget, reference "
                                 "wfcrl/multiagent_env.py:196-207)")
            if log:
                env_kwargs["reuse_buffers"] = False
            inner = VecWindFarmEnv(case, controls, env_batch=batch, start_iter=first_control_iter,
                                   actuation_budget=float("inf"), **env_kwargs)
            env = VecAECWindFarmEnv(inner)
            return VecAECLogWrapper(env) if log else env
        return VecWindFarmEnv(case, controls, env_batch=batch, start_iter=first_control_iter, **env_kwargs)

    flavour, logger = (MAWindFarmEnv, AECLogWrapper) if decentralised else (WindFarmEnv, LogWrapper)
    env = flavour(interface=HipFlorisInterface, farm_case=case, controls=controls, start_iter=first_control_iter,
                  **env_kwargs)
    return logger(env) if log else env


def list_envs():
    return registered_envs
