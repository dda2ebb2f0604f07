"""Episode logic shared by the centralised (Gymnasium) and the per-turbine (PettingZoo AEC) env flavours.

Both flavours of the reference run the same three things around `WindFarmMDP.take_action`:
  * the actuation budget — an actuator may be moving at most 10 % of the elapsed time
    (reference wfcrl/simple_env.py:64-72, multiagent_env.py:196-207),
  * the cooperative reward — production normalised by the free wind of the state BEFORE the step, minus a load
    penalty, then shaped (simple_env.py:78-85, multiagent_env.py:220-227),
  * the bookkeeping of the truncation flag and of the info payload.
`FarmEpisode` owns them once; `simple_env.WindFarmEnv` and `multiagent_env.MAWindFarmEnv` are API adapters.
"""
from __future__ import annotations

import numpy as np

from .mdp import WindFarmMDP
from .rewards import DoNothingReward, RewardShaper, power_reward

ACTUATION_BUDGET = 0.1  # fraction of the elapsed time an actuator may spend moving


class JointStep:
    """Outcome of one farm step."""

    __slots__ = ("state", "reward", "truncated", "powers", "loads")

    def __init__(self, state, reward, truncated, powers, loads):
        self.state, self.reward, self.truncated, self.powers, self.loads = state, reward, truncated, powers, loads

    def info(self, index=None) -> dict:
        """Per-farm (index None) or per-turbine info dict: power [MW], load proxies."""
        pick = (lambda a: a) if index is None else (lambda a: a[index])
        out = {"power": pick(self.powers)}
        if self.loads is not None:
            out["load"] = pick(self.loads)
        return out


class FarmEpisode:
    def __init__(self, interface, farm_case, controls: dict, continuous_control: bool, reward_shaper, start_iter: int,
                 max_num_steps: int, load_coef: float):
        self.mdp = WindFarmMDP(interface=interface, farm_case=farm_case, controls=controls,
                               continuous_control=continuous_control, start_iter=start_iter,
                               horizon=start_iter + max_num_steps)
        self.reward_shaper: RewardShaper = DoNothingReward() if reward_shaper is None else reward_shaper
        self.load_coef = load_coef
        self.dt = farm_case.dt
        self.state = self.mdp.start_state

    # -- actuation budget ----------------------------------------------------------------------------
    def over_budget(self, control: str, accumulated, num_moves: int):
        """True (per turbine) where accumulated |increments| / actuator rate already fills >= 10 % of the elapsed
        time.  None for controls without a rate limit (torque)."""
        rate = self.mdp.ACTUATORS_RATE.get(control)
        if rate is None:
            return None
        return accumulated / rate / num_moves / self.dt >= ACTUATION_BUDGET

    # -- episode -------------------------------------------------------------------------------------
    def reset(self, seed=None, options=None):
        self.mdp.reset(seed, options)
        self.state = self.mdp.start_state
        self.reward_shaper.reset()
        return self.state

    def advance(self, joint_action: dict) -> JointStep:
        ws_before = self.state["freewind_measurements"][0]
        next_state, powers, loads, truncated = self.mdp.take_action(self.state, joint_action)
        raw = power_reward(powers, ws_before, loads, self.load_coef)
        self.state = next_state
        return JointStep(next_state, np.array([self.reward_shaper(raw)]), truncated, powers, loads)
