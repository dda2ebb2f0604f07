"""`WindFarmEnv` — centralised Gymnasium env: one dict action for the whole farm
(reference wfcrl/simple_env.py:13-99)."""
from __future__ import annotations

import copy

import numpy as np

from ._compat import gym
from .mdp import WindFarmMDP
from .rewards import DoNothingReward, RewardShaper, power_reward

ACTUATION_BUDGET = 0.1  # an actuator may move at most 10 % of the time (simple_env.py:64-72)


class WindFarmEnv(gym.Env):
    metadata = {"name": "centralized-windfarm"}

    def __init__(self, interface, farm_case, controls: dict, continuous_control: bool = True,
                 reward_shaper: RewardShaper = None, start_iter: int = 0, max_num_steps: int = 500,
                 load_coef: float = 0.1):
        self.mdp = WindFarmMDP(interface=interface, farm_case=farm_case, controls=controls,
                               continuous_control=continuous_control, start_iter=start_iter,
                               horizon=start_iter + max_num_steps)
        self.continuous_control = continuous_control
        self.action_space = self.mdp.action_space
        self.observation_space = self.mdp.state_space
        self._state = self.mdp.start_state
        self.num_turbines = self.mdp.num_turbines
        self.max_num_steps = max_num_steps
        self.reward_shaper = DoNothingReward() if reward_shaper is None else reward_shaper
        self.controls = controls
        self.dt = farm_case.dt
        self.farm_case = farm_case
        self.accumulated_actions = self.mdp.get_accumulated_actions()
        self.num_moves = 0
        self.load_coef = load_coef

    def reset(self, seed=None, options=None):
        """Returns the observation ONLY (not (obs, info)) — simple_env.py:49-56, SURVEY C4."""
        self.mdp.reset(seed, options)
        self._state = self.mdp.start_state
        self.reward_shaper.reset()
        self.accumulated_actions = self.mdp.get_accumulated_actions()
        self.num_moves = 0
        return copy.deepcopy(self._state)

    def _apply_actuation_constraint(self, actions: dict):
        """Zero (IN PLACE, as the reference does — C5) the increments of turbines whose accumulated
        actuation time exceeds the budget; evaluated before the current action is accumulated."""
        for control in actions:
            rate = self.mdp.ACTUATORS_RATE.get(control)
            if rate is None:
                continue
            busy = self.accumulated_actions[control] / rate / self.num_moves / self.farm_case.dt
            actions[control][busy >= ACTUATION_BUDGET] = 0.0

    def step(self, actions: dict):
        assert self._state is not None, "Call reset before `step`"
        self.num_moves += 1
        self._apply_actuation_constraint(actions)
        ws_prev = self._state["freewind_measurements"][0]  # freestream speed of the state BEFORE the step
        next_state, powers, loads, truncated = self.mdp.take_action(self._state, actions)
        reward = np.array([self.reward_shaper(power_reward(powers, ws_prev, loads, self.load_coef))])
        self._state = next_state
        info = {"power": powers}
        if loads is not None:
            info["load"] = loads
        self.accumulated_actions = self.mdp.get_accumulated_actions()
        return copy.deepcopy(self._state), reward, False, truncated, info

    def close(self):
        pass
