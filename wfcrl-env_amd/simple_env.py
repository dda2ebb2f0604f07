"""`WindFarmEnv` — the centralised Gymnasium flavour: one dict action {"yaw": (N,)} drives the whole farm
(API of reference wfcrl/simple_env.py:13-99; episode logic in env_core.FarmEpisode)."""
from __future__ import annotations

import copy

from ._compat import gym
from .env_core import ACTUATION_BUDGET, FarmEpisode  # noqa: F401  (ACTUATION_BUDGET re-exported)


class WindFarmEnv(gym.Env):
    metadata = {"name": "centralized-windfarm"}

    def __init__(self, interface, farm_case, controls: dict, continuous_control: bool = True, reward_shaper=None,
                 start_iter: int = 0, max_num_steps: int = 500, load_coef: float = 0.1):
        self._episode = FarmEpisode(interface, farm_case, controls, continuous_control, reward_shaper, start_iter,
                                    max_num_steps, load_coef)
        mdp = self.mdp = self._episode.mdp
        self.action_space, self.observation_space = mdp.action_space, mdp.state_space
        self.num_turbines = mdp.num_turbines
        self.continuous_control, self.controls = continuous_control, controls
        self.max_num_steps, self.load_coef = max_num_steps, load_coef
        self.farm_case, self.dt = farm_case, farm_case.dt
        self.accumulated_actions = mdp.get_accumulated_actions()
        self.num_moves = 0

    # attributes the reference exposes directly
    @property
    def reward_shaper(self):
        return self._episode.reward_shaper

    @property
    def _state(self):
        return self._episode.state

    def reset(self, seed=None, options=None):
        """Observation only — not (obs, info) — as the reference."""
        state = self._episode.reset(seed, options)
        self.accumulated_actions = self.mdp.get_accumulated_actions()
        self.num_moves = 0
        return copy.deepcopy(state)

    def step(self, actions: dict):
        assert self._state is not None, "Call reset before `step`"
        self.num_moves += 1
        for control, increments in actions.items():
            blocked = self._episode.over_budget(control, self.accumulated_actions[control], self.num_moves)
            if blocked is not None:
                increments[blocked] = 0.0  # in place, on the caller's array (Appendix C5)
        outcome = self._episode.advance(actions)
        self.accumulated_actions = self.mdp.get_accumulated_actions()
        return copy.deepcopy(outcome.state), outcome.reward, False, outcome.truncated, outcome.info()

    def close(self):
        return None
