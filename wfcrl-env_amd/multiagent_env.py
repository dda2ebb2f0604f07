"""`MAWindFarmEnv` — the PettingZoo AEC flavour: one agent per turbine ("turbine_1" .. "turbine_N"); the farm
advances once every agent of the cycle has submitted its action (API of reference wfcrl/multiagent_env.py:15-257;
episode logic in env_core.FarmEpisode)."""
from __future__ import annotations

import functools
from collections import OrderedDict

import numpy as np

from ._compat import AECEnv, agent_selector, spaces
from .env_core import FarmEpisode

_GLOBAL_ONLY = "freewind_measurements"  # never part of a turbine's local observation


def _scalar_boxes(space_dict, index: int, skip=()):
    return {key: spaces.Box(box.low[index], box.high[index]) for key, box in space_dict.items() if key not in skip}


class MAWindFarmEnv(AECEnv):
    metadata = {"name": "multiagent-windfarm", "is_parallelizable": True}

    def __init__(self, interface, farm_case, controls: dict, continuous_control: bool = True, reward_shaper=None,
                 start_iter: int = 0, max_num_steps: int = 500, load_coef: float = 0.1):
        self._episode = FarmEpisode(interface, farm_case, controls, continuous_control, reward_shaper, start_iter,
                                    max_num_steps, load_coef)
        mdp = self.mdp = self._episode.mdp
        self.num_turbines = mdp.num_turbines
        self.continuous_control, self.controls = continuous_control, controls
        self.max_num_steps, self.load_coef, self.farm_case = max_num_steps, load_coef, farm_case
        self.state_space = mdp.state_space
        self.possible_agents = [f"turbine_{k}" for k in range(1, self.num_turbines + 1)]
        self.agent_name_mapping = {name: k for k, name in enumerate(self.possible_agents)}
        self._build_agent_spaces()

    @property
    def reward_shaper(self):
        return self._episode.reward_shaper

    @property
    def _state(self):
        return self._episode.state

    # -- spaces --------------------------------------------------------------------------------------
    def _build_agent_spaces(self):
        self._obs_spaces, self._action_spaces = {}, {}
        for name, k in self.agent_name_mapping.items():
            self._obs_spaces[name] = _scalar_boxes(self.mdp.state_space, k, skip=(_GLOBAL_ONLY,))
            if self.continuous_control:
                self._action_spaces[name] = _scalar_boxes(self.mdp.action_space, k)
            else:
                self._action_spaces[name] = {key: sp[k] for key, sp in self.mdp.action_space.items()}

    @functools.lru_cache(maxsize=None)
    def observation_space(self, agent):
        return self._obs_spaces[agent]

    @functools.lru_cache(maxsize=None)
    def action_space(self, agent):
        return self._action_spaces[agent]

    def state(self):
        return self._state

    def observe(self, agent):
        k = self.agent_name_mapping[agent]
        return OrderedDict((key, values[k]) for key, values in self.state().items() if key != _GLOBAL_ONLY)

    # -- episode -------------------------------------------------------------------------------------
    def reset(self, seed=None, options=None):
        self._episode.reset(seed, options)
        names = self.agents = list(self.possible_agents)
        zero = lambda: np.array([0.0])  # noqa: E731
        self.rewards = {a: zero() for a in names}
        self._cumulative_rewards = {a: zero() for a in names}
        self.terminations = dict.fromkeys(names, False)
        self.truncations = dict.fromkeys(names, False)
        self.infos = {a: {} for a in names}
        self.actions = dict.fromkeys(names)
        self.observations = {a: self.observe(a) for a in names}
        self.constrained = {a: self.observe(a) for a in names}
        self._num_steps = dict.fromkeys(names, 0)
        totals = self.mdp.get_accumulated_actions()
        self.accumulated_actions = {a: {c: totals[c][k] for c in totals} for k, a in enumerate(names)}
        self.num_moves = 0
        self._agent_selector = agent_selector(names)
        self.agent_selection = self._agent_selector.next()

    def _validate(self, agent, action):
        active = self.mdp.controls
        for control in action:
            if control not in active:
                raise ValueError(f"Control `{control}` for agent {agent} is not activated."
                                 f" List of activated controls: {list(active.keys())}")
        if any(control not in action for control in active):
            raise ValueError(f"Action {action} for agent {agent} is incomplete."
                             f" List of needed controls: {active.keys()}")

    def _join_actions(self, per_agent):
        joint = {c: np.zeros(self.num_turbines, dtype=np.float32) for c in self.mdp.controls}
        for k, action in enumerate(per_agent.values()):
            for control, value in action.items():
                joint[control][k] = np.asarray(value).reshape(-1)[0]
        return joint

    def _publish(self, outcome):
        for name, k in self.agent_name_mapping.items():
            if name not in self.agents:
                continue
            self.infos[name].update(outcome.info(k))
            self.rewards[name] = outcome.reward  # cooperative: one reward for every turbine
            self.observations[name] = self.observe(name)
            self.truncations[name], self.terminations[name] = outcome.truncated, False

    def step(self, action):
        assert self._state is not None, "Call reset before `step`"
        agent = self.agent_selection
        if self.truncations[agent] or self.terminations[agent]:
            self._was_dead_step(action)
            return
        self._num_steps[agent] += 1
        self._validate(agent, action)
        for control, value in action.items():  # per-agent actuation budget, evaluated on this agent's own move count
            blocked = self._episode.over_budget(control, self.accumulated_actions[agent][control], self._num_steps[agent])
            if blocked is not None and blocked:
                value[:] = 0.0
        self._cumulative_rewards[agent] = 0
        self.actions[agent] = action

        if self._agent_selector.is_last():
            self._publish(self._episode.advance(self._join_actions(self.actions)))
            self.num_moves += 1
        else:
            self._clear_rewards()  # nobody is paid until the cycle is complete

        totals = self.mdp.get_accumulated_actions()
        k = self.agent_name_mapping[agent]
        for control in action:
            self.accumulated_actions[agent][control] = totals[control][k]
        self.agent_selection = self._agent_selector.next()
        self._accumulate_rewards()

    def close(self):
        return None
