"""`MAWindFarmEnv` — PettingZoo AEC env, one agent per turbine, joint step when the last agent has
acted (reference wfcrl/multiagent_env.py:15-257)."""
from __future__ import annotations

import functools
from collections import OrderedDict

import numpy as np

from ._compat import AECEnv, agent_selector, spaces
from .mdp import WindFarmMDP
from .rewards import DoNothingReward, RewardShaper, power_reward
from .simple_env import ACTUATION_BUDGET


class MAWindFarmEnv(AECEnv):
    metadata = {"name": "multiagent-windfarm", "is_parallelizable": True}

    def __init__(self, interface, farm_case, controls: dict, continuous_control: bool = True,
                 reward_shaper: RewardShaper = None, start_iter: int = 0, max_num_steps: int = 500,
                 load_coef: float = 0.1):
        self.mdp = WindFarmMDP(interface=interface, farm_case=farm_case, controls=controls,
                               continuous_control=continuous_control, start_iter=start_iter,
                               horizon=start_iter + max_num_steps)
        self.continuous_control = continuous_control
        self.max_num_steps = max_num_steps
        self._state = None
        self.num_turbines = self.mdp.num_turbines
        self.reward_shaper = DoNothingReward() if reward_shaper is None else reward_shaper
        self.controls = controls
        self.farm_case = farm_case
        self.state_space = self.mdp.state_space
        self.load_coef = load_coef
        self.possible_agents = [f"turbine_{i + 1}" for i in range(self.num_turbines)]
        self.agent_name_mapping = {a: i for i, a in enumerate(self.possible_agents)}
        self._build_agent_spaces()

    # -- spaces -------------------------------------------------------------------------------------
    def _build_agent_spaces(self):
        """Per-agent plain dicts of (1,)-shaped boxes; local observations exclude the free wind."""
        self._obs_spaces, self._action_spaces = {}, {}
        for i, agent in enumerate(self.possible_agents):
            self._obs_spaces[agent] = {k: spaces.Box(s.low[i], s.high[i]) for k, s in self.mdp.state_space.items()
                                       if k != "freewind_measurements"}
            if self.continuous_control:
                self._action_spaces[agent] = {k: spaces.Box(s.low[i], s.high[i])
                                              for k, s in self.mdp.action_space.items()}
            else:
                self._action_spaces[agent] = {k: s[i] for k, s in self.mdp.action_space.items()}

    @functools.lru_cache(maxsize=None)
    def observation_space(self, agent):
        return self._obs_spaces[agent]

    @functools.lru_cache(maxsize=None)
    def action_space(self, agent):
        return self._action_spaces[agent]

    def state(self):
        return self._state

    def observe(self, agent):
        i = self.agent_name_mapping[agent]
        return OrderedDict((k, v[i]) for k, v in self.state().items() if k != "freewind_measurements")

    def _join_actions(self, agent_actions):
        joint = {c: np.zeros(self.num_turbines, dtype=np.float32) for c in self.mdp.controls}
        for j, action in enumerate(agent_actions.values()):
            for control in action:
                joint[control][j] = np.asarray(action[control]).reshape(-1)[0]
        return joint

    # -- episode ------------------------------------------------------------------------------------
    def reset(self, seed=None, options=None):
        self.mdp.reset(seed, options)
        self._state = self.mdp.start_state
        self.reward_shaper.reset()
        self.agents = self.possible_agents[:]
        self._num_steps = {a: 0 for a in self.agents}
        self.rewards = {a: np.array([0.0]) for a in self.agents}
        self._cumulative_rewards = {a: np.array([0.0]) for a in self.agents}
        self.terminations = {a: False for a in self.agents}
        self.truncations = {a: False for a in self.agents}
        self.infos = {a: {} for a in self.agents}
        self.actions = {a: None for a in self.agents}
        self.observations = {a: self.observe(a) for a in self.agents}
        self.constrained = {a: self.observe(a) for a in self.agents}
        acc = self.mdp.get_accumulated_actions()
        self.accumulated_actions = {a: {c: acc[c][i] for c in acc} for i, a in enumerate(self.agents)}
        self.num_moves = 0
        self._agent_selector = agent_selector(self.agents)
        self.agent_selection = self._agent_selector.next()

    def step(self, action):
        assert self._state is not None, "Call reset before `step`"
        agent = self.agent_selection
        if self.truncations[agent] or self.terminations[agent]:
            self._was_dead_step(action)
            return
        self._num_steps[agent] += 1

        for control in action:
            if control not in self.mdp.controls:
                raise ValueError(f"Control `{control}` for agent {agent} is not activated."
                                 f" List of activated controls: {list(self.mdp.controls.keys())}")
        if any(c not in action for c in self.mdp.controls):
            raise ValueError(f"Action {action} for agent {agent} is incomplete."
                             f" List of needed controls: {self.mdp.controls.keys()}")

        # actuation budget, per agent (multiagent_env.py:196-207)
        for control in action:
            rate = self.mdp.ACTUATORS_RATE.get(control)
            if rate is None:
                continue
            busy = self.accumulated_actions[agent][control] / rate / self._num_steps[agent] / self.farm_case.dt
            if busy >= ACTUATION_BUDGET:
                action[control][:] = 0.0

        self._cumulative_rewards[agent] = 0
        self.actions[agent] = action

        if self._agent_selector.is_last():
            ws_prev = self.state()["freewind_measurements"][0]
            next_state, powers, loads, truncated = self.mdp.take_action(self._state, self._join_actions(self.actions))
            reward = np.array([self.reward_shaper(power_reward(powers, ws_prev, loads, self.load_coef))])
            self._state = next_state
            for a in self.agents:  # cooperative: the same reward for every turbine
                i = self.agent_name_mapping[a]
                if loads is not None:
                    self.infos[a]["load"] = loads[i]
                self.rewards[a] = reward
                self.observations[a] = self.observe(a)
                self.truncations[a] = truncated
                self.terminations[a] = False
                self.infos[a]["power"] = powers[i]
            self.num_moves += 1
        else:
            self._clear_rewards()

        acc = self.mdp.get_accumulated_actions()
        for control in action:
            self.accumulated_actions[agent][control] = acc[control][self.agent_name_mapping[agent]]
        self.agent_selection = self._agent_selector.next()
        self._accumulate_rewards()

    def close(self):
        pass
