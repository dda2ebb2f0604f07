"""Minimal stand-in for the parts of pettingzoo==1.24 the AEC surface touches."""
from __future__ import annotations


class agent_selector:
    """Cyclic iterator over an agent order; `is_last()` is true right after the last agent was returned."""

    def __init__(self, agent_order):
        self.reinit(agent_order)

    def reinit(self, agent_order):
        self.agent_order = list(agent_order)
        self._current_agent = 0
        self.selected_agent = 0

    def reset(self):
        self.reinit(self.agent_order)
        return self.next()

    def next(self):
        self._current_agent = (self._current_agent + 1) % len(self.agent_order)
        self.selected_agent = self.agent_order[self._current_agent - 1]
        return self.selected_agent

    def is_last(self):
        return self.selected_agent == self.agent_order[-1]

    def is_first(self):
        return self.selected_agent == self.agent_order[0]


class AECEnv:
    metadata: dict = {}
    possible_agents: list
    agents: list
    agent_selection = None

    def observe(self, agent):  # pragma: no cover
        raise NotImplementedError

    def step(self, action):  # pragma: no cover
        raise NotImplementedError

    def reset(self, seed=None, options=None):  # pragma: no cover
        raise NotImplementedError

    def close(self):
        pass

    @property
    def num_agents(self):
        return len(self.agents)

    @property
    def max_num_agents(self):
        return len(self.possible_agents)

    @property
    def unwrapped(self):
        return self

    def _clear_rewards(self):
        for a in self.rewards:
            self.rewards[a] = 0

    def _accumulate_rewards(self):
        for a, r in self.rewards.items():
            self._cumulative_rewards[a] += r

    def _deads_step_first(self):
        dead = [a for a in self.agents if self.terminations[a] or self.truncations[a]]
        if dead:
            self._skip_agent_selection = self.agent_selection
            self.agent_selection = dead[0]
        return self.agent_selection

    def _was_dead_step(self, action):
        if action is not None:
            raise ValueError("when an agent is dead, the only valid action is None")
        agent = self.agent_selection
        assert self.terminations[agent] or self.truncations[agent], "an agent that was not dead was stepped as dead"
        for d in (self.terminations, self.truncations, self.rewards, self._cumulative_rewards, self.infos):
            d.pop(agent, None)
        self.agents.remove(agent)
        dead = [a for a in self.agents if self.terminations[a] or self.truncations[a]]
        if dead:
            if getattr(self, "_skip_agent_selection", None) is None:
                self._skip_agent_selection = self.agent_selection
            self.agent_selection = dead[0]
        else:
            if getattr(self, "_skip_agent_selection", None) is not None:
                self.agent_selection = self._skip_agent_selection
            self._skip_agent_selection = None
        self._clear_rewards()

    def agent_iter(self, max_iter=2**63):
        n = 0
        while self.agents and n < max_iter:
            n += 1
            yield self.agent_selection

    def last(self, observe=True):
        agent = self.agent_selection
        obs = self.observe(agent) if observe else None
        return (obs, self._cumulative_rewards[agent], self.terminations[agent], self.truncations[agent],
                self.infos[agent])


class BaseWrapper(AECEnv):
    """Forwards attributes and methods to the wrapped AEC env."""

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_") or name == "env":
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def observe(self, agent):
        return self.env.observe(agent)

    def step(self, action):
        return self.env.step(action)

    def reset(self, seed=None, options=None):
        return self.env.reset(seed=seed, options=options)

    def close(self):
        return self.env.close()

    def last(self, observe=True):
        return self.env.last(observe)

    def agent_iter(self, max_iter=2**63):
        return self.env.agent_iter(max_iter)

    def observation_space(self, agent):
        return self.env.observation_space(agent)

    def action_space(self, agent):
        return self.env.action_space(agent)

    def state(self):
        return self.env.state()
