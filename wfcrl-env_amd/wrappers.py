"""History loggers around the two env flavours (reference wfcrl/wrappers.py:24-88).
`RandomSimulator` of the reference is a dead stub (its interface hooks are `pass`) and is not built."""
from __future__ import annotations

from ._compat import BaseWrapper, gym

_FIELDS = ("observation", "reward", "load", "power")


def _mirror_env_attributes(wrapper, env):
    wrapper.continuous_control = env.continuous_control
    wrapper.max_num_steps = env.max_num_steps
    wrapper._state = env.mdp.start_state
    wrapper.num_turbines = env.mdp.num_turbines
    wrapper.mdp = env.mdp
    wrapper.controls = env.controls


def _record(history: dict, observation, reward, info):
    history["observation"].append(observation)
    history["reward"].append(reward)
    for key in ("power", "load"):
        if key in info:
            history[key].append(info[key])


class LogWrapper(gym.Wrapper):
    """Appends every step's observation / reward / power / load to `history` (python lists)."""

    def __init__(self, env):
        super().__init__(env)
        self.history = {k: [] for k in _FIELDS}
        _mirror_env_attributes(self, self.env)

    def step(self, action):
        result = self.env.step(action)
        _record(self.history, result[0], result[1], result[4])
        return result

    def reset(self, seed=None, options=None):
        self.history = {k: [] for k in _FIELDS}
        return self.env.reset(seed, options)


class AECLogWrapper(BaseWrapper):
    """Per-agent history, recorded on `last()`."""

    def __init__(self, env):
        super().__init__(env)
        self.history = {a: {k: [] for k in _FIELDS} for a in self.env.possible_agents}
        _mirror_env_attributes(self, self.env)

    def last(self):
        agent = self.env.agent_selection
        result = self.env.last()
        _record(self.history[agent], result[0], result[1], result[4])
        return result

    def reset(self, seed=None, options=None):
        self.history = {a: {k: [] for k in _FIELDS} for a in self.env.possible_agents}
        return self.env.reset(seed, options)
