"""Batched multi-agent adaptor and batched history logger.

`VecParallelWindFarmEnv` exposes B farms x N turbine-agents through a PettingZoo-parallel-style API on top of
`VecWindFarmEnv`: one `step(actions)` call = one joint step of every farm (the AEC env of the reference needs N
Python calls per joint step, wfcrl/multiagent_env.py:159-254).  Agent naming, per-agent observation keys and
the cooperative reward follow the reference (multiagent_env.py:51-53, 102-115, 229-238).
`VecLogWrapper` keeps observation / reward / power / load histories in preallocated ring buffers instead of
Python lists (wrappers.py:61-88).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from ._compat import spaces


class VecParallelWindFarmEnv:
    metadata = {"name": "vectorized-multiagent-windfarm", "is_parallelizable": True}

    def __init__(self, vec_env):
        self.env = vec_env
        self.num_envs = vec_env.num_envs
        self.num_turbines = vec_env.num_turbines
        self.possible_agents = [f"turbine_{i + 1}" for i in range(self.num_turbines)]
        self.agent_name_mapping = {a: i for i, a in enumerate(self.possible_agents)}
        self.agents = []
        sp, ac = vec_env.single_observation_space, vec_env.single_action_space
        self._obs_spaces = {a: {k: spaces.Box(s.low[i], s.high[i]) for k, s in sp.items() if k != "freewind_measurements"}
                            for i, a in enumerate(self.possible_agents)}
        if vec_env.continuous_control:
            self._act_spaces = {a: {k: spaces.Box(s.low[i], s.high[i]) for k, s in ac.items()}
                                for i, a in enumerate(self.possible_agents)}
        else:
            self._act_spaces = {a: {k: s[i] for k, s in ac.items()} for i, a in enumerate(self.possible_agents)}

    def observation_space(self, agent):
        return self._obs_spaces[agent]

    def action_space(self, agent):
        return self._act_spaces[agent]

    def state(self):
        return self._state

    def _split(self, obs):
        self._state = obs
        return {a: OrderedDict((k, v[:, i]) for k, v in obs.items() if k != "freewind_measurements")
                for a, i in self.agent_name_mapping.items()}

    def reset(self, seed=None, options=None):
        self.agents = self.possible_agents[:]
        return self._split(self.env.reset(seed=seed, options=options))

    def _join(self, actions):
        first = next(iter(actions.values()))["yaw"]
        if type(first).__module__.startswith("torch"):
            import torch

            return torch.stack([actions[a]["yaw"].reshape(self.num_envs) for a in self.possible_agents], dim=1)
        return np.stack([np.asarray(actions[a]["yaw"], np.float32).reshape(self.num_envs) for a in self.possible_agents],
                        axis=1)

    def step(self, actions: dict):
        """actions: {agent: {"yaw": (B,) or (B, 1)}} for every agent.  Returns per-agent dicts of batched values."""
        missing = [a for a in self.possible_agents if a not in actions]
        if missing:
            raise ValueError(f"Action dict is incomplete. Missing agents: {missing}")
        obs, reward, term, trunc, info = self.env.step({"yaw": self._join(actions)})
        per_obs = self._split(obs)
        rewards = {a: reward for a in self.possible_agents}  # cooperative: same reward for every turbine
        terms = {a: term for a in self.possible_agents}
        truncs = {a: trunc for a in self.possible_agents}
        infos = {a: {"power": info["power"][:, i], "load": info["load"][:, i]} for a, i in self.agent_name_mapping.items()}
        if bool(trunc[0]):
            self.agents = []
        return per_obs, rewards, terms, truncs, infos

    def close(self):
        self.env.close()


class VecLogWrapper:
    """Ring-buffer history for a VecWindFarmEnv: `history[key]` is an array [min(steps, capacity), B, ...]
    in chronological order."""

    def __init__(self, vec_env, capacity: int = 1024):
        self.env = vec_env
        self.capacity = int(capacity)
        self._buf, self._n = {}, 0

    def __getattr__(self, name):
        if name.startswith("_") or name == "env":
            raise AttributeError(name)
        return getattr(self.env, name)

    def _push(self, key, value):
        if key not in self._buf:
            if type(value).__module__.startswith("torch"):
                import torch

                self._buf[key] = torch.empty((self.capacity,) + tuple(value.shape), dtype=value.dtype, device=value.device)
            else:
                self._buf[key] = np.empty((self.capacity,) + value.shape, dtype=value.dtype)
        self._buf[key][self._n % self.capacity] = value

    def reset(self, seed=None, options=None):
        self._buf, self._n = {}, 0
        return self.env.reset(seed=seed, options=options)

    def step(self, actions):
        obs, reward, term, trunc, info = self.env.step(actions)
        for k, v in obs.items():
            self._push("observation/" + k, v)
        self._push("reward", reward)
        self._push("power", info["power"])
        self._push("load", info["load"])
        self._n += 1
        return obs, reward, term, trunc, info

    @property
    def history(self):
        n, cap = self._n, self.capacity
        out = {}
        for k, b in self._buf.items():
            if n <= cap:
                out[k] = b[:n]
            else:
                s = n % cap
                idx = list(range(s, cap)) + list(range(0, s))
                out[k] = b[idx]
        return out
