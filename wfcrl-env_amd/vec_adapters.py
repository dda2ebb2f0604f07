"""Batched multi-agent adaptors and batched history loggers.

`VecAECWindFarmEnv` is the batched flavour of the reference's PettingZoo AEC env (wfcrl/multiagent_env.py:159-254): the
same agent cycle — `agent_iter()`, `last()`, `step(action_i)` one turbine-agent at a time, the farms advancing when the
last agent of the cycle has acted — with every value carrying a leading batch dimension B, and ONE fused kernel launch
per cycle for all B farms.

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

from ._compat import AECEnv, BaseWrapper, agent_selector, spaces
from .mdp import WindFarmMDP


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


class VecAECWindFarmEnv(AECEnv):
    """B farms x N turbine-agents behind the AEC API of reference wfcrl/multiagent_env.py:15-257.

    `step(action)` takes the current agent's action for all B farms, `{"yaw": (B,) or (B, 1)}` (torch CUDA tensor or
    NumPy); when the last agent of the cycle has acted, the joint action goes to `VecWindFarmEnv.step` (one launch).
    Per-agent observations, the cooperative reward, `rewards` / `_cumulative_rewards` / `truncations` / `infos`
    bookkeeping and the dead-step protocol follow the reference line by line; so does its actuation budget, which is
    evaluated per agent on the accumulator value that agent saw at its previous step (multiagent_env.py:196-207,
    241-245: for every agent but the last of a cycle that value is one joint step old).  The inner env therefore runs
    with the kernel's own budget gate disabled (`actuation_budget=inf`).
    """

    metadata = {"name": "vectorized-multiagent-windfarm-aec", "is_parallelizable": True}

    def __init__(self, vec_env):
        self.env = vec_env
        self.num_envs, self.num_turbines = vec_env.num_envs, vec_env.num_turbines
        self.continuous_control, self.controls = vec_env.continuous_control, vec_env.controls
        self.max_num_steps, self.load_coef, self.farm_case = vec_env.max_num_steps, vec_env.load_coef, vec_env.farm_case
        self.possible_agents = [f"turbine_{i + 1}" for i in range(self.num_turbines)]
        self.agent_name_mapping = {a: i for i, a in enumerate(self.possible_agents)}
        sp, ac = vec_env.single_observation_space, vec_env.single_action_space
        self._obs_spaces = {a: {k: spaces.Box(s.low[i], s.high[i]) for k, s in sp.items() if k != "freewind_measurements"}
                            for i, a in enumerate(self.possible_agents)}
        if vec_env.continuous_control:
            self._act_spaces = {a: {k: spaces.Box(s.low[i], s.high[i]) for k, s in ac.items()}
                                for i, a in enumerate(self.possible_agents)}
        else:
            self._act_spaces = {a: {k: s[i] for k, s in ac.items()} for i, a in enumerate(self.possible_agents)}
        self._state = None
        self.agents = []

    def observation_space(self, agent):
        return self._obs_spaces[agent]

    def action_space(self, agent):
        return self._act_spaces[agent]

    def state(self):
        return self._state

    def observe(self, agent):
        k = self.agent_name_mapping[agent]
        return OrderedDict((key, v[:, k]) for key, v in self._state.items() if key != "freewind_measurements")

    def _zeros(self):
        return self._state["yaw"][:, 0] * 0

    def reset(self, seed=None, options=None):
        self._state = self.env.reset(seed=seed, options=options)
        names = self.agents = list(self.possible_agents)
        self.rewards = {a: self._zeros() for a in names}
        self._cumulative_rewards = {a: self._zeros() for a in names}
        self.terminations = dict.fromkeys(names, False)
        self.truncations = dict.fromkeys(names, False)
        self.infos = {a: {} for a in names}
        self.actions = dict.fromkeys(names)
        self.observations = {a: self.observe(a) for a in names}
        self._num_steps = dict.fromkeys(names, 0)
        # accumulated |dyaw| per farm and turbine as of the last joint step (mdp.get_accumulated_actions), and the value
        # each agent stored at its own previous step (multiagent_env.py:241-245)
        self._totals = self._state["yaw"] * 0
        self._seen = {a: self._totals[:, k] for a, k in self.agent_name_mapping.items()}
        self.num_moves = 0
        self._agent_selector = agent_selector(names)
        self.agent_selection = self._agent_selector.next()

    def _as_batch(self, value):
        if type(value).__module__.startswith("torch"):
            return value.reshape(self.num_envs)
        value = np.asarray(value)
        return value.reshape(self.num_envs)

    def step(self, action):
        assert self._state is not None, "Call reset before `step`"
        agent = self.agent_selection
        if self.truncations[agent] or self.terminations[agent]:
            self._was_dead_step(action)
            return
        self._num_steps[agent] += 1
        active = self.controls
        for control in action:
            if control not in active:
                raise ValueError(f"Control `{control}` for agent {agent} is not activated."
                                 f" List of activated controls: {list(active.keys())}")
        if any(control not in action for control in active):
            raise ValueError(f"Action {action} for agent {agent} is incomplete."
                             f" List of needed controls: {active.keys()}")
        a = self._as_batch(action["yaw"])
        # actuation budget of THIS agent, on the accumulator it saw at its previous step; blocked farms have the raw
        # action zeroed in the caller's array, as the reference does (float32 arithmetic, as NumPy performs it there)
        rate, dt = WindFarmMDP.ACTUATORS_RATE["yaw"], self.farm_case.dt
        seen = self._seen[agent]
        if type(seen).__module__.startswith("torch"):
            import torch

            # tensor / tensor divisions: torch turns `tensor / python_scalar` into a multiplication by the reciprocal,
            # which rounds differently from NumPy's float32 division exactly where discrete actions put the accumulator
            # ON the threshold
            c = torch.tensor([rate, float(self._num_steps[agent]), float(dt), 0.1], dtype=torch.float32, device=seen.device)
            blocked = ((seen / c[0]) / c[1]) / c[2] >= c[3]
            if not type(a).__module__.startswith("torch"):
                a = torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=seen.device)
            a = a.to(torch.float32)
            a.masked_fill_(blocked, 0.0)
        else:
            f32 = np.float32
            blocked = ((seen.astype(f32) / f32(rate)) / f32(self._num_steps[agent])) / f32(dt) >= f32(0.1)
            a = a.astype(np.float32, copy=False)
            a[blocked] = 0.0
        self._cumulative_rewards[agent] = 0
        self.actions[agent] = a

        if self._agent_selector.is_last():
            joint = self._stack([self.actions[n] for n in self.possible_agents])
            obs, reward, term, trunc, info = self.env.step({"yaw": joint})
            self._state = obs
            truncated = bool(trunc[0])
            for name, k in self.agent_name_mapping.items():
                if name not in self.agents:
                    continue
                self.infos[name]["load"] = info["load"][:, k]
                self.rewards[name] = reward  # cooperative: one reward for every turbine
                self.observations[name] = self.observe(name)
                self.truncations[name], self.terminations[name] = truncated, False
                self.infos[name]["power"] = info["power"][:, k]
            self.num_moves += 1
            self._totals = self._accumulated()
        else:
            self._clear_rewards()  # nobody is paid until the cycle is complete

        self._seen[agent] = self._totals[:, self.agent_name_mapping[agent]]
        self.agent_selection = self._agent_selector.next()
        self._accumulate_rewards()

    def _stack(self, per_agent):
        if type(per_agent[0]).__module__.startswith("torch"):
            import torch

            return torch.stack(per_agent, dim=1)
        return np.stack(per_agent, axis=1)

    def _accumulated(self):
        fi = self.env.fi
        if self.env.return_torch:
            return fi.env_get_state(as_torch=True)["acc"]
        return fi.env_get_state()["acc"]

    def _clear_rewards(self):
        for a in self.rewards:
            self.rewards[a] = self._zeros()

    def close(self):
        self.env.close()


def _copy(v):
    if hasattr(v, "clone"):
        return v.clone()
    return np.array(v, copy=True) if isinstance(v, np.ndarray) else v


class VecAECLogWrapper(BaseWrapper):
    """Per-agent history recorded on `last()` (reference wfcrl/wrappers.py:24-58), entries batched over the B farms."""

    def __init__(self, env):
        super().__init__(env)
        self.history = {a: {k: [] for k in ("observation", "reward", "load", "power")} for a in self.env.possible_agents}

    def last(self):
        agent = self.env.agent_selection
        result = self.env.last()
        h = self.history[agent]
        # independent copies, as the reference's AECLogWrapper stores independent values: the env may hand out views of
        # buffers it reuses (VecWindFarmEnv(reuse_buffers=True)) or of the joint step's tensors
        h["observation"].append({k: _copy(v) for k, v in result[0].items()} if isinstance(result[0], dict) else _copy(result[0]))
        h["reward"].append(_copy(result[1]))
        for key in ("power", "load"):
            if key in result[4]:
                h[key].append(_copy(result[4][key]))
        return result

    def reset(self, seed=None, options=None):
        self.history = {a: {k: [] for k in ("observation", "reward", "load", "power")} for a in self.env.possible_agents}
        return self.env.reset(seed, options)


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
