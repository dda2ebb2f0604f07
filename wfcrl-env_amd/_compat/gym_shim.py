"""Minimal stand-in for the parts of gymnasium==0.29 the env surface touches:
spaces.Box / Dict / Discrete / MultiDiscrete, Env, Wrapper.  Behaviour (shapes, dtypes, Dict key
ordering, reprs) follows gymnasium so that the notebook fixtures in tests/golden hold."""
from __future__ import annotations

from collections import OrderedDict

import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None, seed=None):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._rng = np.random.default_rng(seed)

    @property
    def shape(self):
        return self._shape

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]

    def contains(self, x):  # pragma: no cover - overridden
        raise NotImplementedError

    def __contains__(self, x):
        return self.contains(x)


def _short(a: np.ndarray) -> str:
    return str(a.flat[0]) if a.size and np.min(a) == np.max(a) else str(a)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        dtype = np.dtype(dtype)
        if shape is not None:
            shape = tuple(int(s) for s in shape)
        elif isinstance(low, np.ndarray):
            shape = low.shape
        elif isinstance(high, np.ndarray):
            shape = high.shape
        else:
            shape = (1,)  # scalar bounds without a shape -> (1,)  (demo.ipynb:773-774)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape).astype(dtype)
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape).astype(dtype)
        super().__init__(shape, dtype, seed)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({_short(self.low)}, {_short(self.high)}, {self.shape}, {self.dtype})"

    def __eq__(self, other):
        return isinstance(other, Box) and self.shape == other.shape and np.array_equal(self.low, other.low) \
            and np.array_equal(self.high, other.high)


class Discrete(Space):
    def __init__(self, n, start=0, seed=None):
        self.n, self.start = int(n), int(start)
        super().__init__((), np.int64, seed)

    def sample(self):
        return int(self.start + self._rng.integers(self.n))

    def contains(self, x):
        return self.start <= int(x) < self.start + self.n

    def __repr__(self):
        return f"Discrete({self.n})" if self.start == 0 else f"Discrete({self.n}, start={self.start})"

    def __eq__(self, other):
        return isinstance(other, Discrete) and (self.n, self.start) == (other.n, other.start)


class MultiDiscrete(Space):
    def __init__(self, nvec, dtype=np.int64, seed=None):
        self.nvec = np.asarray(nvec, dtype=dtype)
        super().__init__(self.nvec.shape, dtype, seed)

    def sample(self):
        return (self._rng.random(self.nvec.shape) * self.nvec).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= 0) and np.all(x < self.nvec))

    def __getitem__(self, i):
        sub = self.nvec[i]
        return Discrete(int(sub)) if np.ndim(sub) == 0 else MultiDiscrete(sub, self.dtype)

    def __len__(self):
        return len(self.nvec)

    def __repr__(self):
        return f"MultiDiscrete({self.nvec})"


class Dict(Space):
    """A plain dict is sorted by key, an OrderedDict keeps insertion order (gymnasium semantics)."""

    def __init__(self, spaces=None, seed=None, **kw):
        if spaces is None:
            spaces = kw
        if isinstance(spaces, OrderedDict):
            items = list(spaces.items())
        elif isinstance(spaces, dict):
            try:
                items = sorted(spaces.items())
            except TypeError:
                items = list(spaces.items())
        else:
            items = list(spaces)
        self.spaces = OrderedDict(items)
        super().__init__(None, None, seed)

    def sample(self):
        return OrderedDict((k, s.sample()) for k, s in self.spaces.items())

    def contains(self, x):
        return isinstance(x, dict) and x.keys() == self.spaces.keys() and all(x[k] in s for k, s in self.spaces.items())

    def __getitem__(self, k):
        return self.spaces[k]

    def __iter__(self):
        return iter(self.spaces)

    def __len__(self):
        return len(self.spaces)

    def keys(self):
        return self.spaces.keys()

    def values(self):
        return self.spaces.values()

    def items(self):
        return self.spaces.items()

    def __repr__(self):
        return "Dict(" + ", ".join(f"{k!r}: {s}" for k, s in self.spaces.items()) + ")"


class _SpacesNamespace:
    Space, Box, Discrete, MultiDiscrete, Dict = Space, Box, Discrete, MultiDiscrete, Dict


spaces = _SpacesNamespace()


class Env:
    metadata: dict = {}
    action_space = None
    observation_space = None

    def reset(self, seed=None, options=None):  # pragma: no cover
        raise NotImplementedError

    def step(self, action):  # pragma: no cover
        raise NotImplementedError

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    """Forwards unknown attributes to the wrapped env."""

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_") or name == "env":
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def action_space(self):
        return self.env.action_space

    @property
    def observation_space(self):
        return self.env.observation_space

    @property
    def metadata(self):
        return self.env.metadata

    def reset(self, seed=None, options=None):
        return self.env.reset(seed=seed, options=options)

    def step(self, action):
        return self.env.step(action)

    def close(self):
        return self.env.close()

    @property
    def unwrapped(self):
        return self.env.unwrapped
