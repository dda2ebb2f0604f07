"""Fallback shims for `gymnasium` / `pettingzoo`, used ONLY when the real packages are absent
(they are not installed in the build image or on the GPU box).  `from ._compat import gym, spaces,
AECEnv, agent_selector, BaseWrapper` yields the real classes when importable."""
try:  # pragma: no cover - depends on the environment
    import gymnasium as gym
    from gymnasium import spaces

    HAVE_GYMNASIUM = True
except ImportError:
    from . import gym_shim as gym
    from .gym_shim import spaces

    HAVE_GYMNASIUM = False

try:  # pragma: no cover
    from pettingzoo import AECEnv
    from pettingzoo.utils import agent_selector
    from pettingzoo.utils.wrappers import BaseWrapper

    HAVE_PETTINGZOO = True
except ImportError:
    from .pz_shim import AECEnv, BaseWrapper, agent_selector

    HAVE_PETTINGZOO = False

__all__ = ["gym", "spaces", "AECEnv", "agent_selector", "BaseWrapper", "HAVE_GYMNASIUM", "HAVE_PETTINGZOO"]
