"""Reward expression and shapers.

The farm reward is the inline expression of reference wfcrl/simple_env.py:78-85 and multiagent_env.py:220-227
(there is no `PowerReward` class in the reference); `power_reward` states it once for the
single-farm envs and for batches.  The shapers post-process that scalar exactly like reference
wfcrl/rewards.py:4-46 (same class names, constructor arguments and call results).
"""
from __future__ import annotations

import abc

import numpy as np


def power_reward(powers_mw, freewind_speed_prev, loads=None, load_coef: float = 0.1):
    """mean_j(P_j [MW] * 1e3 / ws_prev^3) - load_coef * mean |loads|.

    `powers_mw` (N,) with scalar `freewind_speed_prev`, or (B, N) with (B,); `loads` (N, 4) / (B, N, 4) or None."""
    p = np.asarray(powers_mw)
    batched = p.ndim == 2
    ws_cubed = np.asarray(freewind_speed_prev) ** 3
    production = (p * 1e3 / (ws_cubed[:, None] if batched else ws_cubed)).mean(axis=-1)
    if loads is None:
        return production
    fatigue = np.abs(loads).reshape(p.shape[0], -1).mean(axis=1) if batched else np.mean(np.abs(loads))
    return production - load_coef * fatigue


class RewardShaper(abc.ABC):
    """Maps the raw farm reward to what the learner sees; `reset` is called at every env reset."""

    @abc.abstractmethod
    def __call__(self, reward):
        raise NotImplementedError

    def update(self):
        return None

    def reset(self):
        return None


class DoNothingReward(RewardShaper):
    def __call__(self, reward):
        return reward  # identity


class ReferencePercentage(RewardShaper):
    """Relative gain over a fixed baseline reward."""

    def __init__(self, reference: float):
        self.reference = reference

    def __call__(self, reward):
        baseline = self.reference
        return (reward - baseline) / baseline


class StepPercentage(RewardShaper):
    """Relative change from one step to the next; the first step after a reset yields 0."""

    def __init__(self, reference: float = 0.0):
        self.reference = reference

    def __call__(self, reward):
        previous, self.reference = self.reference, reward
        return 0.0 if previous == 0 else (reward - previous) / previous

    def reset(self, reference: float = 0.0):
        self.reference = reference


# BASELINE.json's wording; NOT a class of the reference
PowerReward = power_reward
