"""Reward shapers — post-processing of the scalar farm reward (reference wfcrl/rewards.py:4-46).

The power reward itself is the inline expression of reference simple_env.py:78-85 (there is no
`PowerReward` class in the reference); `power_reward` below states it once for
both env flavours and for the batched env.
"""
from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np


class RewardShaper(ABC):
    @abstractmethod
    def __call__(self, reward):
        ...

    def update(self):
        return None

    def reset(self):
        return None


class DoNothingReward(RewardShaper):
    """Identity."""

    def __call__(self, reward):
        return reward


class ReferencePercentage(RewardShaper):
    """(r - ref) / ref against a fixed reference."""

    def __init__(self, reference: float):
        self.reference = reference

    def __call__(self, reward):
        return (reward - self.reference) / self.reference


class StepPercentage(RewardShaper):
    """Relative change with respect to the previous reward; 0 on the first call after reset."""

    def __init__(self, reference: float = 0.0):
        self.reference = reference

    def __call__(self, reward):
        shaped = 0.0 if self.reference == 0 else (reward - self.reference) / self.reference
        self.reference = reward
        return shaped

    def reset(self, reference: float = 0.0):
        self.reference = reference


def power_reward(powers_mw, freewind_speed_prev, loads=None, load_coef: float = 0.1):
    """mean_j(P_j[MW] * 1e3 / ws_prev^3) - load_coef * mean|loads|   (simple_env.py:78-84,
    multiagent_env.py:220-226).  Works on (N,) arrays and on (B, N) batches (ws_prev (B,))."""
    powers_mw = np.asarray(powers_mw)
    ws3 = np.asarray(freewind_speed_prev) ** 3
    if powers_mw.ndim == 2:
        r = (powers_mw * 1e3 / ws3[:, None]).mean(axis=1)
        pen = 0 if loads is None else np.abs(loads).reshape(powers_mw.shape[0], -1).mean(axis=1)
    else:
        r = (powers_mw * 1e3 / ws3).mean()
        pen = 0 if loads is None else np.mean(np.abs(loads))
    return r - load_coef * pen


# Alias for BASELINE.json's wording; NOT a class of the reference.
PowerReward = power_reward
