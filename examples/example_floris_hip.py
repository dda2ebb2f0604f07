"""Multi-agent (PettingZoo AEC) episode on the HIP backend — the scenario of the reference's
examples/example_floris.py: Ablaincourt, one agent per turbine, turbine_1 yaws by 15 deg at its 20th move, everyone
else holds; reward = relative step-to-step change of (power - load proxy).
Needs an MI355X (there is no CPU fallback).  Run from the repo root:  python examples/example_floris_hip.py"""
import os
import sys
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfcrl_env_amd import environments as envs  # noqa: E402
from wfcrl_env_amd.rewards import StepPercentage  # noqa: E402

KICK_AGENT, KICK_MOVE, KICK_DEG = "turbine_1", 20, 15.0


def policy(agent: str, move: int) -> dict:
    return {"yaw": np.array([KICK_DEG if (agent, move) == (KICK_AGENT, KICK_MOVE) else 0.0])}


def main():
    env = envs.make("Dec_Ablaincourt_Floris", max_num_steps=100, reward_shaper=StepPercentage(), load_coef=1)
    env.reset()
    returns, moves, finished = defaultdict(float), defaultdict(int), set()
    for agent in env.agent_iter():
        _, reward, terminated, truncated, _ = env.last()
        returns[agent] += float(np.ravel(reward)[0])
        if terminated or truncated:
            finished.add(agent)
        if agent in finished:
            env.step(None)  # a finished agent must pass None (PettingZoo AEC contract)
            continue
        env.step(policy(agent, moves[agent]))
        moves[agent] += 1
    print("moves per agent:", dict(moves))
    print("total reward per agent:", {a: round(r, 6) for a, r in returns.items()})


if __name__ == "__main__":
    main()
