"""The reference's examples/example_floris.py on the HIP backend: PettingZoo-style AEC loop, one agent per turbine.
Needs an MI355X (there is no CPU fallback).  Run from the repo root:  python examples/example_floris_hip.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfcrl_env_amd import environments as envs  # noqa: E402
from wfcrl_env_amd.rewards import StepPercentage  # noqa: E402

env = envs.make("Dec_Ablaincourt_Floris", max_num_steps=100, reward_shaper=StepPercentage(), load_coef=1)


def dummy_policy(agent, i):
    if agent == "turbine_1" and i == 20:
        return {"yaw": np.array([15.0])}
    return {"yaw": np.array([0])}


env.reset()
r = {agent: 0 for agent in env.possible_agents}
done = {agent: False for agent in env.possible_agents}
num_steps = {agent: 0 for agent in env.possible_agents}
for agent in env.agent_iter():
    observation, reward, termination, truncation, info = env.last()
    done[agent] = done[agent] or termination or truncation
    r[agent] += reward
    if done[agent]:
        action = None
    else:
        action = dummy_policy(agent, num_steps[agent])
        num_steps[agent] += 1
    env.step(action)

print(f"Total reward = {r}")
