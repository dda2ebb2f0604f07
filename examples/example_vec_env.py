"""B wind farms per kernel launch: the batched env (state, transition, reward on the device).
Run from the repo root on an MI355X:  python examples/example_vec_env.py [env_batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfcrl_env_amd import environments as envs  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
env = envs.make("HornsRev1_Floris", env_batch=B, max_num_steps=200, wind_sampling="device")
obs = env.reset(seed=0)  # one wind per farm, sampled on the device
print({k: tuple(v.shape) for k, v in obs.items()})
ret = torch.zeros(B, device="cuda")
t0 = time.perf_counter()
steps = 0
while True:
    # a toy "policy": every turbine creeps towards +10 deg of yaw, within the 5 deg/step increment limit
    action = (10.0 - obs["yaw"]).clamp(-5, 5)
    obs, reward, terminated, truncated, info = env.step({"yaw": action})
    ret += reward
    steps += 1
    if bool(truncated[0]):
        break
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{steps} joint steps of {B} farms x {env.num_turbines} turbines in {dt:.2f} s = {B * steps / dt:.3e} farm-steps/s")
print(f"mean episode return {ret.mean().item():.3f}, farm power of env 0: {info['power'][0].sum().item():.2f} MW")
env.close()
