"""GPU box: per-step latency of the B = 1 drop-in interface (HipFlorisInterface.update_command + accessors)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfcrl_env_amd import environments as envs
for name in ("Turb3_Row1_Floris", "Ablaincourt_Floris", "Turb_TCRWP_Floris", "HornsRev1_Floris", "HornsRev2_Floris"):
    env = envs.make(name, max_num_steps=10_000, log=False)
    env.reset(seed=0)
    n = env.num_turbines
    rng = np.random.default_rng(0)
    acts = [{"yaw": rng.uniform(-1, 1, n)} for _ in range(300)]
    for a in acts[:50]: env.step(a)
    t = time.perf_counter()
    for a in acts[50:]: env.step(a)
    dt = (time.perf_counter() - t) / 250
    it = env.mdp.interface
    y = np.zeros(n)
    t = time.perf_counter()
    for _ in range(250): it.update_command(y)
    du = (time.perf_counter() - t) / 250
    print(f"{name:24s} N={n:3d}  env.step {dt*1e6:7.1f} us   interface.update_command {du*1e6:7.1f} us")
