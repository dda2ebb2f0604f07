"""GPU box: what a `make(...).step` of the batched env enqueues — run under `rocprofv3 --kernel-trace --stats` to list the kernels
per step, or alone to time the flavours.   python tools/env_step_trace.py [layout] [B] [steps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wfcrl_env_amd import environments as envs
name = sys.argv[1] if len(sys.argv) > 1 else "HornsRev1_Floris"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for kw in (dict(), dict(reuse_buffers=False)):
    env = envs.make(name, env_batch=B, max_num_steps=10 ** 6, **kw)
    env.reset(options={"wind_speed": 8.0, "wind_direction": 270.0})
    N = env.num_turbines
    acts = [torch.from_numpy(np.random.default_rng(i).uniform(-5, 5, (B, N)).astype(np.float32)).cuda() for i in range(4)]
    for i in range(6):
        env.step({"yaw": acts[i % 4]})
    torch.cuda.synchronize()
    for flavour in ("step", "step_light"):
        f = getattr(env, flavour)
        t0 = time.perf_counter()
        for i in range(steps):
            f({"yaw": acts[i % 4]})
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        print(f"{name} x {B} {kw or 'default'} {flavour}: {t / steps * 1e3:.4f} ms per step (enqueue alone {t_enq / steps * 1e3:.4f} ms)", flush=True)
    env.close()
