"""GPU box: step rate with several layouts in one batch (wf_set_layouts): K jittered copies of a layout, farms assigned
round-robin; one wind for the batch (layouts = groups of a grouped launch, pair-table path), a wind per farm (on the fly),
and a layout per farm.  Also the cost of setting the wind (a geometry per farm).
  python tools/layouts_rate.py [layout] [B]"""
import json, os, sys, time
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wfcrl_env_amd.backend import WfStep

name = sys.argv[1] if len(sys.argv) > 1 else "HornsRev1_"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
lay = json.load(open(os.path.join(ROOT, "wfcrl-env_amd", "environments", "layouts.json")))[name]
x0, y0, N = np.array(lay["xcoords"]), np.array(lay["ycoords"]), lay["num_turbines"]
rng = np.random.default_rng(0)
yaw = (torch.rand((B, N), device="cuda") * 60 - 30).float()
ws_f, wd_f = np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360


def rate(w, label):
    info = w.kernel_info()
    out = w.step(yaw)
    for _ in range(10):  # (clocks ramp up over the first launches)
        w.step(yaw, out)
    w.sync()
    w.timing_begin()
    for _ in range(10):
        w.step(yaw, out)
    ms = w.timing_end() / 10
    k = f'{"one-block" if info["one_block_kernel"] else "slot"} {info["lanes_per_env"]}x{info["slots_per_lane"]}, table {info["pair_table"]}, groups {info["direction_groups"]}, blocks {info["grid_blocks"]}'
    print(f"{name} B={B} {label} [{k}]: {ms:.3f} ms per step, {B / ms * 1e3:.3e} farm-steps/s", flush=True)


for K in (1, 2, 8, 64, B):
    X = x0[None, :] + rng.uniform(-60, 60, (K, N)) * (K > 1)
    Y = y0[None, :] + rng.uniform(-60, 60, (K, N)) * (K > 1)
    w = WfStep(x0, y0, env_batch=B)
    if K > 1:
        w.set_layouts(X, Y, None if K == B else (np.arange(B) % K).astype(np.int32))
    for wind, (ws, wd) in (("one wind", (8.0, 263.0)), ("a wind per farm", (ws_f, wd_f))):
        w.set_wind(ws, wd); w.sync()
        t0 = time.perf_counter()
        w.set_wind(ws, wd); w.sync()
        t_set = (time.perf_counter() - t0) * 1e3
        rate(w, f"{K} layout{'s' if K > 1 else ''}, {wind} (set_wind {t_set:.2f} ms)")
    w.close()
