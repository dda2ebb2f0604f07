"""GPU box: throughput of a model with wind_veer != 0 — the float32 VEER instantiations (register-slot kernel and the
one-block kernel's families), shared wind and a wind per farm, beside veer 0 and the float64 kernel on every farm.
  python tools/veer_rate.py [layout] [B]"""
import json, os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wfcrl_env_amd.backend import WfStep

name = sys.argv[1] if len(sys.argv) > 1 else "HornsRev1_"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
lay = json.load(open(os.path.join(ROOT, "wfcrl-env_amd", "environments", "layouts.json")))[name]
N = lay["num_turbines"]
yaw = (torch.rand((B, N), device="cuda") * 60 - 30).float()
rng = np.random.default_rng(0)
ws_f, wd_f = np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360
V = dict(veer=3.0)
for label, model, mode, choice in (
        ("veer 0", None, 0, None), ("veer 3 deg, picked kernel", V, 0, None), ("veer 3 deg, register-slot kernel", V, 0, dict(one_block=False, fly_one_block=False)),
        ("veer 3 deg, one-block 4x1", V, 0, dict(one_block="4x1")), ("veer 3 deg, one-block 4x2", V, 0, dict(one_block="4x2")),
        ("veer 3 deg, one-block 2x2", V, 0, dict(one_block="2x2")), ("veer 3 deg, picked + float64 re-solve of the flagged farms", V, 1, None),
        ("veer 0, float64 kernel on every farm (mode 2)", None, 2, None)):
    w = WfStep(lay["xcoords"], lay["ycoords"], env_batch=B, model=model, kernel_choice=choice)
    w.set_risk_resolve(mode)
    for wind, (ws, wd) in (("shared wind", (8.0, 270.0)), ("wind per farm", (ws_f, wd_f))):
        if mode == 2 and wind != "shared wind":
            continue
        w.set_wind(ws, wd)
        info = w.kernel_info()
        out = w.step(yaw)
        for _ in range(10 if mode != 2 else 0):  # (clocks ramp up over the first launches)
            w.step(yaw, out)
        w.sync()
        w.timing_begin()
        for _ in range(3):
            w.step(yaw, out)
        ms = w.timing_end() / 3
        k = f'{"one-block" if info["one_block_kernel"] else "slot"} {info["lanes_per_env"]}x{info["slots_per_lane"]} {info["vgprs"]}v'
        print(f"{name} B={B} {label}, {wind} [{k}]: {ms:.2f} ms per step, {B / ms * 1e3:.3e} farm-steps/s", flush=True)
    w.close()
