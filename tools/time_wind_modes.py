import os, sys, json
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from wfcrl_env_amd.backend import WfStep
L = json.load(open("wfcrl-env_amd/environments/layouts.json"))["HornsRev1_"]
B = 65536
w = WfStep(L["xcoords"], L["ycoords"], env_batch=B)
rng = np.random.default_rng(0)
yaw = (torch.rand((B, 80), device="cuda") * 60 - 30).float()
for name, ws, wd in [("shared", 8.0, 270.0), ("speed per farm, one direction", rng.uniform(4, 16, B), np.full(B, 270.0)), ("wind per farm", rng.uniform(4, 16, B), rng.normal(270, 20, B) % 360)]:
    w.set_wind(ws, wd)
    out = w.step(yaw); w.sync()
    w.timing_begin()
    for _ in range(20): w.step(yaw, out)
    ms = w.timing_end() / 20
    print(f"{name:32s} {ms:.3f} ms/step  {B / ms * 1e3:.3e} farm-steps/s  table={w.kernel_info()['pair_table']}")
