"""GPU box: throughput of a model the float32 kernels do not carry (wind_veer != 0): every farm through the float64 kernel.
  python tools/veer_rate.py [layout] [B]"""
import json, os, sys
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
for label, model, mode in (("veer 0, float32 kernels", None, 0), ("veer 3 deg, float32 VEER kernel", dict(veer=3.0), 0), ("veer 3 deg, float32 + float64 re-solve of the flagged farms", dict(veer=3.0), 1),
                           ("veer 0, float64 kernel on every farm (wf_set_risk_resolve mode 2)", None, 2)):
    w = WfStep(lay["xcoords"], lay["ycoords"], env_batch=B, model=model)
    w.set_risk_resolve(mode)
    w.set_wind(8.0, 270.0)
    out = w.step(yaw); w.sync()
    w.timing_begin()
    for _ in range(3):
        w.step(yaw, out)
    ms = w.timing_end() / 3
    print(f"{name} B={B} {label}: {ms:.2f} ms per step, {B / ms * 1e3:.3e} farm-steps/s", flush=True)
    w.close()
