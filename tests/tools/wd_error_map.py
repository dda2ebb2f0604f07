"""GPU box: map of the wind-direction and power error over an 8 x 14 grid at exactly 360 deg, on-the-fly kernel against
the table path (float64 oracle as the reference) — how the core-factor cancellation of DESIGN.md §5 showed up: a bias
growing by one step per upstream row.  usage: python tests/tools/wd_error_map.py [zero]   (zero: all yaw angles 0)"""
import os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from oracle import c_oracle
from oracle.floris_gch_numpy import ModelParams, farm_step
from wfcrl_env_amd.backend import WfStep
D = 100.5
gx, gy = np.meshgrid(np.arange(8) * 5 * D, np.arange(14) * 4 * D, indexing="ij")
x, y = gx.ravel(), gy.ravel()
N, B = x.size, 4
rng = np.random.default_rng(0)
yaw = rng.uniform(-35, 35, (B, N)).astype(np.float32)
if len(sys.argv) > 1 and sys.argv[1] == "zero":
    yaw[:] = 0
ws = np.full(B, 9.0)
hh, shear = 0.9, 0.0
model = dict(rotor_diameter=D, hub_height=hh * D, shear=shear)
mp = ModelParams(D=D, HH=hh * D, shear=shear)
wd = np.full(B, 360.0)
ref = c_oracle.farm_step_batch(x, y, ws, wd, yaw.astype(np.float64), mp, margin=True)
np.set_printoptions(linewidth=250, precision=2, suppress=False)
for label, choice in (("fly", dict(one_block=False, pair_table=False)), ("table", dict(one_block=False))):
    w = WfStep(x, y, env_batch=B, model=model, kernel_choice=choice)
    w.set_wind(ws, wd)
    got = w.step(yaw)
    e = (got["wind_direction"].astype(np.float64) - ref["wind_direction"])[0].reshape(8, 14)   # [lateral column (x index), along-wind row (y index)]
    ep = ((got["power"].astype(np.float64) - ref["power"]) / np.maximum(ref["power"], 1e3))[0].reshape(8, 14)
    print(label, "wd error [x index 0..7][y index 0..13] (wind from north: y index 13 is upstream)")
    print(e)
    print(label, "power rel error")
    print(ep)
    w.close()
