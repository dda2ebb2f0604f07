"""GPU box: wind-direction error of ONE turbine (its own rotation vortex on its own rotor grid) on the fly against the
table path, over num_eps, rotor diameter, hub height, shear and wind speed — the error scaled with 1 / num_eps and
changed sign with D: the float32 cancellation of 1 - exp(-r^2 / eps^2) at r^2 = 2 num_eps^2."""
import os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from oracle import c_oracle
from oracle.floris_gch_numpy import ModelParams
from wfcrl_env_amd.backend import WfStep
np.set_printoptions(linewidth=250, precision=4)
x, y = np.array([0.0]), np.array([0.0])
B = 2
for D, hh, shear, ne in ((100.5, 0.9, 0.0, 0.01), (100.5, 0.9, 0.0, 0.1), (100.5, 0.9, 0.0, 1.0), (126.0, 90.0 / 126.0, 0.12, 0.01), (100.5, 0.9, 0.12, 0.01), (126.0, 0.9, 0.0, 0.01), (100.5, 90.0 / 126.0, 0.0, 0.01)):
    model = dict(rotor_diameter=D, hub_height=hh * D, shear=shear, num_eps=ne)
    mp = ModelParams(D=D, HH=hh * D, shear=shear, num_eps=ne)
    yaw = np.zeros((B, 1), np.float32)
    for wsv in (9.0, 6.0):
        ws, wd = np.full(B, wsv), np.full(B, 360.0)
        ref = c_oracle.farm_step_batch(x, y, ws, wd, yaw.astype(np.float64), mp, margin=True)
        row = []
        for label, choice in (("fly", dict(one_block=False, pair_table=False)), ("table", dict(one_block=False))):
            w = WfStep(x, y, env_batch=B, model=model, kernel_choice=choice)
            w.set_wind(ws, wd)
            got = w.step(yaw)
            row.append(f"{label} wd err {(got['wind_direction'].astype(np.float64) - ref['wind_direction'])[0, 0]:.3e}")
            w.close()
        print(f"D {D} HH {hh:.3f} D shear {shear} num_eps {ne} ws {wsv}: " + " | ".join(row) + f" | wd-360 ref {ref['wind_direction'][0, 0] - 360:.5f}", flush=True)
