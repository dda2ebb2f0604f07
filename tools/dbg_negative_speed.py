import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from wfcrl_env_amd.backend import WfStep
d = np.load("tests/golden/negative_rotor_speed_case.npz")
model = eval(str(d["model"]))
x, y, yaw = d["x"], d["y"], d["yaw"]
for choice in (dict(slot=(32, 4), one_block="4"), dict(slot=(32, 4), one_block=False), dict(one_block="2x2"), None):
    for mode in (0, 1):
        w = WfStep(x, y, env_batch=yaw.shape[0], model=dict(model), kernel_choice=choice)
        w.set_risk_resolve(mode)
        w.set_wind(float(d["ws"][0]), float(d["wd"][0]))
        out = w.step(yaw)
        st = w.resolve_stats() if mode else {}
        print(choice, "mode", mode, "flags", w.risk_flags(), "info", {k: w.kernel_info()[k] for k in ("lanes_per_env", "slots_per_lane", "one_block_kernel")}, "stats", st, "ws min", float(np.asarray(out["wind_speed"]).min()))
        w.close()
