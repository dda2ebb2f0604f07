"""GPU box: where do the float64 kernels' 6e-7 against the oracle come from?  The negative-rotor-speed case (tests/golden) is
the amplifier: mode 2 (every farm in float64) against the C oracle with one model ingredient switched at a time."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from oracle import c_oracle
from oracle.floris_gch_numpy import ModelParams
from wfcrl_env_amd.backend import WfStep
d = np.load("tests/golden/negative_rotor_speed_case.npz")
base = eval(str(d["model"]))
x, y, yaw = d["x"], d["y"], d["yaw"]
ws, wd = float(d["ws"][0]), float(d["wd"][0])
ren = {"rotor_diameter": "D", "hub_height": "HH"}
variants = [("as fuzzed", {}), ("no transverse velocities", dict(enable_transverse_velocities=False)), ("no yaw-added recovery", dict(enable_yaw_added_recovery=False)),
            ("no secondary steering", dict(enable_secondary_steering=False)), ("all three off", dict(enable_transverse_velocities=False, enable_yaw_added_recovery=False, enable_secondary_steering=False)),
            ("dm 1.0", dict(dm=1.0)), ("ad bd 0", dict(ad=0.0, bd=0.0)), ("deflection set = velocity set", dict(defl_ka=base["ka"], defl_kb=base["kb"], defl_alpha=base["alpha"], defl_beta=base["beta"])),
            ("default alpha beta", dict(alpha=0.58, beta=0.077)), ("zero yaw", None)]
for name, over in variants:
    m = dict(base)
    yw = yaw
    if over is None: yw = np.zeros_like(yaw)
    else: m.update(over)
    mp = ModelParams(**{ren.get(k, k): v for k, v in m.items()})
    ref = c_oracle.farm_step_batch(x, y, ws, wd, yw.astype(np.float64), mp)
    w = WfStep(x, y, env_batch=yw.shape[0], model=dict(m))
    w.set_risk_resolve(2)
    w.set_wind(ws, wd)
    out = {k: np.asarray(v, dtype=np.float64) for k, v in w.step(yw).items()}
    w.close()
    free = np.abs(ref["wind_speed"]).max()
    ews = np.abs(out["wind_speed"] - ref["wind_speed"]) / free
    estd = np.abs(out["load"][..., 1] - ref["load"][..., 1]) / free
    eti = np.abs(out["load"][..., 0] - ref["load"][..., 0])
    b, t = np.unravel_index(np.argmax(ews), ews.shape)
    print(f"{name:32s} ws err / free stream: max {ews.max():.2e} (farm {b} turbine {t}, ref ws {ref['wind_speed'][b, t]:.4f})  std_u {estd.max():.2e}  TI {eti.max():.2e}  neg rotors {int((ref['wind_speed'] <= 0).sum())}", flush=True)
