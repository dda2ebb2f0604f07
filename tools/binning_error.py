"""CPU (float64 oracle): what rounding every farm's wind direction to a grid does to the answer — the price of the binned
reset sampling (wf_wind_sample_binned: about twice the throughput of a continuous direction per farm) —
per layout and grid step, under the reference's reset distribution (wfcrl/mdp.py:237-258).
  python tools/binning_error.py [B] > profiles/archive/r03_binning_error.txt"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle

L = json.load(open(os.path.join(ROOT, "wfcrl-env_amd", "environments", "layouts.json")))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(5)
print("# per-turbine power |dP| / max(P, 1 kW) and farm power |dP_farm| / P_farm of direction-binned vs continuous wind, "
      f"{B} farms, yaw ~ U(-30, 30), ws = clip(8 Weibull(8), 3, 28), wd = N(270, 20) mod 360")
for name in ("Ablaincourt_", "Turb16_Row5_", "Turb_TCRWP_", "WMR_", "HornsRev1_", "HornsRev2_"):
    l = L[name]; N = l["num_turbines"]; x, y = l["xcoords"], l["ycoords"]
    yaw = rng.uniform(-30, 30, (B, N))
    ws = np.clip(8 * rng.weibull(8, B), 3, 28)
    wd = rng.normal(270, 20, B) % 360
    ref = c_oracle.farm_step_batch(x, y, ws, wd, yaw)
    for step in (5.0, 2.0, 1.0, 0.5, 0.25):
        wdb = (np.round(wd / step) * step) % 360
        got = c_oracle.farm_step_batch(x, y, ws, wdb, yaw)
        pt = np.abs(got["power"] - ref["power"]) / np.maximum(ref["power"], 1e3)
        pf = np.abs(got["power"].sum(1) - ref["power"].sum(1)) / ref["power"].sum(1)
        dd = np.abs(((got["wind_direction"] - ref["wind_direction"]) + 180) % 360 - 180 - 0.0)
        print(f"{name:14s} step {step:5.2f} deg: per turbine median {np.median(pt):.1e} p99 {np.quantile(pt, 0.99):.1e} max {pt.max():.1e} | "
              f"farm power median {np.median(pf):.1e} p99 {np.quantile(pf, 0.99):.1e} max {pf.max():.1e} | local direction max {dd.max():.2f} deg", flush=True)
