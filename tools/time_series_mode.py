"""GPU box: HornsRev1 x 65536 — shared wind vs series playback (grouped tables) vs binned reset directions vs a continuous
direction per farm.  usage: python tools/time_series_mode.py [T] [step_deg] [B]"""
import json, os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from wfcrl_env_amd.backend import WfStep
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
step = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
L = json.load(open("wfcrl-env_amd/environments/layouts.json"))["HornsRev1_"]
B, N = (int(sys.argv[3]) if len(sys.argv) > 3 else 65536), 80
print(f"# HornsRev1 x {B}")
w = WfStep(L["xcoords"], L["ycoords"], env_batch=B)
g = torch.Generator(device="cuda").manual_seed(1)
yaw = (torch.rand((B, N), device="cuda", generator=g) * 60 - 30).float()
rng = np.random.default_rng(0)
def timed(label, per_step=None, K=10):
    out = w.step(yaw); w.sync()
    best = 1e9
    for r in range(3):
        w.timing_begin()
        for _ in range(K):
            if per_step: per_step()
            w.step(yaw, out)
        best = min(best, w.timing_end() / K)
    k = w.kernel_info()
    print(f"{label:52s} {best:.3f} ms/step  {B / best * 1e3:.3e} farm-steps/s  table={k['pair_table']} groups={k['direction_groups']} blocks={k['grid_blocks']}", flush=True)
w.set_wind(8.0, 270.0); timed("shared wind")
series = np.stack([rng.uniform(6, 12, T + 64), rng.uniform(240, 300, T + 64)], axis=1)
w.set_wind_series(series[:T + 40]); timed(f"series playback, T = {T + 40} rows, tick every step", per_step=w.wind_series_step)
w.set_wind_series(series[:T]); timed(f"series, T = {T} rows, no tick")
w.sample_wind(3, direction_step=step); timed(f"binned reset directions, step {step} deg")
w.sample_wind(3); timed("continuous direction per farm (on the fly)")
os.environ["WF_NO_PAIR_TABLE"] = "1"
w2 = WfStep(L["xcoords"], L["ycoords"], env_batch=B); w, w_old = w2, w
w.set_wind_series(series[:T + 40]); timed(f"series playback without groups (on the fly + geometry per tick)", per_step=w.wind_series_step)
