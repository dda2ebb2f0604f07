"""GPU box: the shared-wind step time as a function of the wind direction (how many pairs pass the 15 D / 2 D gates
depends on it) — the yardstick for the grouped modes, whose groups spread over directions.
  python tools/direction_sweep.py [layout] [B]"""
import json, os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from wfcrl_env_amd.backend import WfStep
name = sys.argv[1] if len(sys.argv) > 1 else "HornsRev1_"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
L = json.load(open("wfcrl-env_amd/environments/layouts.json"))[name]
N = L["num_turbines"]
w = WfStep(L["xcoords"], L["ycoords"], env_batch=B)
g = torch.Generator(device="cuda").manual_seed(1)
yaw = (torch.rand((B, N), device="cuda", generator=g) * 60 - 30).float()
w.set_wind(8.0, 270.0)
out = w.step(yaw)
for _ in range(20):
    w.step(yaw, out)
w.sync()
ts = []
print(f"# {name} x {B}: ms per step of the shared-wind path by wind direction (8 m/s)")
dirs = np.arange(230.0, 311.0, 2.5)
if os.environ.get("WF_SWEEP_REVERSE"):
    dirs = dirs[::-1]
for wd in dirs:
    w.set_wind(8.0, float(wd))
    w.step(yaw, out); w.sync()
    w.timing_begin()
    for _ in range(10):
        w.step(yaw, out)
    ms = w.timing_end() / 10
    ts.append(ms)
    k = w.kernel_info()
    print(f"wd {wd:6.1f}  {ms:.3f} ms  {'one-block' if k['one_block_kernel'] else 'slot'} {k['lanes_per_env']}x{k['slots_per_lane']}", flush=True)
ts = np.array(ts)
print(f"# mean {ts.mean():.3f} ms, min {ts.min():.3f}, max {ts.max():.3f}; mean over 240..300: {ts[4:29].mean():.3f}")
