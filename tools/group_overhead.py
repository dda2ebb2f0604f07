"""GPU box: what a grouped launch (direction groups) costs over the plain shared-wind launch — separated from the cost of
the directions themselves: (a) T series rows all inside a sector where no turbine has a split-TI upstream source
(262..276 deg on HornsRev1: tools/direction_sweep.py), (b) rows spread over 240..300 deg, (c) the plain launch at the
slowest direction of (b).   python tools/group_overhead.py [T] [B]"""
import json, os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from wfcrl_env_amd.backend import WfStep
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
L = json.load(open("wfcrl-env_amd/environments/layouts.json"))["HornsRev1_"]
N = 80
w = WfStep(L["xcoords"], L["ycoords"], env_batch=B)
g = torch.Generator(device="cuda").manual_seed(1)
yaw = (torch.rand((B, N), device="cuda", generator=g) * 60 - 30).float()
rng = np.random.default_rng(0)


def timed(label):
    out = w.step(yaw)
    for _ in range(10):
        w.step(yaw, out)
    w.sync()
    w.timing_begin()
    for _ in range(10):
        w.step(yaw, out)
    ms = w.timing_end() / 10
    k = w.kernel_info()
    print(f"{label:64s} {ms:.3f} ms/step  {k['lanes_per_env']}x{k['slots_per_lane']} groups={k['direction_groups']} blocks={k['grid_blocks']}", flush=True)
    return ms


print(f"# HornsRev1 x {B}, {T} series rows")
w.set_wind(8.0, 270.0); timed("plain, 270 deg")
for lo, hi in ((262.0, 276.0), (240.0, 300.0)):
    wd = rng.uniform(lo, hi, T)
    w.set_wind_series(np.stack([np.full(T, 8.0), wd], axis=1)); timed(f"series rows in {lo:.0f}..{hi:.0f} deg (grouped)")
    worst = 0.0
    for d in wd:
        w.set_wind(8.0, float(d))
        out = w.step(yaw); w.sync()
        w.timing_begin()
        for _ in range(3):
            w.step(yaw, out)
        worst = max(worst, w.timing_end() / 3)
    print(f"{'   slowest of these rows as a plain launch':64s} {worst:.3f} ms/step")
