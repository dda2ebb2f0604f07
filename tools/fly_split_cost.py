"""GPU box: the on-the-fly kernel (a wind per farm) with and without split-TI sources in the batch: directions inside a
sector where HornsRev1 has none (262..276 deg) against the reference's reset distribution N(270, 20).
  python tools/fly_split_cost.py [B]"""
import json, os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from wfcrl_env_amd.backend import WfStep
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
L = json.load(open("wfcrl-env_amd/environments/layouts.json"))["HornsRev1_"]
N = 80
w = WfStep(L["xcoords"], L["ycoords"], env_batch=B)
g = torch.Generator(device="cuda").manual_seed(1)
yaw = (torch.rand((B, N), device="cuda", generator=g) * 60 - 30).float()
rng = np.random.default_rng(0)
ws = np.clip(8 * rng.weibull(8, B), 3, 28)
for label, wd in (("directions U(264, 274): no split-TI sources", rng.uniform(264, 274, B)),
                  ("directions U(279, 281): split-TI sources in every farm", rng.uniform(279, 281, B)),
                  ("directions N(270, 20) (mdp.py:237-258)", rng.normal(270, 20, B) % 360)):
    w.set_wind(ws, wd)
    out = w.step(yaw)
    for _ in range(5):
        w.step(yaw, out)
    w.sync()
    w.timing_begin()
    for _ in range(5):
        w.step(yaw, out)
    ms = w.timing_end() / 5
    k = w.kernel_info()
    print(f"HornsRev1 x {B}, a wind per farm, {label}: {ms:.3f} ms/step [{k['lanes_per_env']}x{k['slots_per_lane']} table {k['pair_table']}]", flush=True)
