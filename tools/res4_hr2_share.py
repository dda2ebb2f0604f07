"""GPU box: the HornsRev2 x 16384 share at 270 deg (39 farms flagged), strict, 30 steps — a short run to profile the four-wave
float64 kernel on (rocprofv3 --kernel-trace / --pmc).   python3 tools/res4_hr2_share.py [steps]"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wfcrl_env_amd.backend import WfStep
L = json.load(open(os.path.join(ROOT, "wfcrl-env_amd", "environments", "layouts.json")))
l = L["HornsRev2_"]; N = l["num_turbines"]; B = 16384
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(1234)
yaw = torch.from_numpy(rng.uniform(-30, 30, (B, N)).astype(np.float32)).cuda()
w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
w.set_wind(8.0, 270.0)
w.set_risk_resolve(1)
o = w.step(yaw); w.sync()
for _ in range(steps): w.step(yaw, o)
w.sync()
print("farms re-solved:", w.resolve_stats()["n_resolved"])
w.close()
