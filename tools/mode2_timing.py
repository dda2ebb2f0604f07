import json, os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from wfcrl_env_amd.backend import WfStep
L = json.load(open("wfcrl-env_amd/environments/layouts.json"))
for name, B in (("HornsRev1_", 65536), ("HornsRev2_", 32768), ("Ormonde_", 65536), ("Ablaincourt_", 65536)):
    l = L[name]; N = l["num_turbines"]
    rng = np.random.default_rng(1)
    yaw = torch.from_numpy(rng.uniform(-30, 30, (B, N)).astype(np.float32)).cuda()
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    w.set_wind(8.0, 270.0); w.set_risk_resolve(2)
    o = w.step(yaw); w.sync()
    w.timing_begin()
    for _ in range(3): w.step(yaw, o)
    print(name, B, "mode 2: %.2f ms per step" % (w.timing_end() / 3), flush=True)
    w.close()
