import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from wfcrl_env_amd.backend import WfStep
L = json.load(open("wfcrl-env_amd/environments/layouts.json"))["HornsRev1_"]
B, N = 65536, 80
w = WfStep(L["xcoords"], L["ycoords"], env_batch=B)
w.set_wind(8.0, 270.0)
w.env_config(load_coef=0.1)
w.env_reset()
rng = np.random.default_rng(0)
acts = [torch.from_numpy(rng.uniform(-5, 5, (B, N)).astype(np.float32)).cuda() for _ in range(4)]
yaw = torch.from_numpy(rng.uniform(-30, 30, (B, N)).astype(np.float32)).cuda()
full = ("reward", "yaw", "power", "wind_speed", "wind_direction", "load")
light = ("reward", "yaw", "wind_speed", "wind_direction")
def t(fn, n=30):
    for _ in range(5): fn()
    w.sync(); best = 1e9
    for _ in range(3):
        w.timing_begin()
        for i in range(n): fn(i)
        best = min(best, w.timing_end() / n)
    return best
o = w.step(yaw)
print("plain step                      %.4f" % t(lambda i=0: w.step(yaw, o)))
for want, nm in ((full, "full"), (light, "light"), (("reward",), "reward only"), (("reward", "power", "wind_speed", "wind_direction", "load"), "full w/o yaw")):
    ob = w.env_step(acts[0], want=want)
    print("env step, action, %-12s  %.4f" % (nm, t(lambda i=0: w.env_step(acts[i % 4], want=want, out=ob))))
    print("env step, no action, %-9s  %.4f" % (nm, t(lambda i=0: w.env_step(None, want=want, out=ob))))
w.close()
