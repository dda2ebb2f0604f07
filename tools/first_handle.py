"""GPU box: is the first handle of a process slower than later ones at the same batch (buffer placement)?
  python tools/first_handle.py [B]"""
import json, os, sys, torch
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
sys.path.insert(0, os.getcwd())
from wfcrl_env_amd.backend import WfStep
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
L = json.load(open("wfcrl-env_amd/environments/layouts.json"))["HornsRev1_"]
N = 80
yaw = (torch.rand((B, N), device="cuda") * 60 - 30).float()
for rep in range(4):
    w = WfStep(L["xcoords"], L["ycoords"], env_batch=B)
    w.set_wind(8.0, 270.0)
    out = w.step(yaw)
    for _ in range(10):
        w.step(yaw, out)
    w.sync()
    ts = []
    for r in range(3):
        w.timing_begin()
        for _ in range(20):
            w.step(yaw, out)
        ts.append(w.timing_end() / 20)
    print(f"B={B} handle {rep}: {min(ts):.4f} ms (runs {' '.join(f'{t:.4f}' for t in ts)})", flush=True)
    w.close()
