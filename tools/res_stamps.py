"""GPU box: where a wave of the float64 re-solve kernel spends its cycles (a -DWF_RES_STAMP build of wf_resolve.hip:
s_memtime deltas per phase, summed over the farms of the launch).  bash tools/res_stamps.sh"""
import ctypes as C, json, os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wfcrl_env_amd.backend import WfStep
from wfcrl_env_amd import _lib
lib = _lib.load()
lay = json.load(open(os.path.join(ROOT, "wfcrl-env_amd", "environments", "layouts.json")))["HornsRev1_"]
B, N = 65536, 80
rng = np.random.default_rng(1)
import torch
yaw = torch.from_numpy(rng.uniform(-30, 30, (B, N)).astype(np.float32)).cuda()
w = WfStep(lay["xcoords"], lay["ycoords"], env_batch=B)
w.set_wind(np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360)
w.set_risk_resolve(1)
out = w.step(yaw); w.sync()
buf = (C.c_ulonglong * 8)()
lib.wfk_res_stamps(buf, 1)
w.step(yaw, out); w.sync()
lib.wfk_res_stamps(buf, 0)
v = list(buf)
n = max(v[6], 1)
names = ["source_begin", "transverse_pass", "source_finish", "deficit_pass", "farm setup", "outputs"]
tot = sum(v[:6])
print(f"{n} farms, {tot / n:.0f} s_memtime ticks per farm (the counter runs at about the shader clock: 2.46 M ticks for a farm that lives ~1.05 ms)")
for k, nm in enumerate(names):
    print(f"  {nm:16s} {v[k] / n:10.0f} ticks per farm  {100 * v[k] / tot:5.1f} %   ({v[k] / n / N:.1f} per source)")
