"""GPU box: two builds (WFSTEP_LIB) over the shapes a priority change touches — ms per step, float32 only."""
import os, subprocess, sys
code = r'''
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from wfcrl_env_amd import _lib
from pathlib import Path
_lib.LIB_PATH = Path(os.environ["WFSTEP_LIB"])
from wfcrl_env_amd.backend import WfStep
L = json.load(open("wfcrl-env_amd/environments/layouts.json"))
tc = L["Turb_TCRWP_"]
L["Turb16_TCRWP_"] = {"num_turbines": 16, "xcoords": tc["xcoords"][:16], "ycoords": tc["ycoords"][:16]}
for name, B, per_farm in (("HornsRev1_", 65536, False), ("HornsRev1_", 8192, False), ("HornsRev1_", 16384, False), ("HornsRev1_", 32768, False), ("HornsRev2_", 16384, False), ("HornsRev2_", 131072, False),
                          ("HornsRev1_", 65536, True), ("HornsRev1_", 8192, True), ("Turb16_TCRWP_", 16384, False), ("Ablaincourt_", 4096, False), ("WMR_", 8192, False)):
    l = L[name]; N = l["num_turbines"]
    rng = np.random.default_rng(5)
    yaw = torch.from_numpy(rng.uniform(-30, 30, (B, N)).astype(np.float32)).cuda()
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    w.set_risk_resolve(0)
    if per_farm: w.set_wind(np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360)
    else: w.set_wind(8.0, 270.0)
    o = w.step(yaw); w.step(yaw, o); w.step(yaw, o); w.sync()
    best = 1e9
    for rep in range(3):
        w.timing_begin()
        for _ in range(8): w.step(yaw, o)
        best = min(best, w.timing_end() / 8)
    i = w.kernel_info()
    print(os.path.basename(os.environ["WFSTEP_LIB"]), f"{name:14s} B={B:6d} {'per-farm' if per_farm else 'shared  '} {best:.4f} ms  {i['lanes_per_env']}x{i['slots_per_lane']} one_block={i['one_block_kernel']}", flush=True)
    w.close()
'''
for lib in sys.argv[1:]:
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, WFSTEP_LIB=os.path.abspath(lib), WFSTEP_NO_AUTOBUILD="1", WF_RISK_RESOLVE="0"), timeout=600)
