"""GPU box: a wind per farm on the one-slot on-the-fly kernels, alternative builds (WFSTEP_LIB) — ms per step, float32 only.
  python tools/fly_s1_ab.py build/alt/lib_a.so wfcrl-env_amd/libwfstep.so"""
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
for name, B, fam in (("HornsRev1_", 8192, "8"), ("HornsRev1_", 12288, "8"), ("HornsRev1_", 16384, "8"), ("HornsRev1_", 16384, "4"), ("HornsRev1_", 24576, "4"),
                     ("HornsRev2_", 16384, "8"), ("HornsRev1_", 16384, None)):
    l = L[name]; N = l["num_turbines"]
    rng = np.random.default_rng(5)
    yaw = torch.from_numpy(rng.uniform(-30, 30, (B, N)).astype(np.float32)).cuda()
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B, kernel_choice=dict(one_block=fam, calibrate=False) if fam else None)
    w.set_risk_resolve(0)
    w.set_wind(np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360)
    o = w.step(yaw); w.step(yaw, o); w.step(yaw, o); w.sync()
    best = 1e9
    for rep in range(3):
        w.timing_begin()
        for _ in range(6): w.step(yaw, o)
        best = min(best, w.timing_end() / 6)
    i = w.kernel_info()
    print(os.path.basename(os.environ["WFSTEP_LIB"]), name, B, "forced", fam, f"{best:.3f} ms", {k: i[k] for k in ("lanes_per_env", "slots_per_lane", "one_block_kernel", "pair_table", "vgprs", "scratch_bytes")}, flush=True)
    w.close()
'''
for lib in sys.argv[1:]:
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, WFSTEP_LIB=os.path.abspath(lib), WFSTEP_NO_AUTOBUILD="1", WF_RISK_RESOLVE="0"), timeout=400)
