"""GPU box: time alternative builds of libwfstep (WFSTEP_LIB=path) on HornsRev1 65536, interleaved rounds."""
import json, os, subprocess, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
libs = sys.argv[1:]
code = r'''
import os, sys, json, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"] if "GRAFT_REPO_ROOT" in os.environ else ".")
from wfcrl_env_amd import _lib
from pathlib import Path
_lib.LIB_PATH = Path(os.environ["WFSTEP_LIB"])
from wfcrl_env_amd.backend import WfStep
L = json.load(open("wfcrl-env_amd/environments/layouts.json"))
name = os.environ.get("LAYOUT", "HornsRev1_"); B = int(os.environ.get("B", 65536))
l = L[name]; N = l["num_turbines"]
w = WfStep(l["xcoords"], l["ycoords"], env_batch=B); w.set_wind(8.0, 270.0)
g = torch.Generator(device="cuda").manual_seed(1)
yaw = (torch.rand((B, N), device="cuda", generator=g) * 80 - 40).float()
out = w.step(yaw); w.sync()
ts = []
for r in range(3):
    w.timing_begin()
    for _ in range(10): w.step(yaw, out)
    ts.append(w.timing_end() / 10)
print(os.path.basename(os.environ["WFSTEP_LIB"]), name, "ms/step", ["%.3f" % t for t in ts], w.kernel_info()["vgprs"], w.kernel_info()["scratch_bytes"])
'''
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, WFSTEP_LIB=os.path.abspath(lib), WFSTEP_NO_AUTOBUILD="1")
        subprocess.run([sys.executable, "-c", code], env=env)
