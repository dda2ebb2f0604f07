import json, os, subprocess, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
code = r'''
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
from wfcrl_env_amd.backend import WfStep
L = json.load(open("wfcrl-env_amd/environments/layouts.json"))[os.environ.get("LAYOUT", "HornsRev1_")]
N = L["num_turbines"]
for B in (1024, 2048, 4096, 6144, 8192, 12288, 16384):
    w = WfStep(L["xcoords"], L["ycoords"], env_batch=B); w.set_wind(8.0, 270.0)
    yaw = (torch.rand((B, N), device="cuda") * 60 - 30).float()
    out = w.step(yaw); w.sync()
    res = []
    for mode in (0, 1):  # the float32 kernel on its own, then the handle's default: flagged farms re-solved in float64 behind it
        w.set_risk_resolve(mode)
        w.step(yaw, out); w.sync()
        best = 1e9
        for r in range(3):
            w.timing_begin()
            for _ in range(20): w.step(yaw, out)
            best = min(best, w.timing_end() / 20)
        res.append(best)
    best = res[0]
    k = w.kernel_info()
    print(f"B={B:7d} {best:.3f} ms {B / best * 1e3:.3e} farm-steps/s  (default mode, re-solve on: {res[1]:.3f} ms)  one_block={k['one_block_kernel']} G={k['lanes_per_env']} S={k['slots_per_lane']} vgprs={k['vgprs']} scratch={k['scratch_bytes']}", flush=True)
    w.close()
'''
for label, env in (("WF_LL=0", {"WF_LL": "0"}), ("WF_LL_G=16", {"WF_LL_G": "16"}), ("WF_LL_G=8", {"WF_LL_G": "8"}), ("WF_LL=0 WF_KERNEL_GS=32x3", {"WF_LL": "0", "WF_KERNEL_GS": "32x3"})):
    print("#", label, flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env))
