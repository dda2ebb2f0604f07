"""GPU box: a wind per farm (reset distribution, mdp.py:237-258) on the on-the-fly one-block kernel, family by family:
which lane-group width serves which layout / batch.   python tools/fly_family_ab.py > gpurun_out/r04_fly_family_ab.txt"""
import json, os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wfcrl_env_amd.backend import WfStep

L = json.load(open(os.path.join(ROOT, "wfcrl-env_amd", "environments", "layouts.json")))


def time_it(lay, B, fam):
    N = lay["num_turbines"]
    w = WfStep(lay["xcoords"], lay["ycoords"], env_batch=B, kernel_choice=None if fam is None else dict(one_block=fam))
    w.sample_wind(1234)
    yaw = (torch.rand((B, N), device="cuda") * 60 - 30).float()
    out = w.step(yaw)
    for _ in range(4):
        w.step(yaw, out)
    w.sync()
    best = 1e9
    for r in range(3):
        w.timing_begin()
        for _ in range(4):
            w.step(yaw, out)
        best = min(best, w.timing_end() / 4)
    k = w.kernel_info()
    w.close()
    return best, f"{k['lanes_per_env']}x{k['slots_per_lane']}" + ("" if k["one_block_kernel"] else "-slot")


for name, Bs in () if __name__ != "__main__" else (("HornsRev1_", (16384, 65536)), ("HornsRev2_", (65536, 131072)), ("Ormonde_", (65536,)), ("WMR_", (65536,)), ("Turb32_Row5_", (65536,))):
    lay = L[name]
    for B in Bs:
        row = []
        for fam in ("2x2", "4x2", "8"):
            if lay["num_turbines"] <= 8:
                continue
            t, k = time_it(lay, B, fam)
            row.append(f"{fam}->{k} {t:.3f} ms")
        print(f"{name} N={lay['num_turbines']} B={B}: " + "   ".join(row), flush=True)
