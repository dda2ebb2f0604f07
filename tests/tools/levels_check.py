"""GPU box: the LEVEL stages of the four-wave float64 kernel (csrc/wf_resolve.hip: Lvl4Shared) against its sequential stages —
the same farms solved with levels on and off must come out BIT for BIT the same (every sum is taken in source order either
way); and how many sources the levels covered, how many farms failed a level's check and were solved again without.
usage: python tests/tools/levels_check.py [B]"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from wfcrl_env_amd import _lib
from wfcrl_env_amd.backend import WfStep
lib = _lib.load()
L = json.load(open(os.path.join(ROOT, "wfcrl-env_amd", "environments", "layouts.json")))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 192


def stats(reset=True):
    buf = (C.c_ulonglong * 8)()
    lib.wfk_res_level_stats(buf, 1 if reset else 0)
    return list(buf)


bad = 0
for name in ("HornsRev1_", "HornsRev2_", "Turb_TCRWP_", "Turb16_Row5_", "Turb32_Row5_", "Ormonde_", "WMR_", "Ablaincourt_", "Turb6_Row2_"):
    l = L[name]; N = l["num_turbines"]
    for mode in ("270", "263", "300", "per_farm"):
        rng = np.random.default_rng(7)
        yaw = rng.uniform(-40, 40, (B, N)).astype(np.float32)
        outs = []
        for lv in (1, 0):
            lib.wfk_set_resolve_levels(lv)
            w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
            w.set_risk_resolve(2)
            if mode == "per_farm":
                r2 = np.random.default_rng(11)
                w.set_wind(np.clip(8 * r2.weibull(8, B), 3, 28), r2.normal(270, 20, B) % 360)
            else:
                w.set_wind(8.0, float(mode))
            stats()
            o = w.step(yaw); w.sync()
            st = stats()
            outs.append({k: np.array(v) for k, v in o.items()})
            if lv:
                st_on = st
            w.close()
        diff = {k: int((outs[0][k].view(np.uint32) != outs[1][k].view(np.uint32)).sum()) for k in outs[0]}
        nd = sum(diff.values())
        bad += nd
        print(f"{name:14s} N={N:3d} wd {mode:8s}: farms {st_on[0]:4d}, solved again without levels {st_on[1]:3d}, level stages {st_on[2]:6d} "
              f"covering {st_on[3]:6d} of {st_on[3] + st_on[4]:6d} sources ({100.0 * st_on[3] / max(1, st_on[3] + st_on[4]):5.1f} %)"
              f"  | values differing from the sequential solve: {nd}", flush=True)
lib.wfk_set_resolve_levels(1)
print("TOTAL differing values:", bad)
sys.exit(1 if bad else 0)
