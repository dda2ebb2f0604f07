"""GPU box: what the LEVEL stages of the four-wave float64 kernel buy on the configurations VERDICT r5 item 1 names — ms per
step float32-only, strict with every source a stage of its own, strict with levels; farms re-solved; share of the sources
inside level stages; farms that failed a level's check.   python tools/levels_ab.py [small]"""
import ctypes as C, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wfcrl_env_amd import _lib
if os.environ.get("WFSTEP_LIB"):  # an alternative build (tools/build_alt_res.sh)
    from pathlib import Path
    _lib.LIB_PATH = Path(os.environ["WFSTEP_LIB"])
    print("library:", os.environ["WFSTEP_LIB"], flush=True)
from wfcrl_env_amd.backend import WfStep
lib = _lib.load()
L = json.load(open(os.path.join(ROOT, "wfcrl-env_amd", "environments", "layouts.json")))
tc = L["Turb_TCRWP_"]
L["Turb16_TCRWP_"] = {"num_turbines": 16, "xcoords": tc["xcoords"][:16], "ycoords": tc["ycoords"][:16]}
small = len(sys.argv) > 1


def stats():
    buf = (C.c_ulonglong * 8)()
    lib.wfk_res_level_stats(buf, 1)
    return list(buf)


CASES = [("cfg2 Ablaincourt x 4096, 270", "Ablaincourt_", 4096, "shared", 270.0),
         ("cfg2 per farm", "Ablaincourt_", 4096, "farm", None),
         ("cfg3 Turb16_TCRWP x 16384, 270 (0 flagged)", "Turb16_TCRWP_", 16384, "shared", 270.0),
         ("cfg3 per farm", "Turb16_TCRWP_", 16384, "farm", None),
         ("HornsRev1 x 8192, 270 (8-GPU shard)", "HornsRev1_", 8192, "shared", 270.0),
         ("HornsRev2 x 16384, 270 (8-GPU shard)", "HornsRev2_", 16384, "shared", 270.0),
         ("cfg4 HornsRev1 x 65536 per farm", "HornsRev1_", 65536, "farm", None),
         ("cfg5 HornsRev2 x 131072 sweep", "HornsRev2_", 131072, "shared", 270.0 + 30.0 * np.sin(2 * np.pi * 37 / 200.0)),
         ("cfg5 per farm", "HornsRev2_", 131072, "farm", None)]
for label, name, B, mode, wd in CASES:
    if small and B > 20000:
        continue
    l = L[name]; N = l["num_turbines"]
    rng = np.random.default_rng(1234)
    yaw = torch.from_numpy(rng.uniform(-30, 30, (B, N)).astype(np.float32)).cuda()
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    if mode == "farm":
        w.set_wind(np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360)
    else:
        w.set_wind(8.0, wd)
    res = []
    for rmode, lv in ((0, 1), (1, 0), (1, 1)):
        lib.wfk_set_resolve_levels(lv)
        w.set_risk_resolve(rmode)
        o = w.step(yaw); w.step(yaw, o); w.sync()
        stats()
        best = 1e9
        for rep in range(3):
            w.timing_begin()
            for _ in range(10): w.step(yaw, o)
            best = min(best, w.timing_end() / 10)
        st = stats()
        res.append((best, st))
    n = w.resolve_stats()["n_resolved"]
    st = res[2][1]
    print(f"{label:46s}: f32 {res[0][0]:.3f} | strict, no levels {res[1][0]:.3f} (+{res[1][0] - res[0][0]:.3f}) | strict, levels {res[2][0]:.3f} "
          f"(+{res[2][0] - res[0][0]:.3f})  {n} farms re-solved, {100.0 * st[3] / max(1, st[3] + st[4]):.0f} % of sources in levels, "
          f"{st[1]} of {st[0]} farm solves repeated without levels", flush=True)
    w.close()
lib.wfk_set_resolve_levels(1)
