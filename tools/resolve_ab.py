"""GPU box: the float64 re-solve of alternative builds (tools/build_alt_res.sh) on the configurations VERDICT r4 item 2 names:
ms per step with the re-solve off / on and the farms re-solved.   python tools/resolve_ab.py build/alt/lib_*.so"""
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
out_line = [os.path.basename(os.environ["WFSTEP_LIB"])]
for name, B, per_farm, sweep in (("Ablaincourt_", 4096, False, False), ("Ablaincourt_", 4096, True, False), ("Turb16_TCRWP_", 16384, True, False),
                                 ("HornsRev1_", 65536, True, False), ("HornsRev2_", 131072, False, True)):
    if os.environ.get("RES_AB_SMALL") and B > 20000: continue
    l = L[name]; N = l["num_turbines"]
    rng = np.random.default_rng(1234)
    yaw = torch.from_numpy(rng.uniform(-30, 30, (B, N)).astype(np.float32)).cuda()
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    if per_farm: w.set_wind(np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360)
    elif sweep: w.set_wind(8.0, 270.0 + 30.0 * np.sin(2 * np.pi * 37 / 200.0))
    else: w.set_wind(8.0, 270.0)
    r = []
    for mode in (0, 1):
        w.set_risk_resolve(mode)
        o = w.step(yaw); w.step(yaw, o); w.sync()
        best = 1e9
        for rep in range(3):
            w.timing_begin()
            for _ in range(10): w.step(yaw, o)
            best = min(best, w.timing_end() / 10)
        r.append(best)
    n = w.resolve_stats()["n_resolved"]
    out_line.append(f"{name.rstrip('_')}x{B}{'/farm' if per_farm else ('/sweep' if sweep else '')}: {r[0]:.3f}->{r[1]:.3f} (+{r[1]-r[0]:.3f}, {n})")
    w.close()
    print(out_line[0], out_line[-1], flush=True)
'''
for lib in sys.argv[1:]:
    try:  # (a variant that hangs must not take the call's whole limit with it)
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, WFSTEP_LIB=os.path.abspath(lib), WFSTEP_NO_AUTOBUILD="1"), timeout=int(os.environ.get("RES_AB_TIMEOUT", 150)))
    except subprocess.TimeoutExpired:
        print(os.path.basename(lib), "TIMEOUT", flush=True)
