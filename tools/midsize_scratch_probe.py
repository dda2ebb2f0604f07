"""GPU box: mid-size farms (30-48 turbines: register-slot variants with three slots per lane, which spill at three waves per
SIMD) — kernel, private segment, ms per step float32-only and in the default mode (re-solve on: a kernel without scratch
follows every step kernel)."""
import json, os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from wfcrl_env_amd.backend import WfStep
L = json.load(open("wfcrl-env_amd/environments/layouts.json"))
h2 = L["HornsRev2_"]
L["HornsRev2_first48"] = {"num_turbines": 48, "xcoords": h2["xcoords"][:48], "ycoords": h2["ycoords"][:48]}
for name in ("Ormonde_", "Turb_TCRWP_", "WMR_", "HornsRev2_first48"):
    l = L[name]; N = l["num_turbines"]
    for B in (2048, 8192, 16384):
        rng = np.random.default_rng(1)
        yaw = torch.from_numpy(rng.uniform(-30, 30, (B, N)).astype(np.float32)).cuda()
        w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
        w.set_wind(8.0, 263.0)
        r = []
        for mode in (0, 1):
            w.set_risk_resolve(mode)
            o = w.step(yaw); w.step(yaw, o); w.step(yaw, o); w.sync()
            best = 1e9
            for rep in range(3):
                w.timing_begin()
                for _ in range(10): w.step(yaw, o)
                best = min(best, w.timing_end() / 10)
            r.append(best)
        i = w.kernel_info()
        print(f"{name:18s} N={N} B={B:6d} {i['lanes_per_env']}x{i['slots_per_lane']} one_block={i['one_block_kernel']} vgprs={i['vgprs']} scratch={i['scratch_bytes']}  float32 {r[0]:.3f} ms  default {r[1]:.3f} ms (+{(r[1]-r[0])*1e3:.0f} us, {w.resolve_stats()['n_resolved']} re-solved)", flush=True)
        w.close()
