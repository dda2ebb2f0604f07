"""Dev script (GPU box): parity of the HIP step against the C oracle on seeded random inputs + rough timing."""
import json, os, sys, time
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity
from wfcrl_env_amd.backend import WfStep
from oracle import c_oracle

L = json.load(open(os.path.join(ROOT, "wfcrl-env_amd", "environments", "layouts.json")))
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["Turb3_Row1_", "Ablaincourt_", "Turb6_Row2_", "Turb16_Row5_", "Turb_TCRWP_", "HornsRev1_", "HornsRev2_"]
Bp = int(os.environ.get("BP", 512))
rng = np.random.default_rng(7)

def errs(got, ref):
    p = np.abs(got["power"] - ref["power"]) / np.maximum(ref["power"], 1e3)
    return {
        "power_rel_max": float(p.max()), "power_rel_p999": float(np.quantile(p, 0.999)), "power_rel_med": float(np.median(p)),
        "frac_gt_1e-4": float((p > 1e-4).mean()),
        "ws_rel_max": float((np.abs(got["wind_speed"] - ref["wind_speed"]) / ref["wind_speed"]).max()),
        "wd_abs_max": float(np.abs(got["wind_direction"] - ref["wind_direction"]).max()),
        "ti_abs_max": float(np.abs(got["load"][..., 0] - ref["load"][..., 0]).max()),
        "stdu_abs_max": float(np.abs(got["load"][..., 1] - ref["load"][..., 1]).max()),
        "stdv_abs_max": float(np.abs(got["load"][..., 2] - ref["load"][..., 2]).max()),
        "stdw_abs_max": float(np.abs(got["load"][..., 3] - ref["load"][..., 3]).max()),
    }

for name in names:
    l = L[name]; N = l["num_turbines"]; x, y = l["xcoords"], l["ycoords"]
    for mode in ("shared270", "perenv"):
        B = Bp
        yaw = rng.uniform(-40, 40, (B, N)).astype(np.float32)
        if mode == "shared270":
            ws, wd = np.array([8.0]), np.array([270.0])
        else:
            ws = np.clip(8 * rng.weibull(8, B), 3, 28); wd = rng.normal(270, 20, B) % 360
        w = WfStep(x, y, env_batch=B)
        w.set_wind(ws, wd)
        got = w.step(yaw)
        flags = w.risk_flags()
        ref = c_oracle.farm_step_batch(x, y, ws, wd, yaw.astype(np.float64), margin=True)
        e = errs(got, ref)
        sm = parity.summarize(got, ref, flags)
        e.update(class_=parity.classify(sm), flagged_frac=sm["n_flagged"] / sm["n"], flagged_overlap=int(((flags & 1) != 0).sum()),
                 flagged_knee=int(((flags & 2) != 0).sum()), n_mismatch_flagged=sm["n_mismatch_flagged"],
                 worst_unflagged=sm["worst_unflagged"], n_farms=sm["n"])
        print(name, mode, w.kernel_info(), json.dumps(e), flush=True)
        w.close()

# timing on HornsRev1 65536
if os.environ.get("TIME", "1") == "1":
    import torch
    for name, B in (("Ablaincourt_", 65536), ("Turb16_Row5_", 65536), ("Turb_TCRWP_", 65536), ("HornsRev1_", 65536), ("HornsRev2_", 65536)):
        l = L[name]; N = l["num_turbines"]
        w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
        w.set_wind(8.0, 270.0)
        yaw = (torch.rand((B, N), device="cuda") * 80 - 40).float()
        torch.cuda.synchronize()
        out = w.step(yaw); w.sync()
        w.timing_begin()
        K = 5
        for _ in range(K): w.step(yaw, out)
        ms = w.timing_end() / K
        print("TIME", name, N, B, w.kernel_info(), f"{ms:.3f} ms/step", f"{B/ms*1e3:.3e} farm-steps/s", flush=True)
        w.close()
