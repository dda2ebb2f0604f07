"""Cost of the float64 re-solve (wf_set_risk_resolve) per step: HIP-event time of wf_step with the re-solve off / on,
for a shared wind (no farm flagged) and the reference's reset distribution per farm (about 2 % flagged).
  python tools/resolve_cost.py [layout] [B]"""
import sys
import os
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wfcrl_env_amd.backend import WfStep  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "HornsRev1_"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
lay = json.load(open(os.path.join(ROOT, "wfcrl-env_amd", "environments", "layouts.json")))[name]
N = lay["num_turbines"]
rng = np.random.default_rng(1)
yaw = torch.from_numpy(rng.uniform(-30, 30, (B, N)).astype(np.float32)).cuda()
w = WfStep(lay["xcoords"], lay["ycoords"], env_batch=B)
for label, ws, wd in (("shared 8 m/s 270 deg", 8.0, 270.0),
                      ("a wind per farm (mdp.py:237-258)", np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360)):
    w.set_wind(ws, wd)
    for mode in (0, 1):
        w.set_risk_resolve(mode)
        out = w.step(yaw)
        for _ in range(3):  # (the handle times its kernel families on its third table-path step: keep that out of the figures)
            w.step(yaw, out)
        w.sync()
        w.timing_begin()
        for _ in range(10):
            w.step(yaw, out)
        ms = w.timing_end() / 10
        st = w.resolve_stats() if mode else {"raw_flags": w.risk_flags(), "n_resolved": 0}
        print(f"{name} B={B} {label}: resolve {'on ' if mode else 'off'} {ms:.3f} ms/step, "
              f"flagged {int((st['raw_flags'] != 0).sum())}, re-solved {st['n_resolved']}", flush=True)
w.close()
