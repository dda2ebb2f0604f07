"""GPU box: soak of the four-wave float64 kernel's LEVEL stages with helper waves (round 6: csrc/wf_resolve.hip).  Inside a level
stage the waves hand work to each other through LDS counters and two flags instead of block barriers (the chain waits for a
count of finished member-chunk passes, the deficit passes for the chain's flag) — both waits are bounded and send the farm to the
sequential solve if they ever ran out.  This soak is the evidence that they do not, and that the result does not depend on
which wave drew which pass: n launches of (a) HornsRev2 x 16384 at 270 deg, strict mode (39 farms flagged, wide launches),
(b) HornsRev2 x 480 and HornsRev1 x 700 with every farm in float64 (wide / narrow launch), every launch's four outputs compared bit
for bit with the first launch's; no farm may have been solved a second time without levels.
python tests/tools/soak_resolve.py [launches]"""
import ctypes as C, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from wfcrl_env_amd import _lib
from wfcrl_env_amd.backend import WfStep
lib = _lib.load()
L = json.load(open(os.path.join(ROOT, "wfcrl-env_amd", "environments", "layouts.json")))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000


def stats():
    buf = (C.c_ulonglong * 8)()
    lib.wfk_res_level_stats(buf, 1)
    return list(buf)


bad_total = 0
for label, name, B, mode, share in (("strict, 39 flagged of 16384", "HornsRev2_", 16384, 1, 1.0),
                                    ("every farm in float64, wide launch", "HornsRev2_", 480, 2, 0.1),
                                    ("every farm in float64, narrow launch", "HornsRev1_", 700, 2, 0.1)):
    l = L[name]; N = l["num_turbines"]
    rng = np.random.default_rng(99)
    yaw = torch.from_numpy(rng.uniform(-30, 30, (B, N)).astype(np.float32)).cuda()
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    if mode == 2:
        w.set_wind(np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360)
    else:
        w.set_wind(8.0, 270.0)
    w.set_risk_resolve(mode)
    bufs = [None, None]
    o = w.step(yaw); w.sync()
    first = {k: v.clone() for k, v in o.items()}
    stats()
    k_launches = max(1, int(n * share))
    diff = torch.zeros((), dtype=torch.int64, device="cuda")
    t0 = time.time()
    for it in range(k_launches):
        o = w.step(yaw, o)
        for k in first:
            diff += (o[k].view(torch.int32) != first[k].view(torch.int32)).sum()
    w.sync()
    st = stats()
    nd = int(diff.item())
    bad_total += nd + st[1]
    print(f"resolve soak, {name} x {B}, {label}: {k_launches} launches in {time.time() - t0:.0f} s, {st[0]} farm solves ({st[5]} with helper waves), "
          f"{st[2]} level stages, solved again without levels: {st[1]}, values differing from the first launch: {nd}", flush=True)
    w.close()
print("resolve soak: violations:", bad_total)
sys.exit(1 if bad_total else 0)
