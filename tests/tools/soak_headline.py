"""GPU box: soak of the HEADLINE launch: >= 1e5 launches of wf_step on BASELINE configs[3] (HornsRev1 x 65536,
ws 8, wd 270) with the same yaw, every launch's power compared bit for bit with the first launch's, all four outputs every 64th
launch.  The hot-record path of wf_step_ll_kernel reads by hand-issued LDS-DMA what the same wave has just stored to global
memory (csrc/wf_kernels_ll.hip: the source log) — an ordering the compiler's wait-count bookkeeping cannot see; a soak is the
cheap evidence beside the bit-identity tests.   python tests/tools/soak_headline.py [launches]"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from wfcrl_env_amd.backend import WfStep
L = json.load(open(os.path.join(ROOT, "wfcrl-env_amd", "environments", "layouts.json")))["HornsRev1_"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
B, N = 65536, 80
rng = np.random.default_rng(2024)
yaw = torch.from_numpy(rng.uniform(-30, 30, (B, N)).astype(np.float32)).cuda()
w = WfStep(L["xcoords"], L["ycoords"], env_batch=B)
w.set_wind(8.0, 270.0)
ref = {k: v.clone() for k, v in w.step(yaw).items()}
w.sync()
info = w.kernel_info()
bufs = [w.step(yaw), w.step(yaw)]
bufs = [{k: v for k, v in b.items()} for b in bufs]
bad = torch.zeros((), dtype=torch.int64, device="cuda")
t0 = time.perf_counter()
for i in range(n):
    o = w.step(yaw, bufs[i & 1])
    bad += (o["power"] != ref["power"]).any().to(torch.int64)
    if i % 64 == 0:
        for k in ("wind_speed", "wind_direction", "load"):
            bad += (o[k] != ref[k]).any().to(torch.int64)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
flags = int(w.risk_flags().any())
print(f"soak: {n} launches of wf_step on HornsRev1 x {B} (kernel {info['lanes_per_env']}x{info['slots_per_lane']}, one-block {info['one_block_kernel']}, "
      f"pair table {info['pair_table']}, float64 re-solve of flagged farms on, the handle's default): launches whose outputs differed from the first launch's: {int(bad)}; "
      f"{dt:.1f} s, {dt / n * 1e3:.4f} ms per launch including the comparison; flags raised: {flags}")
w.close()
sys.exit(1 if int(bad) else 0)
