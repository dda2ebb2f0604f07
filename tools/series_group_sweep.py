import json, os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from wfcrl_env_amd.backend import WfStep
L = json.load(open("wfcrl-env_amd/environments/layouts.json"))["HornsRev1_"]
B, N = 65536, 80
w = WfStep(L["xcoords"], L["ycoords"], env_batch=B)
g = torch.Generator(device="cuda").manual_seed(1)
yaw = (torch.rand((B, N), device="cuda", generator=g) * 60 - 30).float()
rng = np.random.default_rng(0)
def timed(label, K=10):
    out = w.step(yaw); w.sync()
    best = 1e9
    for r in range(3):
        w.timing_begin()
        for _ in range(K): w.step(yaw, out)
        best = min(best, w.timing_end() / K)
    k = w.kernel_info()
    print(f"{label:40s} {best:.3f} ms groups={k['direction_groups']} blocks={k['grid_blocks']}", flush=True)
w.set_wind(8.0, 270.0); timed("shared")
for T in (1, 2, 8, 32, 128):
    series = np.stack([rng.uniform(6, 12, T), rng.uniform(240, 300, T)], axis=1)
    w.set_wind_series(series); timed(f"series T={T}")
# same direction in all rows (tables identical but separate)
series = np.stack([rng.uniform(6, 12, 64), np.full(64, 270.0)], axis=1)
w.set_wind_series(series); timed("series T=64 all 270")
# contiguous farm ids per group (first half row 0, second half row 1) against interleaved ones
series = np.stack([rng.uniform(6, 12, 2), rng.uniform(240, 300, 2)], axis=1)
st = np.zeros(B, np.int32); st[B // 2:] = 1
w.set_wind_series(series, start=st); timed("series T=2, contiguous halves")
st = (np.arange(B) % 2).astype(np.int32)
w.set_wind_series(series, start=st); timed("series T=2, alternating farms")
st = ((np.arange(B) // 64) % 2).astype(np.int32)
w.set_wind_series(series, start=st); timed("series T=2, alternating runs of 64")
st = rng.integers(0, 2, B).astype(np.int32)
w.set_wind_series(series, start=st); timed("series T=2, random starts from the host")
st = np.sort(st)
w.set_wind_series(series, start=st); timed("series T=2, the same starts sorted")
w.set_wind_series(series); timed("series T=2, device-drawn starts")
