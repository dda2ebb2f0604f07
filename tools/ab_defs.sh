#!/bin/bash
# GPU box: A/B of wf_kernels_ll.hip built with different -D flags: bash tools/ab_defs.sh "-DWF_LL_OCC2=3" "-DWF_LL_OCC2=2"
# HornsRev1 x 65536 shared wind on the 2x2 / 4x2 / 4x1 / 8x1 families, and a wind per farm (4x2 on the fly).
cd $GRAFT_REPO_ROOT/wfcrl-env_amd/csrc
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-variable -fno-fast-math -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp"
for v in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS $v -c -o wf_kernels_ll.o wf_kernels_ll.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libwfstep.so wf_kernels_1.o wf_kernels_2.o wf_kernels_ll.o wf_resolve.o wf_abi.o wf_model.o wf_dispatch.o wf_groups.o wf_wind_abi.o wf_env_abi.o
  echo "== $v"
  (cd ../.. && python - <<'PY'
import json, numpy as np, torch
from wfcrl_env_amd.backend import WfStep
L = json.load(open("wfcrl-env_amd/environments/layouts.json"))
for name in ("HornsRev1_", "HornsRev2_"):
    l = L[name]; N = l["num_turbines"]; B = 65536
    yaw = (torch.rand((B, N), device="cuda") * 60 - 30).float()
    for fam in ("2x2", "4x2", "4", "8"):
        w = WfStep(l["xcoords"], l["ycoords"], env_batch=B, kernel_choice=dict(one_block=fam))
        w.set_wind(8.0, 270.0)
        out = w.step(yaw); w.sync()
        best = 1e9
        for r in range(3):
            w.timing_begin()
            for _ in range(10): w.step(yaw, out)
            best = min(best, w.timing_end() / 10)
        print(f"{name} shared wind {fam:4s} {best:.4f} ms", flush=True)
        w.close()
    rng = np.random.default_rng(1)
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    w.set_wind(np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360)
    out = w.step(yaw); w.sync()
    best = 1e9
    for r in range(3):
        w.timing_begin()
        for _ in range(5): w.step(yaw, out)
        best = min(best, w.timing_end() / 5)
    print(f"{name} a wind per farm {w.kernel_info()['lanes_per_env']}x{w.kernel_info()['slots_per_lane']} {best:.4f} ms", flush=True)
    w.close()
PY
)
done
