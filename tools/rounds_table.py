"""GPU box: the measured table behind wf_dispatch.hip's rounds model (kRoundsMs) — ms of ONE round of every kernel family
at 1, 2, 3 blocks per CU, for farms of 32, 48, 64, 80, 91 turbines (the first N of HornsRev2, wind 8 m/s / 263 deg:
no x' ties), on the pair-table path.  A round = (CUs x blocks per CU) blocks = that many x farms-per-block farms.
  python tools/rounds_table.py > gpurun_out/r03_rounds_table.txt"""
import json, os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wfcrl_env_amd.backend import WfStep

L = json.load(open(os.path.join(ROOT, "wfcrl-env_amd", "environments", "layouts.json")))["HornsRev2_"]
NS = (32, 48, 64, 80, 91)
FAMILIES = [("slot", dict(one_block=False)), ("8x1", dict(one_block="8")), ("4x2", dict(one_block="4x2")),
            ("4x1", dict(one_block="4")), ("2x2", dict(one_block="2x2")), ("16x1", dict(one_block="16"))]
if os.environ.get("WF_ROUNDS_ONLY"):  # e.g. WF_ROUNDS_ONLY=16x1: one family's rows
    FAMILIES = [f for f in FAMILIES if f[0] in os.environ["WF_ROUNDS_ONLY"].split(",")]
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
print(f"# device: {torch.cuda.get_device_name(0)}, {n_cu} CUs")
table = {}
SLOT = {32: "8x4", 48: "16x3", 64: "16x4", 80: "16x5", 91: "16x6"}  # pick_variant's throughput choice per N
for fam, choice in FAMILIES:
    for N in NS:
        x, y = L["xcoords"][:N], L["ycoords"][:N]
        if fam == "slot":
            choice = dict(one_block=False, slot=SLOT[N])
        row = []
        for per_cu in (1, 2, 3):
            w = WfStep(x, y, env_batch=64, kernel_choice=choice)
            w.set_wind(8.0, 263.0)
            info = w.kernel_info()
            fpb = info["envs_per_block"]
            w.close()
            B = n_cu * per_cu * fpb
            w = WfStep(x, y, env_batch=B, kernel_choice=choice)
            w.set_wind(8.0, 263.0)
            info = w.kernel_info()
            if fam != "slot" and not info["one_block_kernel"]:
                row.append(0.0)
                w.close()
                continue
            yaw = (torch.rand((B, N), device="cuda") * 60 - 30).float()
            out = w.step(yaw); w.sync()
            best = 1e9
            for r in range(3):
                w.timing_begin()
                for _ in range(10):
                    w.step(yaw, out)
                best = min(best, w.timing_end() / 10)
            row.append(best)
            print(f"{fam:5s} N={N:3d} blocks/CU={per_cu} B={B:6d} G={info['lanes_per_env']} S={info['slots_per_lane']} fpb={fpb} "
                  f"vgprs={info['vgprs']} {best:.4f} ms  {B / best * 1e3:.3e} farm-steps/s", flush=True)
            w.close()
        table[(fam, N)] = row
print("# C initialiser (wf_dispatch.hip: kRoundsMs[family][N][blocks per CU - 1])")
for fam, _ in FAMILIES:
    print("    {" + ", ".join("{" + ", ".join(f"{v:.3f}" for v in table[(fam, N)]) + "}" for N in NS) + "},  // " + fam)
