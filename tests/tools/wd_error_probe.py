"""GPU box: float32 error of the ON-THE-FLY kernels on a large aligned grid (fuzz seed 611 case 516 / seed 622 case 506:
8 x 14 grid at 5 D x 4 D, exact x' ties at 360 deg) over hub height, shear, direction and veer — worst error over the
batch against the float64 oracle, register-slot kernel on the fly, one-block kernel on the fly, and the table path."""
import os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from oracle import c_oracle
from oracle.floris_gch_numpy import ModelParams
from wfcrl_env_amd.backend import WfStep
import parity

D = 100.5
gx, gy = np.meshgrid(np.arange(8) * 5 * D, np.arange(14) * 4 * D, indexing="ij")
x, y = gx.ravel(), gy.ravel()
N, B = x.size, 64
rng = np.random.default_rng(0)
yaw = rng.uniform(-35, 35, (B, N)).astype(np.float32)
ws = rng.uniform(5, 20, B)
args = sys.argv[1:]
for hh in (0.56, 0.714, 0.9):
    for shear in (0.0, 0.12):
        for wd0 in (360.0, 357.0):
            for veer in (0.0, -6.0):
                model = dict(rotor_diameter=D, hub_height=hh * D, shear=shear, veer=veer)
                mp = ModelParams(D=D, HH=hh * D, shear=shear, veer=veer)
                wd = np.full(B, wd0)
                ref = c_oracle.farm_step_batch(x, y, ws, wd, yaw.astype(np.float64), mp, margin=True)
                row = []
                for label, choice, per_farm in (("slot fly", dict(one_block=False, pair_table=False), False), ("ll fly", dict(one_block="4x2"), True),
                                                ("table", dict(one_block=False), False)):
                    w = WfStep(x, y, env_batch=B, model=model, kernel_choice=choice)
                    if per_farm:
                        import torch
                        w.set_wind(torch.from_numpy(ws).cuda(), torch.from_numpy(wd).cuda())  # device arrays: a direction per farm
                    else:
                        w.set_wind(ws, wd)
                    got = w.step(yaw)
                    got = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in got.items()}
                    e = parity.errors(got, ref)
                    fl = w.risk_flags()
                    k = w.kernel_info()
                    row.append(f"{label} [{'ll' if k['one_block_kernel'] else 'slot'} {k['lanes_per_env']}x{k['slots_per_lane']} t{k['pair_table']}] wd {e['wd'][fl == 0].max():.1e} P {e['power'][fl == 0].max():.1e} ws {e['ws'][fl == 0].max():.1e}")
                    w.close()
                print(f"HH {hh} D shear {shear} wd {wd0} veer {veer}: " + " | ".join(row), flush=True)
