"""GPU box: are the risk flags of a farm sticky from step to step?  (a float64 re-solve started
speculatively on LAST step's flagged farms only pays if most of this step's flagged farms were flagged before.)
HornsRev1 / HornsRev2 under the reference's reset distribution (a wind per farm, fixed over the episode), yaw as bench.py
moves it (random walk, dyaw ~ U(-5, 5) clipped to +-40): per step, the flagged farms and how many of them were flagged at
the previous step, by flag type.
usage: python tests/tools/flag_stickiness.py [layout] [B] [steps]"""
import json, os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
sys.path.insert(0, os.getcwd())
import numpy as np
import torch


def main():
    from wfcrl_env_amd.backend import WfStep

    name = sys.argv[1] if len(sys.argv) > 1 else "HornsRev1_"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    lay = json.load(open(os.path.join(os.getcwd(), "wfcrl-env_amd", "environments", "layouts.json")))[name]
    N = len(lay["xcoords"])
    rng = np.random.default_rng(11)
    w = WfStep(lay["xcoords"], lay["ycoords"], env_batch=B)
    w.set_wind(np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360)
    g = torch.Generator(device="cuda").manual_seed(3)
    yaw = torch.zeros((B, N), device="cuda")
    prev = None
    tot = dict(flagged=0, sticky=0, ov=0, ov_sticky=0, tab=0, tab_sticky=0)
    for t in range(steps + 4):
        yaw = (yaw + (torch.rand((B, N), device="cuda", generator=g) * 10 - 5)).clamp_(-40, 40)
        w.step(yaw)
        f = w.risk_flags()
        if prev is not None and t >= 4:
            fl, pv = f != 0, prev != 0
            ov, tab = (f & 1) != 0, (f & 6) != 0
            tot["flagged"] += int(fl.sum()); tot["sticky"] += int((fl & pv).sum())
            tot["ov"] += int(ov.sum()); tot["ov_sticky"] += int((ov & ((prev & 1) != 0)).sum())
            tot["tab"] += int(tab.sum()); tot["tab_sticky"] += int((tab & ((prev & 6) != 0)).sum())
        prev = f
    print(f"{name} B={B}, {steps} steps: flagged per step {tot['flagged'] / steps:.0f} ({100.0 * tot['flagged'] / steps / B:.2f} %), "
          f"of which flagged at the previous step too {100.0 * tot['sticky'] / max(tot['flagged'], 1):.1f} %")
    print(f"   overlap flag: {tot['ov'] / steps:.0f} per step, sticky {100.0 * tot['ov_sticky'] / max(tot['ov'], 1):.1f} %;   "
          f"knee / ramp flags: {tot['tab'] / steps:.0f} per step, sticky {100.0 * tot['tab_sticky'] / max(tot['tab'], 1):.1f} %")
    print(f"   newly flagged per step (the serial float64 tail a speculative re-solve would leave): {(tot['flagged'] - tot['sticky']) / steps:.0f} farms")
    w.close()


if __name__ == "__main__":
    main()
