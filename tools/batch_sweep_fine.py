"""GPU box: throughput over the env batch in steps of 4096: what the rounds model picks, what every
forced family delivers, and where the pick stands against the best family at that batch and against the running best
(the "envelope": the highest farm-steps/s of any family at any batch up to this one).
  python tools/batch_sweep_fine.py [layout ...] > gpurun_out/r03_batch_sweep_fine.txt      (SWEEP_BS=36864,69632: those batches only;
with SWEEP_BS the envelope is that of the listed batches)"""
import json, os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wfcrl_env_amd.backend import WfStep

L = json.load(open(os.path.join(ROOT, "wfcrl-env_amd", "environments", "layouts.json")))
FAMS = [("slot", dict(one_block=False)), ("16x1", dict(one_block="16")), ("8x1", dict(one_block="8")), ("4x2", dict(one_block="4x2")),
        ("4x1", dict(one_block="4")), ("2x2", dict(one_block="2x2"))]


def time_it(lay, B, choice):
    N = lay["num_turbines"]
    w = WfStep(lay["xcoords"], lay["ycoords"], env_batch=B, kernel_choice=choice)
    w.set_wind(8.0, 270.0)
    yaw = (torch.rand((B, N), device="cuda") * 60 - 30).float()
    out = w.step(yaw)
    for _ in range(8):  # (the first handle timed at a batch used to run 3-5 % cold: the pick was measured before the forced families)
        w.step(yaw, out)
    w.sync()
    best = 1e9
    for r in range(3):
        w.timing_begin()
        for _ in range(6):
            w.step(yaw, out)
        best = min(best, w.timing_end() / 6)
    k = w.kernel_info()
    w.close()
    fam = "slot" if not k["one_block_kernel"] else f"{k['lanes_per_env']}x{k['slots_per_lane']}"
    if k.get("mixed_main_farms"):  # (round 5: whole rounds on the family + the remainder on the slot kernel)
        fam += f"+slot@{k['mixed_main_farms']}"
    return best, fam


if __name__ == "__main__":
    for name in (sys.argv[1:] or ["HornsRev1_", "HornsRev2_"]):
        lay = L[name]
        print(f"# {name} N={lay['num_turbines']}  (ms per step; farm-steps/s of the pick; pick / best family at this batch; pick / envelope)")
        env_best = 0.0
        worst_fam, worst_env = 1.0, 1.0
        step = int(os.environ.get("SWEEP_STEP", 4096))
        Bs = [int(v) for v in os.environ["SWEEP_BS"].split(",")] if os.environ.get("SWEEP_BS") else range(step, 131072 + 1, step)
        for B in Bs:
            time_it(lay, B, None)  # (the first handle after a change of batch runs ~5 % slow whatever it is: tools/pick_vs_forced.py)
            ts = {f: time_it(lay, B, c)[0] for f, c in FAMS}
            t_pick, fam_pick = time_it(lay, B, None)  # (round 4: the handle calibrates itself on its third step — wf_kernel_choice::calibrate)
            t_best = min(ts.values())
            thr = B / t_pick * 1e3
            env_best = max(env_best, B / min(t_best, t_pick) * 1e3)
            r_fam, r_env = min(t_best / t_pick, 1.0), thr / env_best
            worst_fam, worst_env = min(worst_fam, r_fam), min(worst_env, r_env)
            print(f"B={B:7d} pick {fam_pick:16s} {t_pick:.3f} ms {thr:.3e}  " + " ".join(f"{f}={t:.3f}" for f, t in ts.items())
                  + f"  pick/best {r_fam:.3f}  pick/envelope {r_env:.3f}", flush=True)
        print(f"# {name}: worst pick / best family {worst_fam:.3f}, worst pick / envelope {worst_env:.3f}")
