"""GPU box: randomised fuzz of the batched env (make(..., env_batch=B) -> VecWindFarmEnv, fused wf_env_step) against B
single-farm envs with the reference's semantics (simple_env / mdp mirror) running on the float64 oracle
(tests/helpers.py): random layout, controls, discrete / continuous actions, load_coef, episode length, reset by seed
(host-side draws in the reference's order) or by options, actions that overshoot the step and trip the actuation
budget.  usage: python tests/tools/fuzz_env.py [n_episodes] [seed]"""
import os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np


def run(n_episodes, seed):
    import torch

    from helpers import OracleFlorisInterface
    from wfcrl_env_amd import environments as envs
    from wfcrl_env_amd.environments.registration import get_case
    from wfcrl_env_amd.simple_env import WindFarmEnv

    rng = np.random.default_rng(seed)
    names = ["Turb3_Row1_", "Turb6_Row2_", "Ablaincourt_", "Turb16_Row5_", "Ormonde_", "Turb_TCRWP_"]
    nbad = 0
    for ep in range(n_episodes):
        name = names[rng.integers(0, len(names))]
        discrete = bool(rng.random() < 0.4)
        lim, step = float(rng.choice([20, 30, 40])), float(rng.choice([2, 4, 5]))
        controls = {"yaw": (-lim, lim, step)}
        B, T = int(rng.integers(1, 9)), int(rng.integers(4, 30))
        load_coef = float(rng.choice([0.0, 0.1, 0.5]))
        gs = None
        if rng.random() < 0.5:  # multi-slot kernel variants only large batches pick by themselves
            N0 = get_case(name, "Floris").num_turbines
            fits = [(g, s) for g in (4, 8, 16, 32) for s in range(1, 7) if g * s >= N0 and (s <= 4 or g == 16)]
            g, s = fits[rng.integers(0, len(fits))]
            gs = f"{g}x{s}"
        kw = dict(controls=dict(controls), max_num_steps=T, continuous_control=not discrete, load_coef=load_coef)
        venv = envs.make(name + "Floris", env_batch=B, kernel_choice=dict(slot=gs) if gs else None, **kw)
        N = venv.num_turbines
        ctx = dict(ep=ep, name=name, B=B, T=T, discrete=discrete, controls=controls, load_coef=load_coef, gs=gs)
        if rng.random() < 0.5:
            sd = int(rng.integers(0, 10000))
            obs = venv.reset(seed=sd)
        else:
            obs = venv.reset(options={"wind_speed": float(rng.uniform(4, 18)), "wind_direction": float(rng.choice([270.0, rng.uniform(0, 360)]))})
        fw = obs["freewind_measurements"].cpu().numpy()
        refs = []
        for b in range(B):
            e = WindFarmEnv(interface=OracleFlorisInterface, farm_case=get_case(name, "Floris").clone(), **{**kw, "controls": dict(controls)})
            o = e.reset(options={"wind_speed": fw[b, 0], "wind_direction": fw[b, 1]})
            if np.abs(obs["wind_speed"][b].cpu().numpy() - o["wind_speed"]).max() > 3e-5 * 28:
                nbad += 1
                print("BAD reset obs", ctx, flush=True)
            refs.append(e)
        for t in range(T - 1):
            if discrete:
                a = rng.integers(0, 3, (B, N)).astype(np.float32)
            else:
                a = rng.uniform(-1.6 * step, 1.6 * step, (B, N)).astype(np.float32)
            obs, rew, term, trunc, info = venv.step({"yaw": torch.from_numpy(a).cuda()})
            for b in range(B):
                o, r, te, tr, i = refs[b].step({"yaw": a[b].copy()})
                why = None
                if not np.array_equal(obs["yaw"][b].cpu().numpy(), o["yaw"]):
                    why = "yaw"
                elif bool(trunc[b]) != tr or bool(term[b]) != te:
                    why = "flags"
                elif abs(float(rew[b]) - r[0]) > 1e-4 * abs(r[0]) + 1e-7:
                    why = f"reward {float(rew[b])} vs {r[0]}"
                elif not np.allclose(info["power"][b].cpu().numpy(), i["power"], rtol=2e-3, atol=1e-6):
                    why = "power"
                elif np.abs(obs["wind_direction"][b].cpu().numpy() - o["wind_direction"]).max() > 1e-3:
                    why = "wind_direction"
                if why:
                    nbad += 1
                    print("BAD", why, dict(ctx, t=t, b=b), flush=True)
                    break
            if why:
                break
        venv.close()
    print(f"env fuzz: {n_episodes} episodes, violations: {nbad}")
    return nbad


if __name__ == "__main__":
    a = sys.argv
    sys.exit(1 if run(int(a[1]) if len(a) > 1 else 40, int(a[2]) if len(a) > 2 else 1) else 0)
