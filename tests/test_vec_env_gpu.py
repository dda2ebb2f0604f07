"""GPU: the fused env step (wf_env_*, VecWindFarmEnv) and the B = 1 HipFlorisInterface through make(),
against the host-side mirror of the reference env running on the float64 oracle (tests/helpers.py)."""
import numpy as np
import pytest

from helpers import OracleFlorisInterface

pytestmark = pytest.mark.gpu


def _host_env(name, layouts, **kw):
    """Reference-semantics env (simple_env/mdp mirror) on the oracle, one farm."""
    from wfcrl_env_amd.environments.registration import get_case, get_default_control
    from wfcrl_env_amd.simple_env import WindFarmEnv

    case = get_case(name, "Floris").clone()
    controls = kw.pop("controls", get_default_control(["yaw"]))
    return WindFarmEnv(interface=OracleFlorisInterface, farm_case=case, controls=controls, **kw)


def test_make_single_env_on_gpu_matches_reference_kat_and_oracle_env(layouts, kat1):
    from wfcrl_env_amd import environments as envs

    env = envs.make("Ablaincourt_Floris", max_num_steps=70)
    opts = {"wind_speed": kat1["wind_speed_free"], "wind_direction": kat1["wind_direction_free"]}
    obs = env.reset(options=opts)
    assert np.abs(obs["wind_speed"] / np.array(kat1["wind_speed"]) - 1).max() < 2e-6
    assert np.abs(obs["wind_direction"] - np.array(kat1["wind_direction"])).max() < 1e-4
    ref = _host_env("Ablaincourt_", layouts, max_num_steps=70)
    ref.reset(options=opts)
    rng = np.random.default_rng(0)
    for _ in range(12):
        a = rng.uniform(-5, 5, 7)
        o1, r1, t1, tr1, i1 = env.step({"yaw": a.copy()})
        o2, r2, t2, tr2, i2 = ref.step({"yaw": a.copy()})
        assert np.array_equal(o1["yaw"], o2["yaw"]) and tr1 == tr2
        assert np.allclose(i1["power"], i2["power"], rtol=1e-4)
        assert abs(r1[0] - r2[0]) < 2e-5 * abs(r2[0])
        assert np.abs(o1["wind_direction"] - o2["wind_direction"]).max() < 2e-4


def test_aec_example_loop_on_gpu():
    """The reference's examples/example_floris.py loop, verbatim usage."""
    from wfcrl_env_amd import environments as envs
    from wfcrl_env_amd.rewards import StepPercentage

    env = envs.make("Dec_Ablaincourt_Floris", max_num_steps=20, reward_shaper=StepPercentage(), load_coef=1)
    env.reset()
    r = {agent: 0 for agent in env.possible_agents}
    done = {agent: False for agent in env.possible_agents}
    num_steps = {agent: 0 for agent in env.possible_agents}
    for agent in env.agent_iter():
        observation, reward, termination, truncation, info = env.last()
        done[agent] = done[agent] or termination or truncation
        r[agent] += reward
        if done[agent]:
            action = None
        else:
            action = {"yaw": np.array([15.0])} if (agent == "turbine_1" and num_steps[agent] == 5) else {"yaw": np.array([0])}
            num_steps[agent] += 1
        env.step(action)
    assert set(num_steps.values()) == {19}
    assert np.isfinite(float(np.ravel(r["turbine_1"])[0]))


@pytest.mark.parametrize("name,discrete", [("Turb6_Row2_", False), ("Ablaincourt_", False), ("Turb16_Row5_", True)])
def test_vec_env_matches_B_reference_envs(layouts, name, discrete):
    import torch

    from wfcrl_env_amd import environments as envs

    B, T = 12, 26
    N = layouts[name]["num_turbines"]
    controls = {"yaw": (-30, 30, 4)} if discrete else {"yaw": (-40, 40, 5)}
    venv = envs.make(name + "Floris", controls=dict(controls), env_batch=B, max_num_steps=T,
                     continuous_control=not discrete, load_coef=0.25)
    obs = venv.reset(seed=77)
    fw = obs["freewind_measurements"].cpu().numpy()
    # seeded batch draw order: weibull x B, then normal x B
    rng = np.random.default_rng(77)
    ws = np.clip(8 * rng.weibull(8, B), 3, 28)
    wd = np.clip(rng.normal(270, 20, B) % 360, 0, 360)
    assert np.allclose(fw[:, 0], ws) and np.allclose(fw[:, 1], wd)
    refs = []
    for b in range(B):
        e = _host_env(name, layouts, controls=dict(controls), max_num_steps=T, continuous_control=not discrete,
                      load_coef=0.25)
        o = e.reset(options={"wind_speed": fw[b, 0], "wind_direction": fw[b, 1]})
        assert np.abs(obs["wind_speed"][b].cpu().numpy() - o["wind_speed"]).max() < 2e-5 * 28
        refs.append(e)
    arng = np.random.default_rng(5)
    n_steps = 0
    while True:
        if discrete:
            a = arng.integers(0, 3, (B, N)).astype(np.float32)
        else:
            a = arng.uniform(-7, 7, (B, N)).astype(np.float32)  # beyond the +-5 step: exercises the clip
        obs, rew, term, trunc, info = venv.step({"yaw": torch.from_numpy(a).cuda()})
        n_steps += 1
        for b in range(B):
            o, r, t, tr, i = refs[b].step({"yaw": a[b].copy()})
            assert np.array_equal(obs["yaw"][b].cpu().numpy(), o["yaw"]), (n_steps, b)  # float32 MDP arithmetic: exact
            assert bool(trunc[b]) == tr and not bool(term[b])
            assert abs(float(rew[b]) - r[0]) <= 3e-5 * abs(r[0]), (n_steps, b, float(rew[b]), r[0])
            assert np.allclose(info["power"][b].cpu().numpy(), i["power"], rtol=2e-4, atol=1e-6)
            assert np.abs(info["load"][b].cpu().numpy() - i["load"]).max() < 1e-4
        if bool(trunc[0]):
            break
    assert n_steps == T - 1  # one solve consumed by reset (Appendix C8)
    # the budget gate must have triggered for somebody during the episode (otherwise the test is vacuous)
    acc = np.stack([e.mdp.get_accumulated_actions()["yaw"] for e in refs])
    assert (acc / 0.3 / n_steps / 60 >= 0.1).any()
    venv.close()


def test_vec_env_light_step_and_numpy_mode(layouts):
    from wfcrl_env_amd import environments as envs

    venv = envs.make("Turb3_Row1_Floris", env_batch=5, max_num_steps=6, return_torch=False)
    obs = venv.reset(seed=1)
    assert isinstance(obs["yaw"], np.ndarray) and obs["wind_speed"].shape == (5, 3)
    a = np.full((5, 3), 2.0, np.float32)
    o1, r1, trunc = venv.step_light(a)
    assert np.allclose(o1["yaw"], 2.0) and r1.shape == (5,) and trunc is False
    o2, r2, term, tr, info = venv.step(a)
    assert np.allclose(o2["yaw"], 4.0) and info["power"].shape == (5, 3) and info["power"].max() < 6.0  # MW
    venv.close()
