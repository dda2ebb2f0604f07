"""Episode traces produced by the REFERENCE'S OWN Python (interface.py, mdp.py, simple_env.py, multiagent_env.py,
rewards.py, wrappers.py, registration.py imported from /root/reference in the build container) replayed through this
repo's envs.  tests/golden/make_episode_traces.py is the generator; the physics under the reference's code was the
float64 NumPy oracle ("reference Python over shimmed containers, oracle physics" — not FLORIS-pinned).

CPU: the host-side envs (`make(...)` on an oracle-backed float64 interface) — yaw, action mutation, truncation, dtypes
bit for bit; rewards / observations / powers / loads at 1e-12 (the C and NumPy oracles differ by summation order).
GPU: `make(...)` on the HIP interface and the batched `VecWindFarmEnv` / `VecAECWindFarmEnv` at the float32 tolerances.
SURVEY §8 a9, f1, f3; reference wfcrl/mdp.py:233-319, simple_env.py:58-96, multiagent_env.py:159-254."""
import json
import os
import warnings

import numpy as np
import pytest

from conftest import ROOT
from helpers import OracleFlorisInterface64


@pytest.fixture(scope="module")
def traces():
    z = np.load(os.path.join(ROOT, "tests", "golden", "episode_traces.npz"))
    header = json.loads(bytes(z["header_json"]).decode())
    assert "oracle physics" in header["label"] or "oracle" in header["label"]
    return z, header["scenarios"]


def _names(kind):
    z = np.load(os.path.join(ROOT, "tests", "golden", "episode_traces.npz"))
    sc = json.loads(bytes(z["header_json"]).decode())["scenarios"]
    return [n for n, m in sc.items() if m["scenario"]["kind"] == kind]


def _shaper(spec):
    from wfcrl_env_amd import rewards

    name, arg = spec
    return rewards.DoNothingReward() if name == "DoNothingReward" else getattr(rewards, name)(arg)


def _make(registration, sc, series_path=None, **extra):
    kw = dict(max_num_steps=sc["max_num_steps"], load_coef=sc["load_coef"], continuous_control=sc["continuous"],
              reward_shaper=_shaper(sc["shaper"]), **extra)
    if series_path is not None:
        kw["wind_time_series"] = series_path
    if sc.get("global_np_seed") is not None:
        np.random.seed(sc["global_np_seed"])  # the series' random start draws from the global RNG (interface.py:518)
    controls = {k: tuple(v) for k, v in sc["controls"].items()} if sc["controls"] else ["yaw"]
    return registration.make(sc["env_id"], controls=controls, **kw)


def _series_csv(z, tmp_path):
    p = tmp_path / "wind.csv"
    p.write_text("speed,direction\n" + "\n".join(f"{a!r},{b!r}" for a, b in z["series"].tolist()) + "\n")
    return str(p)


def _rel(a, b, floor=1e-30):
    return float(np.max(np.abs(np.asarray(a, float) - b) / np.maximum(np.abs(b), floor)))


def reward_tol(sc, want, rel):
    """|d reward| allowed for a relative error `rel` of the UNSHAPED reward: a percentage shaper (rewards.py:24-46) turns
    r into (r - ref) / ref, which divides the error by the small difference it reports."""
    kind = sc["shaper"][0]
    if kind == "DoNothingReward":
        return rel * max(abs(want), 1e-3)
    return rel * abs(want + 1.0) * (2.0 if kind == "StepPercentage" else 1.0) + 1e-15


def replay_central(env, z, name, meta, tol, yaw_exact=True):
    sc = meta["scenario"]
    g = lambda k: z[f"{name}/{k}"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        obs = env.reset(seed=sc.get("reset_seed"), options=sc.get("options"))
    assert list(obs.keys()) == meta["obs_keys"]
    worst = {}

    def check_obs(obs, t, dtypes):
        for k in meta["obs_keys"]:
            want = g("obs_" + k)[t]
            assert str(np.asarray(obs[k]).dtype) == dtypes[k], (name, t, k)
            if k == "yaw" and yaw_exact:
                assert np.array_equal(obs[k], want), (name, t)
            elif k == "wind_direction":
                worst["wd"] = max(worst.get("wd", 0), float(np.abs(np.asarray(obs[k], float) - want).max()))
            else:
                worst[k] = max(worst.get(k, 0), _rel(obs[k], want))

    check_obs(obs, 0, meta["obs_dtypes_reset"])
    T = meta["num_steps"]
    for t in range(T):
        a = g("actions_in")[t].copy()
        obs, r, term, trunc, info = env.step({"yaw": a})
        assert np.array_equal(a, g("actions_after")[t]), (name, t)  # the budget gate zeroes the caller's array in place
        assert bool(trunc) == bool(g("truncated")[t]) and bool(term) == bool(g("terminated")[t]), (name, t)
        assert isinstance(r, np.ndarray) and r.shape == (1,)
        check_obs(obs, t + 1, meta["obs_dtypes_step"])
        want_r = g("reward")[t]
        worst["reward"] = max(worst.get("reward", 0), abs(float(r[0]) - want_r) / reward_tol(sc, want_r, 1.0))
        worst["power"] = max(worst.get("power", 0), float(np.max(np.abs(info["power"] - g("power")[t]) / np.maximum(g("power")[t], 1e-3))))
        worst["load"] = max(worst.get("load", 0), float(np.abs(info["load"] - g("load")[t]).max()))
    assert bool(g("truncated")[-1])
    for k, v in worst.items():
        assert v <= tol[k], (name, k, v)
    return worst


def replay_aec(env, z, name, meta, tol):
    sc = meta["scenario"]
    g = lambda k: z[f"{name}/{k}"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        env.reset(seed=sc.get("reset_seed"), options=sc.get("options"))
    i = 0
    worst = dict(reward=0.0, ws=0.0, wd=0.0, power=0.0, load=0.0)
    for agent in env.agent_iter():
        assert agent == f"turbine_{int(g('agent')[i]) + 1}", (name, i)
        o, r, term, trunc, info = env.last()
        assert list(o.keys()) == meta["obs_keys"]
        assert float(o["yaw"]) == g("obs_yaw")[i], (name, i)
        worst["ws"] = max(worst["ws"], abs(float(o["wind_speed"]) - g("obs_wind_speed")[i]) / g("obs_wind_speed")[i])
        worst["wd"] = max(worst["wd"], abs(float(o["wind_direction"]) - g("obs_wind_direction")[i]))
        want_r = g("reward")[i]
        worst["reward"] = max(worst["reward"], abs(float(np.ravel(r)[0]) - want_r) / reward_tol(sc, want_r, 1.0))
        assert bool(term) == bool(g("terminated")[i]) and bool(trunc) == bool(g("truncated")[i]), (name, i)
        wp = g("info_power")[i]
        assert ("power" in info) == bool(np.isfinite(wp)), (name, i)
        if "power" in info:
            worst["power"] = max(worst["power"], abs(float(info["power"]) - wp) / max(wp, 1e-3))
            worst["load"] = max(worst["load"], float(np.abs(np.asarray(info["load"], float) - g("info_load")[i]).max()))
        if term or trunc:
            assert np.isnan(g("action_in")[i])
            env.step(None)
        else:
            a = np.array([g("action_in")[i]], np.float32)
            env.step({"yaw": a})
            assert float(a[0]) == g("action_after")[i], (name, i)  # per-agent budget: zeroed in place
        i += 1
    assert i == meta["num_calls"]
    assert {a: len(h["reward"]) for a, h in env.history.items()} == meta["history_lengths"]
    for k, v in worst.items():
        assert v <= tol[k], (name, k, v)
    return worst


TOL64 = dict(reward=1e-12, power=1e-12, load=1e-12, wd=1e-10, ws=1e-12, wind_speed=1e-12, freewind_measurements=1e-14, yaw=0)
# float32 device surface under the host arithmetic (tests/parity.py: power 1e-4, speed 5e-5, direction 3e-4 deg)
TOL32 = dict(reward=3e-5, power=1e-4, load=1e-4, wd=3e-4, ws=5e-5, wind_speed=5e-5, freewind_measurements=1e-14, yaw=0)


@pytest.fixture()
def oracle_registration(monkeypatch):
    from wfcrl_env_amd.environments import registration

    monkeypatch.setattr(registration, "HipFlorisInterface", OracleFlorisInterface64)
    return registration


def test_every_scenario_kind_is_covered(traces):
    z, scs = traces
    kinds = {m["scenario"]["kind"] for m in scs.values()}
    assert kinds == {"central", "aec"}
    assert any(m["gate_fired"] for m in scs.values()) and any(not m["gate_fired"] for m in scs.values())
    assert any(not m["scenario"]["continuous"] for m in scs.values())
    assert {m["scenario"]["shaper"][0] for m in scs.values()} == {"DoNothingReward", "StepPercentage", "ReferencePercentage"}


@pytest.mark.parametrize("name", _names("central"))
def test_host_env_reproduces_reference_trace(oracle_registration, traces, name, tmp_path):
    z, scs = traces
    meta = scs[name]
    series = _series_csv(z, tmp_path) if name.endswith("series") else None
    env = _make(oracle_registration, meta["scenario"], series)
    replay_central(env, z, name, meta, TOL64)
    assert len(env.history["reward"]) == meta["num_steps"]  # LogWrapper (wrappers.py:61-88)


@pytest.mark.parametrize("name", _names("aec"))
def test_host_aec_env_reproduces_reference_trace(oracle_registration, traces, name):
    z, scs = traces
    meta = scs[name]
    env = _make(oracle_registration, meta["scenario"])
    replay_aec(env, z, name, meta, TOL64)


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", _names("central"))
def test_hip_env_reproduces_reference_trace(traces, name, tmp_path):
    """`make(...)` as the reference's user calls it, HIP interface (B = 1) underneath."""
    from wfcrl_env_amd.environments import registration

    z, scs = traces
    meta = scs[name]
    series = _series_csv(z, tmp_path) if name.endswith("series") else None
    env = _make(registration, meta["scenario"], series)
    replay_central(env, z, name, meta, TOL32)


@pytest.mark.gpu
@pytest.mark.parametrize("name", _names("aec"))
def test_hip_aec_env_reproduces_reference_trace(traces, name):
    from wfcrl_env_amd.environments import registration

    z, scs = traces
    meta = scs[name]
    env = _make(registration, meta["scenario"])
    replay_aec(env, z, name, meta, TOL32)


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in _names("central") if not n.endswith("series")])
def test_vec_env_reproduces_reference_trace(traces, name):
    """The fused device-resident env (wf_env_*): B copies of the traced farm, the trace's actions for every one."""
    import torch

    from wfcrl_env_amd.environments import registration

    z, scs = traces
    meta = scs[name]
    sc = meta["scenario"]
    B = 3
    g = lambda k: z[f"{name}/{k}"]
    fw = g("obs_freewind_measurements")[0]
    venv = _make(registration, sc, env_batch=B)
    obs = venv.reset(options={"wind_speed": np.full(B, fw[0]), "wind_direction": np.full(B, fw[1])})
    for k, tol in (("wind_speed", 5e-5),):
        want = g("obs_" + k)[0]
        assert np.abs(obs[k].cpu().numpy() / want - 1).max() <= tol
    assert np.array_equal(obs["yaw"].cpu().numpy(), np.broadcast_to(g("obs_yaw")[0], (B, meta["num_turbines"])))
    for t in range(meta["num_steps"]):
        a = np.broadcast_to(g("actions_in")[t], (B, meta["num_turbines"])).copy()
        obs, rew, term, trunc, info = venv.step({"yaw": torch.from_numpy(a).cuda()})
        for b in range(B):
            assert np.array_equal(obs["yaw"][b].cpu().numpy(), g("obs_yaw")[t + 1]), (name, t)  # incl. the gate's effect
            assert bool(trunc[b]) == bool(g("truncated")[t]) and not bool(term[b])
            want_r = g("reward")[t]
            assert abs(float(rew[b]) - want_r) <= reward_tol(sc, want_r, 3e-5), (name, t, float(rew[b]), want_r)
            assert np.allclose(info["power"][b].cpu().numpy(), g("power")[t], rtol=1e-4, atol=1e-7)
            assert np.abs(info["load"][b].cpu().numpy() - g("load")[t]).max() < 1e-4
            assert np.abs(obs["wind_speed"][b].cpu().numpy() / g("obs_wind_speed")[t + 1] - 1).max() <= 5e-5
            assert np.abs(obs["wind_direction"][b].cpu().numpy() - g("obs_wind_direction")[t + 1]).max() <= 3e-4
    venv.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", _names("aec"))
def test_vec_aec_env_reproduces_reference_trace(traces, name):
    """The batched AEC env (vec_adapters.VecAECWindFarmEnv): one launch per agent cycle, B farms."""
    import torch

    from wfcrl_env_amd.environments import registration

    z, scs = traces
    meta = scs[name]
    sc = meta["scenario"]
    B = 2
    g = lambda k: z[f"{name}/{k}"]
    env = _make(registration, sc, env_batch=B)
    opts = sc.get("options")
    if opts is None:  # the trace's seeded draw (mdp.py:235-253) handed over explicitly: a batch draws B winds from the seed
        rng = np.random.default_rng(sc["reset_seed"])
        ws = float(np.clip(8 * rng.weibull(8), 3, 28))
        wd = float(np.clip(rng.normal(270, 20) % 360, 0, 360))
        opts = {"wind_speed": ws, "wind_direction": wd}
    env.reset(options={k: np.full(B, v) for k, v in opts.items()})
    i = 0
    for agent in env.agent_iter():
        assert agent == f"turbine_{int(g('agent')[i]) + 1}", (name, i)
        o, r, term, trunc, info = env.last()

        def f(v):  # (B, 1) view of a batched value, a per-farm tensor or a plain Python scalar alike
            v = np.asarray(v.cpu() if hasattr(v, "cpu") else v, float)
            return np.broadcast_to(v.reshape(-1, 1) if v.ndim else v, (B, 1))

        assert np.all(f(o["yaw"]) == g("obs_yaw")[i]), (name, i)
        assert np.abs(f(o["wind_speed"]) / g("obs_wind_speed")[i] - 1).max() <= 5e-5
        assert np.abs(f(o["wind_direction"]) - g("obs_wind_direction")[i]).max() <= 3e-4
        want_r = g("reward")[i]
        assert np.abs(f(r) - want_r).max() <= reward_tol(sc, want_r, 3e-5), (name, i, f(r), want_r)
        assert bool(np.all(f(term) == g("terminated")[i])) and bool(np.all(f(trunc) == g("truncated")[i])), (name, i)
        if np.all(f(trunc) != 0) or np.all(f(term) != 0):
            env.step(None)
        else:
            a = torch.full((B, 1), float(g("action_in")[i]), dtype=torch.float32, device="cuda")
            env.step({"yaw": a})
        i += 1
    assert i == meta["num_calls"]
