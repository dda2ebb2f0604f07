"""GPU: the float64 re-solve (include/wfstep.h: wf_set_risk_resolve; csrc/wf_resolve.hip).

With mode 1 the parity contract is UNCONDITIONAL: every farm of the batch — also the ones the float32 kernels flag — is
inside tests/parity.py's TOL (north_star: per-turbine power within 1e-4 of the float64 path; reference
wfcrl/interface.py:564 evaluates in float64, wfcrl/mdp.py:237-258 gives every env its own wind).  The checker is the CPU
oracle throughout; the device float64 kernel is never compared with itself.
"""
import os
import zlib

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _float32_unless_asked(monkeypatch):
    """The tests of this file hold the FLOAT32 kernels to their flag contract (tests/parity.py) and switch the float64
    re-solve on themselves where they test it: new handles start with it off here (WF_RISK_RESOLVE seeds wf_create; the
    library's own default — on — is what test_default_mode_is_the_reference_contract checks)."""
    monkeypatch.setenv("WF_RISK_RESOLVE", "0")


def _oracle(x, y, ws, wd, yaw, mp=None):
    from oracle import c_oracle

    return c_oracle.farm_step_batch(x, y, ws, wd, np.asarray(yaw, dtype=np.float64), mp, margin=True)


def _wind(rng, B, mode):
    if mode == "shared":
        return np.array([8.0]), np.array([270.0])
    return np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360


CASES = [("Turb3_Row1_", 300), ("Turb6_Row2_", 300), ("Ablaincourt_", 1000), ("Turb16_Row5_", 500), ("Turb32_Row5_", 200),
         ("Turb_TCRWP_", 200), ("Ormonde_", 150), ("WMR_", 150), ("HornsRev1_", 160), ("HornsRev2_", 130)]


@pytest.mark.parametrize("name,B", CASES)
@pytest.mark.parametrize("mode", ["shared", "per_env"])
def test_float64_kernel_is_the_oracle(layouts, name, B, mode):
    """mode 2: every farm solved by the float64 kernel.  Against the CPU oracle what is left is the float32 rounding of
    the outputs — including the exact x' ties of the grid layouts at 270 deg (transverse velocities across ties)."""
    import parity
    from wfcrl_env_amd.backend import WfStep

    l = layouts[name]
    N = l["num_turbines"]
    rng = np.random.default_rng(zlib.crc32(f"f64/{name}/{mode}".encode()))
    yaw = rng.uniform(-40, 40, (B, N)).astype(np.float32)
    ws, wd = _wind(rng, B, mode)
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    w.set_risk_resolve(2)
    w.set_wind(ws, wd)
    out = w.step(yaw)
    st = w.resolve_stats()
    assert st["n_resolved"] == B and not w.risk_flags().any()
    parity.check_strict(out, _oracle(l["xcoords"], l["ycoords"], ws, wd, yaw), parity.TOL_F64)
    w.close()


@pytest.mark.parametrize("name,B", CASES)
@pytest.mark.parametrize("mode", ["shared", "per_env"])
def test_resolve_makes_every_farm_strict(layouts, name, B, mode):
    """mode 1 on the cases of test_hip_parity.test_parity_random_yaw_and_wind: no FLAGGED_BOUND, no flagged fraction —
    every farm strict; exactly the flagged farms were re-solved; the unflagged ones keep their float32 bits."""
    import parity
    from wfcrl_env_amd.backend import WfStep

    l = layouts[name]
    N = l["num_turbines"]
    rng = np.random.default_rng(zlib.crc32(f"{name}/{mode}".encode()))
    yaw = rng.uniform(-40, 40, (B, N)).astype(np.float32)
    ws, wd = _wind(rng, B, mode)
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    w.set_wind(ws, wd)
    plain = {k: v.copy() for k, v in w.step(yaw).items()}
    raw = w.risk_flags()
    w.set_risk_resolve(1)
    out = w.step(yaw)
    st = w.resolve_stats()
    assert np.array_equal(st["raw_flags"], raw) and st["n_resolved"] == int((raw != 0).sum())
    assert not w.risk_flags().any()
    for k in out:
        assert np.array_equal(out[k][raw == 0], plain[k][raw == 0]), k
    parity.check_strict(out, _oracle(l["xcoords"], l["ycoords"], ws, wd, yaw))
    w.close()


@pytest.mark.parametrize("name", ["HornsRev1_", "HornsRev2_"])
def test_a_wind_per_farm_4096_farms_all_strict(layouts, name):
    """VERDICT r2 'done when': HornsRev1 / HornsRev2 with the reference's reset distribution per farm (mdp.py:237-258),
    4096 farms: 0 farms outside TOL with the re-solve on; the flagged fraction is the ~2 % the plain path exempts."""
    import parity
    from wfcrl_env_amd.backend import WfStep

    l = layouts[name]
    N, B = l["num_turbines"], 4096
    rng = np.random.default_rng(zlib.crc32(f"4096/{name}".encode()))
    yaw = rng.uniform(-40, 40, (B, N)).astype(np.float32)
    ws, wd = _wind(rng, B, "per_env")
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    w.set_risk_resolve(1)
    w.set_wind(ws, wd)
    out = w.step(yaw)
    st = w.resolve_stats()
    assert 0 < st["n_resolved"] < 0.06 * B
    parity.check_strict(out, _oracle(l["xcoords"], l["ycoords"], ws, wd, yaw))
    w.close()


def _regime(name):
    d = np.load(os.path.join(ROOT, "tests", "golden", "regime_cases.npz"))
    inp = {k: d[f"{name}_{k}"] for k in ("x", "y", "ws", "wd", "yaw")}
    ref = {k[len(name) + 5:]: d[k] for k in d.files if k.startswith(name + "_ref_")}
    return inp, ref


@pytest.mark.parametrize("name", ["thrust_ramp", "overlap_flip"])
def test_regime_cases_are_strict_with_the_resolve(name):
    """The two fuzzer-found farms the plain path can only bound (a row of turbines on the cut-in ramp of the thrust table;
    a farm 3.7e-7 from the overlap threshold, where float32 counts a grid point the other way): strict now, on the pair
    table path and on the fly, against the committed float64 reference values."""
    import parity
    from wfcrl_env_amd.backend import WfStep

    i, ref = _regime(name)
    B = 64
    w = WfStep(i["x"], i["y"], env_batch=B)
    w.set_risk_resolve(1)
    for per_farm in (False, True):
        ws, wd = (np.repeat(i["ws"], B), np.repeat(i["wd"], B)) if per_farm else (float(i["ws"][0]), float(i["wd"][0]))
        w.set_wind(ws, wd)
        out = w.step(np.repeat(i["yaw"], B, axis=0).astype(np.float32))
        assert w.resolve_stats()["n_resolved"] == B
        parity.check_strict(out, {k: np.repeat(v, B, axis=0) for k, v in ref.items()})
    w.close()


def test_resolve_on_grouped_launches_and_the_fused_env_step(layouts):
    """Direction groups (series rows, binned resets) hand the re-solve the geometry of the farm's group; the fused env
    step hands it the yaw state after the transition and gets the reward of the re-solved farms back."""
    import parity
    from wfcrl_env_amd.backend import WfStep

    l = layouts["HornsRev1_"]
    x, y, N, B, T = l["xcoords"], l["ycoords"], 80, 700, 7
    rng = np.random.default_rng(77)
    series = np.stack([rng.uniform(3.2, 14, T), rng.uniform(200, 340, T)], axis=1)
    start = rng.integers(0, T, B).astype(np.int32)
    w = WfStep(x, y, env_batch=B)
    w.set_risk_resolve(1)
    w.set_wind_series(series, start=start)
    assert w.kernel_info()["direction_groups"] == T
    n_res = 0
    for t in range(3):
        if t:
            w.wind_series_step()
        ws, wd = w.get_wind()
        yaw = rng.uniform(-35, 35, (B, N)).astype(np.float32)
        out = w.step(yaw)
        n_res += w.resolve_stats()["n_resolved"]
        parity.check_strict(out, _oracle(x, y, ws, wd, yaw))
    w.sample_wind(17, direction_step=5.0)
    ws, wd = w.get_wind()
    out = w.step(yaw)
    n_res += w.resolve_stats()["n_resolved"]
    parity.check_strict(out, _oracle(x, y, ws, wd, yaw))
    assert n_res > 0  # (the cases above do flag farms: the grouped geometry lookup of the re-solve was exercised)

    # fused env step, a wind per farm, every farm re-solved (mode 2): reward and observations of the float64 path
    w.set_risk_resolve(2)
    ws, wd = _wind(rng, B, "per_env")
    w.set_wind(ws, wd)
    w.env_config(load_coef=0.1)
    w.env_reset()
    act = rng.uniform(-5, 5, (B, N)).astype(np.float32)
    out = w.env_step(act)
    ref = _oracle(x, y, ws, wd, out["yaw"])
    parity.check_strict(out, ref, parity.TOL_F64)
    r_ref = (ref["power"] / 1e6 * 1e3 / ws[:, None] ** 3).mean(axis=1) - 0.1 * np.abs(ref["load"]).reshape(B, -1).mean(axis=1)
    assert np.abs(out["reward"] / r_ref - 1).max() < 1e-6
    w.close()


@pytest.mark.parametrize("name,B", [("Ormonde_", 96), ("HornsRev1_", 160), ("Turb16_Row5_", 200), ("Ablaincourt_", 300)])
def test_wind_veer(layouts, name, B):
    """wind_veer != 0 (reference case.yaml:36 is user-editable and goes straight to FLORIS: the Gaussian of the deficit is
    rotated by the veer angle, gauss.py rCalt).  Float32: wf_step_kernel's VEER instantiation (9 instead of 6 SOSFS sums
    per slot; these batches are below the one-block kernel's) under the per-farm contract; with the re-solve on every farm
    strict; mode 2 (every farm in float64) down to output rounding.  Shared 270 deg (exact x' ties on the grids) and a
    wind per farm."""
    import parity
    from oracle.floris_gch_numpy import ModelParams
    from wfcrl_env_amd.backend import WfStep

    l = layouts[name]
    x, y, N = l["xcoords"], l["ycoords"], l["num_turbines"]
    rng = np.random.default_rng(zlib.crc32(f"veer/{name}".encode()))
    yaw = rng.uniform(-30, 30, (B, N)).astype(np.float32)
    for veer in (3.0, -7.5):
        mp = ModelParams(veer=veer)
        w = WfStep(x, y, env_batch=B, model=dict(veer=veer))
        for mode in ("shared", "per_env"):
            ws, wd = _wind(rng, B, mode)
            w.set_wind(ws, wd)
            info = w.kernel_info()
            # the pair table holds the transverse pass only (veer does not touch it): a shared wind keeps it
            assert info["one_block_kernel"] == 0 and info["pair_table"] == (1 if mode == "shared" else 0)
            ref = _oracle(x, y, ws, wd, yaw, mp)
            w.set_risk_resolve(0)
            out = w.step(yaw)
            parity.check({k: v.copy() for k, v in out.items()}, ref, w.risk_flags(), max_flagged_frac=0.1)
            w.set_risk_resolve(1)
            parity.check_strict(w.step(yaw), ref)
            w.set_risk_resolve(2)
            parity.check_strict(w.step(yaw), ref, parity.TOL_F64)
        w.close()
    ref0 = _oracle(x, y, ws, wd, yaw)
    assert np.abs(ref["power"] / np.maximum(ref0["power"], 1e3) - 1).max() > 1e-3  # (veer does change the answer)


@pytest.mark.parametrize("one_block", ["4x1", "4x2", "2x2"])
def test_wind_veer_one_block_kernel(layouts, one_block):
    """The one-block kernel's VEER instantiations (9 SOSFS sums per slot; the source log's sigma_y0 carries cos(veer)):
    table path at 4x1 / 4x2 / 2x2, on the fly at 4x2, forced at a batch the pick would leave to the register-slot
    kernel; G = 8 is not instantiated with veer and falls back."""
    import parity
    from oracle.floris_gch_numpy import ModelParams
    from wfcrl_env_amd.backend import WfStep

    l = layouts["HornsRev1_"]
    x, y, N, B = l["xcoords"], l["ycoords"], 80, 512
    rng = np.random.default_rng(zlib.crc32(f"veer-ll/{one_block}".encode()))
    yaw = rng.uniform(-30, 30, (B, N)).astype(np.float32)
    mp = ModelParams(veer=-5.0)
    w = WfStep(x, y, env_batch=B, model=dict(veer=-5.0), kernel_choice=dict(one_block=one_block))
    for mode in ("shared", "shared_axis", "per_env"):
        ws, wd = (9.0, 270.0) if mode == "shared_axis" else _wind(rng, B, mode)
        w.set_wind(ws, wd)
        info = w.kernel_info()
        assert info["one_block_kernel"] == 1  # (on the fly: always 4x2)
        assert f'{info["lanes_per_env"]}x{info["slots_per_lane"]}' == ("4x2" if mode == "per_env" else one_block)
        ref = _oracle(x, y, ws, wd, yaw, mp)
        w.set_risk_resolve(0)
        out = w.step(yaw)
        parity.check({k: v.copy() for k, v in out.items()}, ref, w.risk_flags(), max_flagged_frac=0.1)
        w.set_risk_resolve(1)
        parity.check_strict(w.step(yaw), ref)
    w.close()
    w = WfStep(x, y, env_batch=B, model=dict(veer=-5.0), kernel_choice=dict(one_block="8x1"))
    w.set_wind(9.0, 263.0)
    assert w.kernel_info()["one_block_kernel"] == 0
    parity.check(w.step(yaw), _oracle(x, y, 9.0, 263.0, yaw, mp), w.risk_flags(), max_flagged_frac=0.1)
    w.close()


def test_veer_toggles_the_kernel_family(layouts):
    """Setting a model with veer on a live handle moves it to the VEER kernels (and back) — another family of the
    one-block kernel may serve it: the wind has to be set again, as after wf_set_kernel_choice."""
    import parity
    from oracle.floris_gch_numpy import ModelParams
    from wfcrl_env_amd.backend import WfStep

    l = layouts["HornsRev1_"]
    x, y, N, B = l["xcoords"], l["ycoords"], 80, 32768
    w = WfStep(x, y, env_batch=B)
    w.set_wind(8.0, 263.0)
    assert w.kernel_info()["one_block_kernel"] == 1
    w.set_model(dict(veer=2.0))
    with pytest.raises(ValueError, match="wf_set_wind"):
        w.step(np.zeros((B, N), np.float32))
    w.set_wind(8.0, 263.0)
    info = w.kernel_info()  # the one-block kernel's VEER instantiations exist for G <= 4 only
    assert info["one_block_kernel"] == 1 and info["lanes_per_env"] <= 4 and info["pair_table"] == 1
    rng = np.random.default_rng(2)
    yaw = rng.uniform(-30, 30, (B, N)).astype(np.float32)
    out = w.step(yaw)
    idx = np.arange(0, B, 256)
    ref = _oracle(x, y, 8.0, 263.0, yaw[idx], ModelParams(veer=2.0))
    parity.check({k: v[idx] for k, v in out.items()}, ref, w.risk_flags()[idx], max_flagged_frac=0.1)
    w.set_model(dict(veer=0.0))
    w.set_wind(8.0, 263.0)
    assert w.kernel_info()["one_block_kernel"] == 1
    w.close()


def test_fixed_seed_fuzz_sample_is_strict_with_the_resolve():
    """The fixed-seed sample of tests/tools/fuzz_parity.py (random regular / jittered / holed grids and clouds, axis-aligned
    and random directions, every kernel variant and one-block family, default and non-default models, three wind modes)
    with the re-solve on: no farm outside TOL — "flip" is not a class any more."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tests", "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    nflip, nbad = fz.run(120, 2024, resolve=True)
    assert nbad == 0 and nflip == 0


def test_kernel_choice_round_trip_and_errors(layouts):
    """wf_set_kernel_choice / wf_get_kernel_choice: the request comes back as set, impossible shapes are rejected with the
    handle left usable, and a choice drops the wind (it has to be set again)."""
    from wfcrl_env_amd.backend import WfStep

    l = layouts["HornsRev1_"]
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=256)
    assert w.kernel_choice() == dict(slot_G=0, slot_S=0, one_block=-1, ll_G=0, ll_S=0, pair_table=-1, fly_one_block=-1, far_skip=-1, calibrate=-1, mixed=-1)
    w.set_wind(8.0, 263.0)
    w.set_kernel_choice(one_block="4x2", slot="16x5")
    c = w.kernel_choice()
    assert (c["one_block"], c["ll_G"], c["ll_S"], c["slot_G"], c["slot_S"]) == (1, 4, 2, 16, 5)
    with pytest.raises(ValueError, match="wf_set_wind"):
        w.step(np.zeros((256, 80), np.float32))
    for bad in (dict(one_block="3x2"), dict(slot="7x9")):
        with pytest.raises(ValueError):
            w.set_kernel_choice(**bad)
    w.set_wind(8.0, 263.0)
    info = w.kernel_info()
    assert (info["one_block_kernel"], info["lanes_per_env"], info["slots_per_lane"]) == (1, 4, 2)
    assert np.isfinite(w.step(np.zeros((256, 80), np.float32))["power"]).all()
    w.close()


def test_float64_kernel_at_the_abi_turbine_limit():
    """256 turbines (WF_MAX_TURBINES): the float64 kernel's LDS-resident state is 72 KB per farm there — more than the
    64 KB a gfx9 workgroup used to get, inside gfx950's 160 KB."""
    import parity
    from wfcrl_env_amd.backend import WfStep

    rng = np.random.default_rng(256)
    N, B = 256, 12
    x = (np.arange(N) % 16) * 640.0 + rng.uniform(-40, 40, N)
    y = (np.arange(N) // 16) * 560.0 + rng.uniform(-40, 40, N)
    yaw = rng.uniform(-30, 30, (B, N)).astype(np.float32)
    ws, wd = rng.uniform(5, 14, B), rng.uniform(0, 360, B)
    w = WfStep(x, y, env_batch=B)
    w.set_risk_resolve(2)
    w.set_wind(ws, wd)
    out = w.step(yaw)
    assert w.resolve_stats()["n_resolved"] == B
    parity.check_strict(out, _oracle(x, y, ws, wd, yaw), parity.TOL_F64)
    w.close()


def test_both_float64_kernels_by_flagged_count(layouts):
    """Mode 2 (every farm) on Ormonde: 400 farms fit one residency of the four-wave kernel; 1500 do not — since round 6 its
    persistent blocks walk them all the same (farms of 16 turbines and more), and with the policy of rounds 3-5
    (wfk_set_resolve_policy(1): "both", by the list's length) the one-wave kernel serves them.  All three against the CPU
    oracle; the two four-wave runs bit for bit on the farms they share, the one-wave kernel at 2e-6 (another order of summation).
    (Behind a step of modes 0 / 1 the four-wave kernel serves a list of any length:
    test_a_flagged_list_longer_than_a_residency_is_one_kernels_work.)"""
    import parity
    from wfcrl_env_amd import _lib
    from wfcrl_env_amd.backend import WfStep

    lib = _lib.load()
    l = layouts["Ormonde_"]
    x, y, N = l["xcoords"], l["ycoords"], l["num_turbines"]
    rng = np.random.default_rng(1024)
    Bmax = 1500
    yaw = rng.uniform(-35, 35, (Bmax, N)).astype(np.float32)
    ws, wd = _wind(rng, Bmax, "per_env")
    outs = {}
    try:
        for tag, B, both in (("four_400", 400, 0), ("four_1500", Bmax, 0), ("one_wave_1500", Bmax, 1)):
            lib.wfk_set_resolve_policy(both)
            w = WfStep(x, y, env_batch=B)
            w.set_risk_resolve(2)
            w.set_wind(ws[:B], wd[:B])
            _level_stats()
            outs[tag] = {k: v.copy() for k, v in w.step(yaw[:B]).items()}
            st = _level_stats()
            assert w.resolve_stats()["n_resolved"] == B
            assert st["farms"] == (0 if both else B), (tag, st)  # (the four-wave kernel counts the farms it solves)
            parity.check_strict(outs[tag], _oracle(x, y, ws[:B], wd[:B], yaw[:B]), parity.TOL_F64)
            w.close()
    finally:
        lib.wfk_set_resolve_policy(0)
    for k in outs["four_400"]:
        assert np.array_equal(outs["four_400"][k].view(np.uint32), outs["four_1500"][k][:400].view(np.uint32)), k
        assert np.abs(outs["four_1500"][k].astype(np.float64) - outs["one_wave_1500"][k]).max() <= 2e-6 * max(1.0, np.abs(outs["four_1500"][k]).max()), k


def test_env_surface_switches(layouts):
    """The single-farm drop-in (`HipFlorisInterface`, what `make("<layout>_Floris")` builds) AND the vectorised env have the
    re-solve on by default — the reference computes every step in float64 —; `risk_resolve=False` opts out.  On the farm the float32
    kernel is known to flag (tests/golden/regime_cases.npz) the interface lands on the committed float64 values."""
    import parity
    from wfcrl_env_amd import environments as envs
    from wfcrl_env_amd.interface import HipFlorisInterface

    i, ref = _regime("overlap_flip")
    N = len(i["x"])
    fi = HipFlorisInterface(N, i["x"], i["y"], wind_speed=float(i["ws"][0]), wind_direction=float(i["wd"][0]))
    fi.update_command(i["yaw"][0])
    got = {"power": fi.avg_powers()[None], "wind_speed": fi.get_measure("wind_speed")[None],
           "wind_direction": fi.get_measure("wind_direction")[None], "load": fi.get_measure("load")[None] / 1e7}
    assert fi.fi.resolve_stats()["n_resolved"] == 1
    parity.check_strict(got, ref)
    off = HipFlorisInterface(N, i["x"], i["y"], wind_speed=float(i["ws"][0]), wind_direction=float(i["wd"][0]), risk_resolve=False)
    off.update_command(i["yaw"][0])
    assert off.fi.risk_flags()[0] != 0
    with pytest.raises(RuntimeError):  # nothing is recorded with the re-solve off: stale flags are not handed out
        off.fi.resolve_stats()
    import torch

    venv = envs.make("HornsRev1_Floris", env_batch=512, log=False)  # default: strict
    venv.reset(seed=3)
    venv.step({"yaw": torch.zeros((512, 80), device="cuda")})
    assert not venv.fi.risk_flags().any()
    venv.close()
    fast = envs.make("HornsRev1_Floris", env_batch=512, log=False, risk_resolve=False)  # opt-out: flags stay up
    fast.reset(seed=3)
    fast.step({"yaw": torch.zeros((512, 80), device="cuda")})
    assert fast.fi.risk_flags().any()
    fast.close()


@pytest.mark.parametrize("name", ["thrust_ramp", "overlap_flip"])
def test_default_mode_is_the_reference_contract(name, monkeypatch):
    """ABI 6: a new handle starts in wf_set_risk_resolve mode 1 — a binding that calls nothing but wf_create ... wf_step gets
    what the reference computes in float64 (interface.py:564): on the two farms float32 cannot decide, no flag is left and
    every turbine is inside the tolerances, on the host path and on the device path, plain step and fused env step."""
    import parity
    import torch
    from wfcrl_env_amd.backend import WfStep

    monkeypatch.delenv("WF_RISK_RESOLVE", raising=False)  # (this file's fixture seeds mode 0 for the float32-contract tests)
    i, ref = _regime(name)
    B = 32
    w = WfStep(i["x"], i["y"], env_batch=B)
    assert w.risk_resolve() == 1
    w.set_wind(float(i["ws"][0]), float(i["wd"][0]))
    yaw = np.repeat(i["yaw"], B, axis=0).astype(np.float32)
    want = {k: np.repeat(v, B, axis=0) for k, v in ref.items()}
    out = w.step(yaw)
    assert not w.risk_flags().any() and w.resolve_stats()["n_resolved"] == B
    parity.check_strict(out, want)
    out = w.step(torch.from_numpy(yaw).cuda())
    assert not w.risk_flags().any()
    parity.check_strict({k: v.cpu().numpy() for k, v in out.items()}, want)
    w.env_config(load_coef=0.1)
    w.env_reset()
    st = w.env_get_state()
    st["yaw"][:] = yaw
    w.env_set_state(st)
    e = w.env_step(None, want=("power", "wind_speed", "wind_direction", "load"))
    assert not w.risk_flags().any()
    parity.check_strict(e, want)
    w.set_risk_resolve(0)  # the opt-out: the float32 kernel on its own raises its flag
    w.step(yaw)
    assert w.risk_resolve() == 0 and w.risk_flags().all()
    w.close()


@pytest.mark.parametrize("kernel", ["", "4x2", "slot"])
def test_thrust_coefficient_near_one_is_never_left_in_float32(kernel, monkeypatch):
    """a thrust table that reaches 0.9999 (a user table; nrel_5MW peaks at 0.99).  Behind a turbine with
    Ct > 0.995 float32 has no bound (1 - Ct cancels, velocities approach zero) — round 4 exempted such farms from every
    bound on the float32-only path.  Now the kernels raise WF_RISK_THRUST_UNITY for them and they are re-solved in float64 in
    EVERY mode: with the library's default every farm is strict; with wf_set_risk_resolve(0) the unity farms are strict
    too (their flag is gone), the other flagged farms stay within their per-flag bounds, nothing is exempt."""
    import parity
    from oracle.floris_gch_numpy import ModelParams
    from wfcrl_env_amd.backend import WfStep

    monkeypatch.delenv("WF_RISK_RESOLVE", raising=False)
    rng = np.random.default_rng(995)
    D = 126.0
    gx, gy = np.meshgrid(np.arange(7) * 3.0 * D, np.arange(6) * 2.5 * D)  # a dense farm: 42 turbines, 3 D x 2.5 D
    x, y = (gx + 0.07 * D * rng.standard_normal(gx.shape)).ravel(), (gy + 0.07 * D * rng.standard_normal(gy.shape)).ravel()
    N, B = x.size, 384
    mp = ModelParams()
    tct = np.clip(np.asarray(mp.table_ct) * 1.35, 0.0, 0.9999)
    mp.table_ct = list(tct)
    model = dict(table_ws=list(mp.table_ws), table_ct=list(tct), table_cp=list(mp.table_cp))
    choice = None if not kernel else (dict(one_block=False) if kernel == "slot" else dict(one_block=kernel))
    yaw = rng.uniform(-30, 30, (B, N)).astype(np.float32)
    w = WfStep(x, y, env_batch=B, model=model, kernel_choice=choice)
    assert w.risk_resolve() == 1
    for ws, wd in ((rng.uniform(3.5, 12.0, B), np.full(B, 268.0)), (rng.uniform(3.5, 12.0, B), rng.normal(270, 20, B) % 360)):
        ref = _oracle(x, y, ws, wd, yaw, mp)
        w.set_risk_resolve(1)
        w.set_wind(ws, wd)
        out = w.step(yaw)
        raw = w.resolve_stats()["raw_flags"]
        assert (raw & parity.RISK_THRUST_UNITY).any() and not w.risk_flags().any()
        parity.check_strict(out, ref)
        w.set_risk_resolve(0)  # the float32-only opt-out: unity farms are re-solved all the same
        out0 = {k: v.copy() for k, v in w.step(yaw).items()}
        fl0 = w.risk_flags()
        assert not (fl0 & parity.RISK_THRUST_UNITY).any()
        unity = (raw & parity.RISK_THRUST_UNITY) != 0
        assert not fl0[unity].any()  # solved in float64, flags cleared
        parity.check_strict({k: v[unity] for k, v in out0.items()}, {k: v[unity] for k, v in ref.items()})
        parity.check(out0, ref, fl0, max_flagged_frac=1.0)
    w.close()


def _level_stats(reset=True):
    import ctypes as C

    from wfcrl_env_amd import _lib

    buf = (C.c_ulonglong * 8)()
    _lib.load().wfk_res_level_stats(buf, 1 if reset else 0)
    return dict(zip(("farms", "repeated_without_levels", "level_stages", "sources_in_levels", "sequential_stages", "farms_with_helper_waves"), list(buf)))


@pytest.mark.parametrize("name", ["HornsRev1_", "HornsRev2_", "Turb_TCRWP_", "Turb16_Row5_", "Ormonde_", "WMR_"])
@pytest.mark.parametrize("wind", ["270", "287.5", "per_farm"])
def test_level_stages_are_the_sequential_solve_bit_for_bit(layouts, name, wind):
    """Round 6 (csrc/wf_resolve.hip: Lvl4Shared): consecutive sources that put no deficit on each other are solved as ONE
    stage of the four-wave kernel.  Every sum is still taken in source order, so the outputs must be the same BITS as with
    every source a stage of its own (wfk_set_resolve_levels(0)); no farm may have failed a level's check; the levels must
    have been used where the layout has them; and the result is the CPU oracle's at TOL_F64 like every float64 solve."""
    import parity
    from wfcrl_env_amd import _lib
    from wfcrl_env_amd.backend import WfStep

    lib = _lib.load()
    l = layouts[name]
    x, y, N = l["xcoords"], l["ycoords"], l["num_turbines"]
    B = 96
    rng = np.random.default_rng(zlib.crc32(f"levels/{name}/{wind}".encode()))
    yaw = rng.uniform(-40, 40, (B, N)).astype(np.float32)
    if wind == "per_farm":
        ws, wd = _wind(rng, B, "per_env")
    else:
        ws, wd = np.array([9.0]), np.array([float(wind)])
    outs, stats = [], []
    try:
        for on in (1, 0):
            lib.wfk_set_resolve_levels(on)
            w = WfStep(x, y, env_batch=B)
            w.set_risk_resolve(2)
            w.set_wind(ws, wd)
            _level_stats()
            outs.append({k: v.copy() for k, v in w.step(yaw).items()})
            stats.append(_level_stats())
            w.close()
    finally:
        lib.wfk_set_resolve_levels(1)
    on, off = stats
    assert on["farms"] == B and off["farms"] == B and off["level_stages"] == 0 and off["sequential_stages"] == B * N
    assert on["repeated_without_levels"] == 0  # (the geometric rule is a heuristic: a failure is not an error, but none is expected here)
    assert on["sources_in_levels"] + on["sequential_stages"] == B * N
    if name.startswith("HornsRev"):
        assert on["sources_in_levels"] >= 0.5 * B * N, on  # grid farms: most sources stand in columns
    for k in outs[0]:
        assert np.array_equal(outs[0][k].view(np.uint32), outs[1][k].view(np.uint32)), k
    parity.check_strict(outs[0], _oracle(x, y, ws, wd, yaw), parity.TOL_F64)


def _helper_farms():
    import ctypes as C

    from wfcrl_env_amd import _lib

    buf = (C.c_ulonglong * 8)()
    _lib.load().wfk_res_level_stats(buf, 1)
    return int(buf[0]), int(buf[5])


@pytest.mark.parametrize("name", ["HornsRev2_", "Ormonde_", "Turb16_Row5_"])
def test_helper_waves_leave_the_bits_alone(layouts, name):
    """Round 6 (csrc/wf_resolve.hip: HELPER WAVES): a launch of 512 threads gives a farm's block four more waves for the pair
    passes of its level stages.  The sums are taken in member order whoever computes the terms: with the helper waves on every
    launch (wfk_set_resolve_helpers(2)) and on none (0) the outputs are the same BITS, and the farms were solved the way asked."""
    from wfcrl_env_amd import _lib
    from wfcrl_env_amd.backend import WfStep

    lib = _lib.load()
    l = layouts[name]
    x, y, N = l["xcoords"], l["ycoords"], l["num_turbines"]
    B = 160
    rng = np.random.default_rng(zlib.crc32(f"helpers/{name}".encode()))
    yaw = rng.uniform(-40, 40, (B, N)).astype(np.float32)
    ws, wd = _wind(rng, B, "per_env")
    outs = []
    try:
        for mode in (2, 0):
            lib.wfk_set_resolve_helpers(mode)
            w = WfStep(x, y, env_batch=B)
            w.set_risk_resolve(2)
            w.set_wind(ws, wd)
            _helper_farms()
            outs.append({k: v.copy() for k, v in w.step(yaw).items()})
            farms, helped = _helper_farms()
            assert farms == B and helped == (B if mode else 0), (mode, farms, helped)
            w.close()
    finally:
        lib.wfk_set_resolve_helpers(1)
    for k in outs[0]:
        assert np.array_equal(outs[0][k].view(np.uint32), outs[1][k].view(np.uint32)), k


def test_the_launch_width_follows_the_length_of_the_previous_list(layouts):
    """The flagged list's length is known on the device only; the host picks the width of the float64 launch — 512 threads with
    helper waves for a short list, 256 for a long one — from the length the PREVIOUS launch found (wf_resolve.h: seen_host).  The
    first launch of a handle is a narrow one; behind a short list the next is wide; behind a long list narrow again.  Which width
    ran changes no bit: every step's outputs are compared with a handle that never uses the helper waves."""
    from wfcrl_env_amd import _lib
    from wfcrl_env_amd.backend import WfStep

    lib = _lib.load()
    l = layouts["HornsRev2_"]
    x, y, N = l["xcoords"], l["ycoords"], l["num_turbines"]
    B = 4096
    rng = np.random.default_rng(20260)
    yaw = rng.uniform(-30, 30, (B, N)).astype(np.float32)
    ws, wd = _wind(rng, B, "per_env")
    w = WfStep(x, y, env_batch=B)
    w.set_risk_resolve(1)
    w.set_wind(ws, wd)
    ref = WfStep(x, y, env_batch=B)
    ref.set_risk_resolve(1)
    ref.set_wind(ws, wd)
    seen = []
    try:
        for band in (1.0e-5, 1.0e-5, 3.0e-3, 3.0e-3, 1.0e-5, 1.0e-5):
            w.set_risk_guard(band)
            ref.set_risk_guard(band)
            lib.wfk_set_resolve_helpers(1)
            _helper_farms()
            o = {k: v.copy() for k, v in w.step(yaw).items()}
            w.sync()
            farms, helped = _helper_farms()
            assert farms == w.resolve_stats()["n_resolved"]
            seen.append((farms, helped))
            lib.wfk_set_resolve_helpers(0)
            o_ref = ref.step(yaw)
            for k in o:
                assert np.array_equal(o[k].view(np.uint32), o_ref[k].view(np.uint32)), (band, k)
    finally:
        lib.wfk_set_resolve_helpers(1)
        w.close()
        ref.close()
    (f0, h0), (f1, h1), (f2, h2), (f3, h3), (f4, h4), (f5, h5) = seen
    assert 0 < f0 <= 320 and f1 == f0 and f2 > 512 and f3 == f2 and f4 == f0, seen
    assert h0 == 0          # nothing known yet: narrow
    assert h1 == f1         # behind a short list: wide, the helper waves at work
    assert h2 == 0          # a long list met by a wide launch: the helper waves have returned
    assert h3 == 0 and h4 == 0  # behind a long list: narrow (whatever the list turns out to be)
    assert h5 == f5         # ... and wide again behind the short one


def test_a_failed_level_check_falls_back_to_the_sequential_solve(layouts):
    """The rule that admits turbines to a level is geometric (8.6 wake widths apart); the guarantee is the check at the end
    of the stage.  A model whose wakes grow twenty times faster than the rule assumes (ka, kb of a user's case.yaml) makes
    members reach each other: such farms must be solved again without levels and still be the oracle's."""
    import parity
    from oracle.floris_gch_numpy import ModelParams
    from wfcrl_env_amd.backend import WfStep

    l = layouts["Turb16_Row5_"]
    x, y, N = l["xcoords"], l["ycoords"], l["num_turbines"]
    B = 64
    rng = np.random.default_rng(99)
    yaw = rng.uniform(-25, 25, (B, N)).astype(np.float32)
    ws, wd = np.array([9.0]), np.array([283.0])
    outs = {}
    for wide in (False, True):
        model = dict(ambient_ti=0.06)
        mp = ModelParams()
        if wide:  # the level rule prices the growth rate at TI = 0.3: an ambient TI of 0.9 triples it
            model = dict(ambient_ti=0.9)
            mp = ModelParams(ambient_ti=0.9)
        w = WfStep(x, y, env_batch=B, model=model)
        w.set_risk_resolve(2)
        w.set_wind(ws, wd)
        _level_stats()
        out = w.step(yaw)
        st = _level_stats()
        parity.check_strict(out, _oracle(x, y, ws, wd, yaw, mp), parity.TOL_F64)
        outs[wide] = st
        w.close()
    assert outs[False]["repeated_without_levels"] == 0 and outs[False]["level_stages"] > 0
    # ... and at three times the assumed growth rate the members of Turb16_Row5's columns do reach each other at 283 deg: every
    # farm failed a check, was solved again stage by stage — and is the oracle's (asserted above)
    assert outs[True]["farms"] == B and outs[True]["repeated_without_levels"] == B, outs[True]


def test_a_flagged_list_longer_than_a_residency_is_one_kernels_work(layouts):
    """Round 6: ONE float64 launch behind a step — the four-wave kernel's persistent blocks walk a flagged list of any length
    (rounds 3-5: lists beyond a residency went to the one-wave kernel, enqueued beside it).  A guard band of 1e-2 flags
    thousands of Ormonde farms: every one must be re-solved (strict, flags cleared), by the four-wave kernel alone."""
    import parity
    from wfcrl_env_amd.backend import WfStep

    l = layouts["Ormonde_"]
    x, y, N = l["xcoords"], l["ycoords"], l["num_turbines"]
    B = 8192
    rng = np.random.default_rng(2468)
    yaw = rng.uniform(-35, 35, (B, N)).astype(np.float32)
    ws, wd = _wind(rng, B, "per_env")
    w = WfStep(x, y, env_batch=B)
    w.set_risk_guard(1e-2)
    w.set_risk_resolve(1)
    w.set_wind(ws, wd)
    _level_stats()
    out = w.step(yaw)
    st = w.resolve_stats()
    lv = _level_stats()
    assert st["n_resolved"] > 1500 and not w.risk_flags().any(), st["n_resolved"]
    assert lv["farms"] == st["n_resolved"]  # (the four-wave kernel's own count: it solved them all)
    idx = np.concatenate([np.nonzero(st["raw_flags"])[0][:300], rng.choice(B, 100, replace=False)])
    ref = _oracle(x, y, ws[idx], wd[idx], yaw[idx])
    parity.check_strict({k: v[idx] for k, v in out.items()}, ref)
    w.close()
