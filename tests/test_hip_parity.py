"""GPU: parity of the HIP path, called through the C ABI (ctypes), against the float64 oracle.

The contract is in tests/parity.py (fp32 device arithmetic vs float64 oracle; north_star: per-turbine power within
1e-4): STRICT on every farm whose risk flags are 0 — power |dP| / max(P, 1 kW) <= 1e-4, wind speed 5e-5 relative, wind
direction 3e-4 deg, TI 5e-6, std u/v/w 1e-4 m/s, on every turbine, no count allowance.  A farm is flagged by the
kernel itself (include/wfstep.h WF_RISK_*) when a deficit comes within the guard band of the overlap threshold
`deficit * Uinit > 0.05` (the one state-dependent discontinuity of the model) or a turbine sits on a knee
of the power table; flagged farms must stay inside the bounded signature of such an event, must be few, and the flag
must not be spurious (the oracle's own margin to the threshold is checked).
"""
import json
import os

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

GOLD = os.path.join(ROOT, "tests", "golden", "oracle_goldens.npz")


def _check(got, ref, max_flagged_frac=0.05):
    """`got` carries the kernel's risk flags under "flags" (see _with_flags)."""
    import parity

    return parity.check(got, ref, got["flags"], max_flagged_frac)


def _with_flags(w, out):
    out = dict(out)
    out["flags"] = w.risk_flags()
    return out


def _oracle(x, y, ws, wd, yaw, mp=None):
    from oracle import c_oracle

    return c_oracle.farm_step_batch(x, y, ws, wd, np.asarray(yaw, dtype=np.float64), mp, margin=True)


def _step(x, y, ws, wd, yaw):
    from wfcrl_env_amd.backend import WfStep

    w = WfStep(x, y, env_batch=yaw.shape[0])
    w.set_wind(ws, wd)
    out = _with_flags(w, w.step(yaw))
    info = w.kernel_info()
    w.close()
    return out, info


def test_extension_is_loaded_not_a_fallback():
    from wfcrl_env_amd import _lib

    lib = _lib.load()
    assert os.path.samefile(lib._name, os.path.join(ROOT, "wfcrl-env_amd", "libwfstep.so"))


def test_reference_kat_on_gpu(kat1):
    yaw = np.zeros((1, 7), np.float32)
    out, _ = _step(kat1["xcoords"], kat1["ycoords"], kat1["wind_speed_free"], kat1["wind_direction_free"], yaw)
    assert np.abs(out["wind_speed"][0] / np.array(kat1["wind_speed"]) - 1).max() <= 2e-6
    assert np.abs(out["wind_direction"][0] - np.array(kat1["wind_direction"])).max() <= 1e-4


def test_committed_goldens(layouts):
    g = np.load(GOLD)
    for key in sorted({k.split("__")[0] for k in g.files}):
        l = layouts[key + "_"]
        yaw = g[f"{key}__yaw"].astype(np.float32)
        out, _ = _step(l["xcoords"], l["ycoords"], g[f"{key}__ws"], g[f"{key}__wd"], yaw)
        ref = {k: g[f"{key}__{k}"] for k in ("power", "wind_speed", "wind_direction", "load")}
        _check(out, ref)


@pytest.mark.parametrize("name,B", [("Turb3_Row1_", 300), ("Turb6_Row2_", 300), ("Ablaincourt_", 1000),
                                    ("Turb16_Row5_", 500), ("Turb32_Row5_", 200), ("Turb_TCRWP_", 200),
                                    ("Ormonde_", 150), ("WMR_", 150), ("HornsRev1_", 160), ("HornsRev2_", 130)])
@pytest.mark.parametrize("mode", ["shared", "per_env"])
def test_parity_random_yaw_and_wind(layouts, name, B, mode):
    from oracle import c_oracle

    l = layouts[name]
    N = l["num_turbines"]
    import zlib

    rng = np.random.default_rng(zlib.crc32(f"{name}/{mode}".encode()))  # deterministic across processes
    yaw = rng.uniform(-40, 40, (B, N)).astype(np.float32)
    if mode == "shared":
        ws, wd = np.array([8.0]), np.array([270.0])  # exact x' ties on the grid layouts
    else:
        ws = np.clip(8 * rng.weibull(8, B), 3, 28)
        wd = rng.normal(270, 20, B) % 360
    out, info = _step(l["xcoords"], l["ycoords"], ws, wd, yaw)
    assert info["lanes_per_env"] * info["slots_per_lane"] >= N
    ref = _oracle(l["xcoords"], l["ycoords"], ws, wd, yaw)
    _check(out, ref)


@pytest.mark.parametrize("table", ["nrel_5MW_floris3", "nrel_5MW_survey_a5"])
def test_above_rated_wind_speeds(layouts, table):
    """The above-rated branch of the power table (rotor-effective speeds >= 11.5 m/s: `options={"wind_speed": 13}`, any
    time series), on both nrel_5MW Cp columns shipped as data: 12 and 18 m/s shared, and a speed per farm from rated to
    past cut-out (the cut-out drop raises WF_RISK_POWER_KNEE).  With the default column the front row makes 5.000 MW at
    18 m/s; with SURVEY A.5's it makes 4.9936 MW — the difference VERDICT r2 asked to be written down."""
    from oracle.floris_gch_numpy import ModelParams, turbine_table
    from wfcrl_env_amd.backend import WfStep
    from wfcrl_env_amd.backend import turbine_table as product_table

    l = layouts["Ormonde_"]
    x, y, N, B = l["xcoords"], l["ycoords"], l["num_turbines"], 256
    rng = np.random.default_rng(1218)
    yaw = rng.uniform(-25, 25, (B, N)).astype(np.float32)
    yaw[0] = 0.0
    mp = ModelParams(**turbine_table(table))
    w = WfStep(x, y, env_batch=B, model=product_table(table))
    for ws in (12.0, 18.0):
        w.set_wind(ws, 270.0)
        out = _with_flags(w, w.step(yaw))
        _check(out, _oracle(x, y, ws, 270.0, yaw, mp))
        if ws == 18.0:  # the unwaked front row at yaw 0: rated power on the plateau column, 0.13 % short of it on the other
            top = out["power"][0].max()
            assert abs(top / 5.0e6 - 1) < (1e-4 if table == "nrel_5MW_floris3" else 3e-3)
            assert (abs(top / 5.0e6 - 1) > 1e-3) == (table == "nrel_5MW_survey_a5")
    ws = rng.uniform(11.0, 26.0, B)
    wd = rng.normal(270, 20, B) % 360
    w.set_wind(ws, wd)
    out = _with_flags(w, w.step(yaw))
    ref = _oracle(x, y, ws, wd, yaw, mp)
    s = _check(out, ref, max_flagged_frac=0.2)
    past = ref["wind_speed"] * np.cos(np.radians(yaw)) ** (1.88 / 3) > 25.03  # rotor-effective speed past cut-out
    assert past.any() and (out["power"][past] == 0).all() and s["n_flagged"] > 0
    w.close()


def test_baseline_config2_named_turb16_tcrwp(layouts):
    """BASELINE.json configs[2] as named: `Turb16_TCRWP_Floris` = the first 16 turbines of the TCRWP layout (the
    reference's README spelling; its registry only has Turb16_Row5 and Turb_TCRWP — SURVEY Appendix C2), random yaw
    actions as a random walk, shared 8 m/s / 270 deg and per-farm wind."""
    t = layouts["Turb_TCRWP_"]
    x, y = t["xcoords"][:16], t["ycoords"][:16]
    rng = np.random.default_rng(1236)
    B = 512
    yaw = np.zeros((B, 16), np.float32)
    from wfcrl_env_amd.backend import WfStep

    w = WfStep(x, y, env_batch=B)
    for mode in ("shared", "per_env"):
        if mode == "shared":
            ws, wd = np.array([8.0]), np.array([270.0])
        else:
            ws, wd = np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360
        w.set_wind(ws, wd)
        for _ in range(3):
            yaw = np.clip(yaw + rng.uniform(-5, 5, (B, 16)), -40, 40).astype(np.float32)
            _check(_with_flags(w, w.step(yaw)), _oracle(x, y, ws, wd, yaw))
    w.close()


def test_risk_flags_mark_exactly_the_farms_near_the_threshold(layouts):
    """The kernel's WF_RISK_OVERLAP flag against the oracle's own margin to the overlap threshold, on a case with many
    partial overlaps (oblique directions), for several guard bands: every farm whose float64 margin is well inside the
    band is flagged, no farm whose margin is far outside is, and with the band at 0 nothing is flagged."""
    from wfcrl_env_amd.backend import WfStep

    l = layouts["HornsRev1_"]
    rng = np.random.default_rng(31)
    B = 2048
    yaw = rng.uniform(-30, 30, (B, 80)).astype(np.float32)
    ws, wd = rng.uniform(6, 12, B), rng.uniform(0, 360, B)
    ref = _oracle(l["xcoords"], l["ycoords"], ws, wd, yaw)
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    w.set_wind(ws, wd)
    for band in (0.0, 5e-5, 1e-3, 1e-2):
        w.set_risk_guard(band)
        out = w.step(yaw)
        fl = (w.risk_flags() & 1) != 0
        if band == 0.0:
            assert not fl.any()
            continue
        # float32 rounding of the deficit (<~ 1e-5 relative after the recurrence) and the +-5 % row dependence of the band
        assert fl[ref["margin"] < 0.9 * band - 3e-5].all(), band
        assert not fl[ref["margin"] > 1.1 * band + 1e-4].any(), band
        if band >= 1e-3:
            assert fl.any()
    w.set_risk_guard(1e-5)
    w.close()


def test_procedural_rows_and_single_turbine():
    from oracle import c_oracle

    rng = np.random.default_rng(5)
    for n in (1, 2, 5, 12):
        x = [i * 4 * 126.0 for i in range(n)]  # reference data_cases.py:513-519
        y = [0.0] * n
        yaw = rng.uniform(-40, 40, (64, n)).astype(np.float32)
        out, _ = _step(x, y, 8.0, 270.0, yaw)
        ref = _oracle(x, y, 8.0, 270.0, yaw)
        _check(out, ref)


def test_wind_edge_cases(layouts):
    from oracle import c_oracle

    l = layouts["Turb6_Row2_"]
    rng = np.random.default_rng(11)
    yaw = rng.uniform(-40, 40, (8, 6)).astype(np.float32)
    yaw[0] = 0.0
    yaw[1] = 40.0
    yaw[2] = -40.0
    for ws, wd in [(3.0, 270.0), (24.9, 300.0), (28.0, 250.0), (11.4, -90.0), (8.0, 630.0), (8.0, 271.0), (5.0, 0.0)]:
        out, _ = _step(l["xcoords"], l["ycoords"], ws, wd, yaw)
        ref = _oracle(l["xcoords"], l["ycoords"], ws, wd, yaw)
        for k, v in out.items():
            assert np.isfinite(v).all(), k
        # ws = 3 m/s puts every turbine on the cut-in knee of the power table (WF_RISK_POWER_KNEE on every farm)
        _check(out, ref, max_flagged_frac=1.0)
    # above cut-out the power table returns 0 (waked turbines may fall back below cut-out and produce)
    out, _ = _step([0.0], [0.0], 28.0, 270.0, np.zeros((1, 1), np.float32))
    assert out["power"][0, 0] == 0.0


def test_max_turbines_and_ragged_batch():
    """N = 256 (WF_MAX_TURBINES) on the 64-lane x 4-slot variant; batch sizes that do not fill a block."""
    from oracle import c_oracle

    rng = np.random.default_rng(2)
    x = (np.arange(256) % 16) * 700.0 + rng.uniform(-50, 50, 256)
    y = (np.arange(256) // 16) * 600.0 + rng.uniform(-50, 50, 256)
    yaw = rng.uniform(-30, 30, (3, 256)).astype(np.float32)
    ws = np.array([7.0, 9.0, 12.0])
    wd = np.array([268.0, 281.0, 255.0])
    out, info = _step(x, y, ws, wd, yaw)
    assert (info["lanes_per_env"], info["slots_per_lane"]) == (64, 4)
    ref = _oracle(x, y, ws, wd, yaw)
    _check(out, ref)
    for B in (1, 31, 33):
        l = json.load(open(os.path.join(ROOT, "wfcrl-env_amd", "environments", "layouts.json")))["Ablaincourt_"]
        yaw = rng.uniform(-40, 40, (B, 7)).astype(np.float32)
        out, _ = _step(l["xcoords"], l["ycoords"], 8.0, 270.0, yaw)
        ref = _oracle(l["xcoords"], l["ycoords"], 8.0, 270.0, yaw)
        _check(out, ref)


def test_full_size_properties_hornsrev1(layouts):
    """BASELINE config 4 at full size (N = 80, B = 65536): size-independent properties."""
    import torch

    from oracle import c_oracle
    from wfcrl_env_amd.backend import WfStep

    l = layouts["HornsRev1_"]
    N, B = 80, 65536
    g = torch.Generator(device="cpu").manual_seed(1238)
    yaw = (torch.rand((B, N), generator=g) * 80 - 40).float()
    yaw[B // 2:] = yaw[: B // 2].flip(0)  # second half = first half, reversed env order
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    w.set_wind(8.0, 270.0)
    d = w.step(yaw.cuda())
    w.sync()
    out = {k: v.cpu().numpy() for k, v in d.items()}
    flags = w.risk_flags()
    for v in out.values():
        assert np.isfinite(v).all()
    # (1) an env's result does not depend on where it sits in the batch: bit-exact
    for k, v in out.items():
        assert np.array_equal(v[B // 2:], v[: B // 2][::-1]), k
    # (2) idempotence: a second step on the same inputs is bit-identical (the solve is stateless)
    d2 = w.step(yaw.cuda())
    w.sync()
    for k in out:
        assert np.array_equal(out[k], d2[k].cpu().numpy()), k
    # (3) the most upstream turbines are unwaked: 0.99670412 * ws
    assert abs(out["wind_speed"].max() / 8.0 - 0.99670412) < 2e-6
    # (4) physical bounds
    assert out["power"].min() >= 0 and out["power"].max() <= 1.70e6
    assert out["load"][..., 0].min() >= 0.06 - 1e-7
    # (5) a random subset against the oracle
    idx = np.random.default_rng(0).choice(B, 96, replace=False)
    ref = _oracle(l["xcoords"], l["ycoords"], 8.0, 270.0, yaw.numpy()[idx])
    _check(dict({k: v[idx] for k, v in out.items()}, flags=flags[idx]), ref)
    # (6) the flagged fraction of the BASELINE configuration itself: HornsRev1 at 270 deg has no partial overlaps
    assert (flags != 0).mean() <= 0.01, (flags != 0).mean()
    w.close()


def _full_size_properties(w, l, N, B, yaw, ws, wd, n_sample=96, max_flagged=0.05, upstream=0.99670412 * 8.0):
    """The size-independent properties of test_full_size_properties_hornsrev1 on a handle whose wind is set: `yaw` (B, N)
    torch CPU tensor whose second half is the first half in reversed farm order.  Returns (outputs, flags)."""
    d = w.step(yaw.cuda())
    w.sync()
    out = {k: v.cpu().numpy() for k, v in d.items()}
    flags = w.risk_flags()
    for v in out.values():
        assert np.isfinite(v).all()
    for k, v in out.items():  # batch-position independence, bit-exact
        assert np.array_equal(v[B // 2:], v[: B // 2][::-1]), k
    assert np.array_equal(flags[B // 2:], flags[: B // 2][::-1])
    d2 = w.step(yaw.cuda())  # idempotence
    w.sync()
    for k in out:
        assert np.array_equal(out[k], d2[k].cpu().numpy()), k
    assert abs(out["wind_speed"].max() - upstream) < 2e-6 * upstream
    assert out["power"].min() >= 0 and out["load"][..., 0].min() >= 0.06 - 1e-7
    idx = np.random.default_rng(0).choice(B, n_sample, replace=False)
    ref = _oracle(l["xcoords"], l["ycoords"], ws, wd, yaw.numpy()[idx])
    _check(dict({k: v[idx] for k, v in out.items()}, flags=flags[idx]), ref, max_flagged)
    return out, flags, idx, ref


def test_full_size_properties_turb16_tcrwp(layouts):
    """BASELINE configs[2] at full size: Turb16_TCRWP (the first 16 TCRWP turbines) x 16384 farms, random-walk yaw."""
    import torch

    import parity
    from wfcrl_env_amd.backend import WfStep

    t = layouts["Turb_TCRWP_"]
    l = {"xcoords": t["xcoords"][:16], "ycoords": t["ycoords"][:16]}
    N, B = 16, 16384
    g = torch.Generator(device="cpu").manual_seed(1237)
    yaw = torch.zeros((B, N))
    for _ in range(6):  # SURVEY cfg3: dyaw ~ U(-5, 5) random walk clipped to +-40
        yaw = (yaw + torch.rand((B, N), generator=g) * 10 - 5).clamp_(-40, 40)
    yaw = yaw.float()
    yaw[B // 2:] = yaw[: B // 2].flip(0)
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    w.set_wind(8.0, 270.0)
    out, flags, idx, ref = _full_size_properties(w, l, N, B, yaw, 8.0, 270.0)
    assert (flags != 0).mean() <= 0.02, (flags != 0).mean()
    # with the re-solve on (the envs' default): every sampled farm strict, no flag left
    w.set_risk_resolve(1)
    d = w.step(yaw.cuda())
    w.sync()
    assert not w.risk_flags().any()
    parity.check_strict({k: v.cpu().numpy()[idx] for k, v in d.items()}, ref)
    w.close()


def test_full_size_properties_hornsrev2_sweep(layouts):
    """BASELINE configs[4] at full size: HornsRev2 (91 turbines) x 131072 farms at two directions of the sweep
    wd(t) = 270 + 30 sin(2 pi t / 200), set on the device like bench.py --config cfg5 does."""
    import torch

    import parity
    from wfcrl_env_amd.backend import WfStep

    l = layouts["HornsRev2_"]
    N, B = 91, 131072
    g = torch.Generator(device="cpu").manual_seed(1239)
    yaw = torch.zeros((B // 2, N))
    for _ in range(6):
        yaw = (yaw + torch.rand((B // 2, N), generator=g) * 10 - 5).clamp_(-40, 40)
    yaw = torch.cat([yaw, yaw.flip(0)]).float()
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    for t in (0, 37):
        wd = float(270.0 + 30.0 * np.sin(2 * np.pi * t / 200.0))
        w.set_wind(torch.full((1,), 8.0, device="cuda", dtype=torch.float64), torch.full((1,), wd, device="cuda", dtype=torch.float64))
        w.set_risk_resolve(0)
        out, flags, idx, ref = _full_size_properties(w, l, N, B, yaw, 8.0, wd, n_sample=64)
        assert (flags != 0).mean() <= 0.05, (flags != 0).mean()
        w.set_risk_resolve(1)
        d = w.step(yaw.cuda())
        w.sync()
        assert not w.risk_flags().any()
        # the sample plus up to 32 of the farms the float32 kernel flagged: all strict after the re-solve
        pick = np.unique(np.concatenate([idx, np.flatnonzero(flags != 0)[:32]]))
        refp = _oracle(l["xcoords"], l["ycoords"], 8.0, wd, yaw.numpy()[pick])
        parity.check_strict({k: v.cpu().numpy()[pick] for k, v in d.items()}, refp)
    w.close()


@pytest.mark.parametrize("name,wd,veer", [("HornsRev1_", 270.0, 0.0), ("HornsRev1_", 283.0, 0.0), ("HornsRev1_", 231.0, 0.0),
                                          ("HornsRev2_", 270.0, 0.0), ("HornsRev2_", 255.0, 0.0), ("HornsRev2_", 300.0, 0.0),
                                          ("HornsRev1_", 270.0, 4.0), ("HornsRev2_", 291.0, -6.0)])
def test_far_skip_is_a_no_op_in_float32(layouts, name, wd, veer):
    """The far-source / far-pair skip of the one-block kernel (csrc/wf_kernels_ll.hip: far_bound, pass2) claims that what
    it leaves out cannot change a float32 result.  Same farms with the skip on and off (wf_kernel_choice::far_skip) on
    every table-path family: identical risk flags and BIT-IDENTICAL outputs (a skipped pair would have added ~1e-17 of a
    deficit^2 to a sum of ~1e-4: less than half an ulp).  |yaw| up to 40 deg, wind speeds from cut-in to rated."""
    import torch

    from wfcrl_env_amd.backend import WfStep

    l = layouts[name]
    N, B = len(l["xcoords"]), 2048
    rng = np.random.default_rng(int(wd) * 7 + N)
    yaw = torch.from_numpy(rng.uniform(-40, 40, (B, N)).astype(np.float32)).cuda()
    ws = np.where(rng.random(B) < 0.5, rng.uniform(3.2, 6.0, B), rng.uniform(6.0, 12.0, B))
    fams = (("2x2", 8.0), ("2x2", ws), ("4x2", ws), ("4", 8.0), ("8", ws))  # one speed (constants in SGPRs) / a speed per farm
    if veer:  # wind veer: the rotated Gaussian, bounded through its larger width; the families instantiated with it
        fams = (("2x2", 8.0), ("4x2", ws), ("4", ws))
    for fam, wsx in fams:
        res = {}
        for skip in (True, False):
            w = WfStep(l["xcoords"], l["ycoords"], env_batch=B, model=dict(veer=veer) if veer else None,
                       kernel_choice=dict(one_block=fam, far_skip=skip))
            w.set_wind(wsx, wd)
            assert w.kernel_info()["one_block_kernel"] == 1
            d = w.step(yaw)
            w.sync()
            res[skip] = ({k: v.cpu().numpy() for k, v in d.items()}, w.risk_flags())
            w.close()
        (a, fa), (b, fb) = res[True], res[False]
        assert np.array_equal(fa, fb), fam
        n_diff = 0
        for k in a:
            same = a[k] == b[k]
            n_diff += int((~same).sum())
            ulp = np.abs(a[k].view(np.int32).astype(np.int64) - b[k].view(np.int32).astype(np.int64))
            # (std values near zero are differences of nearly equal numbers: held to an absolute 1e-9 m/s instead)
            ok = (ulp <= 2) | (np.abs(a[k].astype(np.float64) - b[k]) <= 1e-9)
            assert ok.all(), (fam, k, int((~ok).sum()), float(np.abs(a[k].astype(np.float64) - b[k]).max()))
        print(f"far skip on / off, {name} wd {wd} veer {veer} family {fam}: {n_diff} of {B * N * 7} output values differ")
        # measured on the round's final build (profiles/r04_far_skip_identity.txt): 0 of 1.1-1.3 M values in every case — a
        # skipped pair would add less than half an ulp to any sum that is not itself negligible — so the claim is held to that
        assert n_diff == 0, (fam, n_diff)


def test_per_farm_launch_order_follows_the_kernel_shape():
    """The geometry pass of a wind per farm lays out the tie flags and the direction-sorted launch order for ONE block
    shape of the on-the-fly kernel.  The shape changing afterwards — a grouped launch (wind series) that had moved the
    handle from the 128-farm blocks of G = 2 to G = 4, left again by the next wf_set_wind; a veer model switched on and
    off — must not leave the launch reading an order laid out for other blocks (round 4: farm slots past the end of the
    list read as farm 0, whose env state was then stepped once per such slot).  Fused env steps of a batch that is not
    a multiple of any block size, against the float32 transition of the reference and a handle configured in one go."""
    import sys

    import torch

    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    from fuzz_api import mdp_step_f32
    from wfcrl_env_amd.backend import WfStep

    rng = np.random.default_rng(5)
    N, B = 43, 1537
    x, y = rng.uniform(0, 4000, N), rng.uniform(0, 4000, N)
    envp = dict(yaw_lo=-30.0, yaw_hi=40.0, yaw_step=4.0, actuator_rate=0.3, dt=60.0, budget=0.02, load_coef=0.1, discrete=True)
    ws, wd = rng.uniform(4, 20, B), rng.uniform(0, 360, B)
    series = np.stack([rng.uniform(4, 20, 5), rng.uniform(0, 360, 5)], axis=1)
    for fam in ("2x2", "4x2", "8"):
        w = WfStep(x, y, env_batch=B, kernel_choice=dict(one_block=fam))
        w.env_config(**envp)
        w.set_wind_series(series, start=rng.integers(0, 5, B).astype(np.int32))  # direction groups: may change the block shape
        w.env_reset()
        w.env_step(rng.integers(0, 3, (B, N)).astype(np.float32))
        ref = WfStep(x, y, env_batch=B, kernel_choice=dict(one_block=fam))
        ref.env_config(**envp)
        for leg, veer in (("after a series", None), ("veer on", 3.0), ("veer off", 0.0)):
            if veer is not None:
                w.set_model(dict(veer=veer))
                ref.set_model(dict(veer=veer))
            w.set_wind(torch.from_numpy(ws).cuda(), torch.from_numpy(wd).cuda())  # (a veer toggle asks for the wind again)
            ref.set_wind(ws, wd)  # (the reference handle lays its geometry out after every change)
            st = w.env_get_state()
            ref.env_set_state(st)
            act = rng.integers(0, 3, (B, N)).astype(np.float32)
            got = w.env_step(torch.from_numpy(act).cuda())
            exp = ref.env_step(act)
            yaw_new = mdp_step_f32(st, act, envp)
            assert np.array_equal(got["yaw"].cpu().numpy(), yaw_new), (fam, leg)
            st2 = w.env_get_state()
            assert np.array_equal(st2["moves"], st["moves"] + 1), (fam, leg)
            for k in ("power", "wind_speed", "reward"):
                assert np.array_equal(got[k].cpu().numpy(), np.asarray(exp[k])), (fam, leg, k)
        w.close()
        ref.close()


def test_per_handle_kernel_calibration(layouts):
    """wf_kernel_choice::calibrate (default on): BEFORE its first table-path launch of a configuration the handle times the
    kernel families the rounds model prices close to its best guess on its own batch and keeps the fastest (csrc/
    wf_dispatch.hip: calibrate_families) — so every step of a handle comes from one family: step 1 == step 4 bit for bit.
    The process caches the result: a re-created handle does not time again and computes the same bits; wf_calibrate times
    again on request; wf_set_calibration replays a saved choice; a reconfiguration starts over; calibrate=False leaves the
    rounds model's guess alone.  Results stay inside the contract whichever family wins."""
    import torch

    from wfcrl_env_amd.backend import WfStep

    l = layouts["HornsRev1_"]
    N, B = 80, 24576
    rng = np.random.default_rng(77)
    yaw = torch.from_numpy(rng.uniform(-30, 30, (B, N)).astype(np.float32)).cuda()
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    w.set_wind(8.0, 270.0)
    assert w.calibration()["shape"] is None
    guess = w.kernel_info()
    out1 = {k: v.clone() for k, v in w.step(yaw).items()}
    cal = w.calibration()
    assert cal["shape"] is not None and len(cal["family_ms"]) >= 2, cal
    fastest = min(cal["family_ms"].values())  # the rounds model's guess stands within 4 % of the fastest (near-ties are not left to noise)
    guess_shape = f"{guess['lanes_per_env']}x{guess['slots_per_lane']}" if guess["one_block_kernel"] else "slot"
    assert cal["family_ms"][cal["shape"]] == fastest or (cal["shape"] == guess_shape and cal["family_ms"][cal["shape"]] <= 1.04 * fastest), (cal, guess)
    info = w.kernel_info()
    want = (16, 5) if cal["shape"] == "slot" else tuple(int(v) for v in cal["shape"].split("x"))
    assert (info["lanes_per_env"], info["slots_per_lane"]) == want, (info, cal, guess)
    for _ in range(2):
        w.step(yaw)
    out4 = w.step(yaw)
    assert w.calibration() == cal
    for k in out1:  # one family from the first launch on
        assert torch.equal(out1[k], out4[k]), k
    idx = rng.choice(B, 48, replace=False)
    ref = _oracle(l["xcoords"], l["ycoords"], 8.0, 270.0, yaw.cpu().numpy()[idx])
    flags = w.risk_flags()
    _check(dict({k: v.cpu().numpy()[idx] for k, v in out4.items()}, flags=flags[idx]), ref)
    # a re-created handle of the same configuration: the process-wide result, no timing, the same bits
    w2 = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    w2.set_wind(8.0, 270.0)
    o2 = w2.step(yaw)
    assert w2.calibration() == cal and w2.kernel_info() == info
    for k in out1:
        assert torch.equal(out1[k], o2[k]), k
    w2.calibrate()  # times again on request (scratch buffers, zero yaw); whatever it finds is a family of this layout
    cal2 = w2.calibration()
    assert cal2["shape"] in cal2["family_ms"] and len(cal2["family_ms"]) >= 2, cal2
    w2.close()
    # a saved calibration replayed: nothing is timed, the named family serves the handle — also one the timing did not pick
    for shape in (cal["shape"], "4x2", "slot"):
        w3 = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
        w3.set_calibration(shape=shape)
        w3.set_wind(8.0, 270.0)
        o3 = w3.step(yaw)
        c3, i3 = w3.calibration(), w3.kernel_info()
        assert c3["shape"] == shape, c3
        assert (i3["lanes_per_env"], i3["slots_per_lane"]) == ((16, 5) if shape == "slot" else tuple(int(v) for v in shape.split("x"))), i3
        if shape == cal["shape"]:
            for k in out1:
                assert torch.equal(out1[k], o3[k]), k
        _check(dict({k: v.cpu().numpy()[idx] for k, v in o3.items()}, flags=w3.risk_flags()[idx]), ref)
        w3.close()
    w.set_batch(4096 * 3)  # a reconfiguration starts over
    assert w.calibration()["shape"] is None
    w.close()
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B, kernel_choice=dict(calibrate=False))
    w.set_wind(8.0, 270.0)
    for _ in range(4):
        w.step(yaw)
    assert w.calibration()["shape"] is None and w.kernel_info() == guess
    w.close()


def test_slot_kernel_pick_on_the_table_path_does_not_leak_into_other_wind_regimes(layouts):
    """when the calibration keeps the register-slot kernel for the shared-wind table path, the handle's one-block
    shape stays what the rounds model chose — a later wind per farm still takes the on-the-fly one-block path, grouped
    launches still run the one-block kernel (rounds 3-4 zeroed the shape: neither did, until the batch was reconfigured)."""
    from wfcrl_env_amd.backend import WfStep

    l = layouts["HornsRev1_"]
    N, B = 80, 24576
    rng = np.random.default_rng(81)
    yaw = rng.uniform(-30, 30, (B, N)).astype(np.float32)
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    w.set_calibration(shape="slot")  # (what a timing could have found)
    w.set_wind(8.0, 270.0)
    out = w.step(yaw)
    assert w.kernel_info()["one_block_kernel"] == 0 and w.calibration()["shape"] == "slot"
    idx = rng.choice(B, 32, replace=False)
    _check(dict({k: v[idx] for k, v in out.items()}, flags=w.risk_flags()[idx]), _oracle(l["xcoords"], l["ycoords"], 8.0, 270.0, yaw[idx]))
    ws, wd = np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360
    w.set_wind(ws, wd)
    assert w.kernel_info()["one_block_kernel"] == 1 and w.kernel_info()["pair_table"] == 0
    out = w.step(yaw)
    _check(dict({k: v[idx] for k, v in out.items()}, flags=w.risk_flags()[idx]), _oracle(l["xcoords"], l["ycoords"], ws[idx], wd[idx], yaw[idx]))
    series = np.stack([rng.uniform(5, 14, 5), rng.uniform(200, 340, 5)], axis=1)
    w.set_wind_series(series, start=rng.integers(0, 5, B).astype(np.int32))
    ki = w.kernel_info()
    assert ki["direction_groups"] == 5 and ki["one_block_kernel"] == 1, ki
    w.close()


def test_on_the_fly_calibration(layouts):
    """A wind per farm (the reference's resets, mdp.py:237-258): before the first step there the handle times the one-block
    kernel of the table path's family against the register-slot kernel on its own batch and keeps the slot kernel only when
    it wins by 4 % (csrc/wf_dispatch.hip: calibrate_fly); the steps are inside the contract whichever runs, a forced family
    or calibrate=False is left alone, and a reconfiguration starts over."""
    from wfcrl_env_amd.backend import WfStep

    l = layouts["HornsRev1_"]
    N, B = 80, 16384
    rng = np.random.default_rng(78)
    yaw = rng.uniform(-30, 30, (B, N)).astype(np.float32)
    ws, wd = np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    w.set_wind(ws, wd)
    assert w.kernel_info()["one_block_kernel"] == 1 and w.calibration()["on_the_fly"] is None
    out = w.step(yaw)
    cal = w.calibration()
    ms = cal["on_the_fly_ms"]
    assert cal["on_the_fly"] in ("one_block", "slot") and set(ms) == {"one_block", "slot"}, cal
    assert cal["on_the_fly"] == ("slot" if ms["slot"] < 0.96 * ms["one_block"] else "one_block"), cal
    assert w.kernel_info()["one_block_kernel"] == (0 if cal["on_the_fly"] == "slot" else 1)
    idx = rng.choice(B, 48, replace=False)
    ref = _oracle(l["xcoords"], l["ycoords"], ws[idx], wd[idx], yaw[idx])
    _check(dict({k: v[idx] for k, v in out.items()}, flags=w.risk_flags()[idx]), ref)
    print(f"on the fly, HornsRev1 x {B}: {cal}")
    w.set_wind(ws[::-1].copy(), wd[::-1].copy())  # another draw of winds: the choice stands
    assert w.calibration()["on_the_fly"] == cal["on_the_fly"]
    w.set_batch(8192 * 3)  # a reconfiguration starts over
    assert w.calibration()["on_the_fly"] is None
    w.close()
    for choice in (dict(calibrate=False), dict(one_block="4x2")):
        w = WfStep(l["xcoords"], l["ycoords"], env_batch=B, kernel_choice=choice)
        w.set_wind(ws, wd)
        for _ in range(4):
            w.step(yaw)
        assert w.calibration()["on_the_fly"] is None and w.kernel_info()["one_block_kernel"] == 1
        w.close()


@pytest.mark.parametrize("per_farm", [False, True])
def test_calibration_inside_env_steps_leaves_no_trace(layouts, per_farm):
    """A handle that only ever sees fused env steps calibrates too — in its first step, with probes that carry no action (a
    solve at the current yaw state: no transition, no reward, no move counted).  Every step of the sequence, the calibrating
    one included, must advance the env state exactly once: yaw against the reference's float32 transition, `moves`, the
    actuation accumulator, and the outputs of the last step against the oracle."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    from fuzz_api import mdp_step_f32
    from wfcrl_env_amd.backend import WfStep

    l = layouts["HornsRev1_"]
    N, B = 80, 16384 + 128 if per_farm else 24576 + 256  # (batches no other test uses: the process cache holds nothing for them)
    rng = np.random.default_rng(79)
    envp = dict(yaw_lo=-40.0, yaw_hi=40.0, yaw_step=5.0, actuator_rate=0.3, dt=60.0, budget=0.1, load_coef=0.1, discrete=False)
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    w.env_config(**envp)
    if per_farm:
        ws, wd = np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360
    else:
        ws, wd = 8.0, 270.0
    w.set_wind(ws, wd)
    w.env_reset()
    key = "on_the_fly" if per_farm else "shape"
    for k in range(4):
        st = w.env_get_state()
        act = rng.uniform(-8, 8, (B, N)).astype(np.float32)
        got = w.env_step(act)
        assert np.array_equal(got["yaw"], mdp_step_f32(st, act, envp)), k
        st2 = w.env_get_state()
        assert np.array_equal(st2["moves"], st["moves"] + 1), k
        assert w.calibration()[key] is not None and len(w.calibration()["on_the_fly_ms" if per_farm else "family_ms"]) >= 2, (k, w.calibration())
    idx = rng.choice(B, 32, replace=False)
    ref = _oracle(l["xcoords"], l["ycoords"], ws[idx] if per_farm else ws, wd[idx] if per_farm else wd, got["yaw"][idx])
    _check(dict({k: np.asarray(got[k])[idx] for k in ("power", "wind_speed", "wind_direction", "load")}, flags=w.risk_flags()[idx]), ref)
    w.close()


def test_time_varying_direction_sweep_hornsrev2(layouts):
    """BASELINE config 5: wd(t) = 270 + 30 sin(2 pi t/200), shared and per-env (+U(-10,10))."""
    from oracle import c_oracle
    from wfcrl_env_amd.backend import WfStep

    l = layouts["HornsRev2_"]
    N, B = 91, 48
    rng = np.random.default_rng(1239)
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    yaw = np.zeros((B, N), np.float32)
    jitter = rng.uniform(-10, 10, B)
    for t in (0, 37, 50, 150):
        yaw = np.clip(yaw + rng.uniform(-5, 5, (B, N)), -40, 40).astype(np.float32)
        wd_t = 270 + 30 * np.sin(2 * np.pi * t / 200)
        for wd in (np.array([wd_t]), wd_t + jitter):
            ws = np.full_like(wd, 8.0)
            w.set_wind(ws, wd)
            out = _with_flags(w, w.step(yaw))
            ref = _oracle(l["xcoords"], l["ycoords"], ws, wd, yaw)
            _check(out, ref)
    w.close()


def test_error_behaviour(layouts):
    from wfcrl_env_amd.backend import WfStep

    l = layouts["Turb3_Row1_"]
    with pytest.raises(ValueError):
        WfStep(l["xcoords"], l["ycoords"][:2])
    with pytest.raises(ValueError):
        WfStep([0.0] * 300, [0.0] * 300)
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=4)
    with pytest.raises(ValueError, match="wf_set_wind"):
        w.step(np.zeros((4, 3), np.float32))
    with pytest.raises(ValueError):
        w.set_wind(np.array([8.0, 8.0]), np.array([270.0, 270.0]))  # count must be 1 or B
    with pytest.raises(ValueError):
        w.set_wind(-1.0, 270.0)
    with pytest.raises(ValueError, match="veer"):
        w.set_model({"veer": float("inf")})  # (a finite veer is served by the float64 kernel: tests/test_resolve_gpu.py)
    for bad, msg in [({"num_eps": 0.0}, "num_eps"), ({"ambient_ti": -0.1}, "turbulence_intensity"), ({"tsr": float("nan")}, "TSR"),
                     ({"hub_height": 60.0}, "hub_height"), ({"table_ws": [0.0, 5.0, 4.0], "table_ct": [0, 0.8, 0.5], "table_cp": [0, 0.4, 0.3]}, "ascending"),
                     ({"table_ws": [0.0, 5.0, 9.0], "table_ct": [0, -0.8, 0.5], "table_cp": [0, 0.4, 0.3]}, "non-negative")]:
        with pytest.raises(ValueError, match=msg):
            w.set_model(bad)
    w.set_wind(8.0, 270.0)
    assert np.isfinite(w.step(np.zeros((4, 3), np.float32))["power"]).all()  # the handle is still usable after rejected models
    w.close()


def test_custom_model_table_is_data(layouts):
    """The power/thrust table and model constants are data (wf_set_model), not code."""
    from oracle import c_oracle
    from oracle.floris_gch_numpy import ModelParams
    from wfcrl_env_amd.backend import WfStep, default_model

    l = layouts["Ablaincourt_"]
    m = default_model()
    ws_tab = [0.0, 3.0, 6.0, 9.0, 12.0, 25.0, 25.5]
    ct_tab = [0.0, 0.9, 0.85, 0.75, 0.5, 0.1, 0.0]
    cp_tab = [0.0, 0.2, 0.42, 0.45, 0.4, 0.05, 0.0]
    custom = dict(ambient_ti=0.08, shear=0.14, rotor_diameter=120.0, hub_height=85.0, tsr=7.5, ka=0.3, kb=0.005,
                  alpha=0.6, beta=0.08, table_ws=ws_tab, table_ct=ct_tab, table_cp=cp_tab)
    m.update(custom)
    rng = np.random.default_rng(9)
    yaw = rng.uniform(-30, 30, (200, 7)).astype(np.float32)
    ws = np.clip(8 * rng.weibull(8, 200), 3, 28)
    wd = rng.normal(270, 20, 200) % 360
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=200, model=custom)
    w.set_wind(ws, wd)
    out = _with_flags(w, w.step(yaw))
    w.close()
    p = ModelParams(ambient_ti=0.08, shear=0.14, D=120.0, HH=85.0, TSR=7.5, ka=0.3, kb=0.005, alpha=0.6, beta=0.08,
                    table_ws=ws_tab, table_ct=ct_tab, table_cp=cp_tab)
    ref = _oracle(l["xcoords"], l["ycoords"], ws, wd, yaw, p)
    _check(out, ref)


def test_low_hub_model_uses_general_mirror_core_kernel(layouts):
    """With a low hub the ground-mirror vortices sit close to the rotor: more than one mirror offset has a core
    factor != 1.0f, which selects the MC1 = false instantiation of the step kernel."""
    from oracle import c_oracle
    from oracle.floris_gch_numpy import ModelParams
    from wfcrl_env_amd.backend import WfStep

    l = layouts["Turb16_Row5_"]
    rng = np.random.default_rng(21)
    B = 300
    yaw = rng.uniform(-35, 35, (B, 16)).astype(np.float32)
    ws = np.clip(8 * rng.weibull(8, B), 3, 28)
    wd = rng.normal(270, 25, B) % 360
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B, model=dict(hub_height=70.0))
    w.set_wind(ws, wd)
    out = _with_flags(w, w.step(yaw))
    w.close()
    ref = _oracle(l["xcoords"], l["ycoords"], ws, wd, yaw, ModelParams(HH=70.0))
    _check(out, ref)
    # the flag itself: exp(-(2*70 - 94.5 + 0.001)^2 / 25.2^2) = 3.8e-2 and the next class 6e-5 are both > 2^-25
    import math
    assert math.exp(-((2 * 70 - 63 + 0.001) ** 2) / 25.2**2) > 2.9e-8


def test_interface_from_yaml_on_gpu(layouts, tmp_path):
    """The reference's constructor path FlorisInterface(num_turbines, simul_file) via a FLORIS case.yaml."""
    import yaml

    from oracle import c_oracle
    from oracle.floris_gch_numpy import ModelParams
    from wfcrl_env_amd.environments.data_cases import named_cases_dictionary
    from wfcrl_env_amd.interface import HipFlorisInterface
    from wfcrl_env_amd.simul_utils import case_config

    case = named_cases_dictionary["Ablaincourt_"][1]
    cfg = case_config(case.dict())
    cfg["flow_field"].update(turbulence_intensity=0.09, wind_shear=0.15, wind_speeds=[9.5], wind_directions=[255.0])
    path = tmp_path / "case.yaml"
    path.write_text(yaml.safe_dump(cfg))
    it = HipFlorisInterface.from_yaml(str(path), max_iter=10)
    assert (it.wind_speed, it.wind_dir, it.num_turbines) == (9.5, 255.0, 7)
    yaw = np.array([20.0, -10.0, 5.0, 0.0, 15.0, -25.0, 0.0])
    it.update_command(yaw)
    ref = _oracle(case.xcoords, case.ycoords, 9.5, 255.0, yaw[None], ModelParams(ambient_ti=0.09, shear=0.15))
    assert np.abs(it.avg_powers() / ref["power"][0] - 1).max() < 1e-4
    assert np.abs(it.get_measure("wind_direction") - ref["wind_direction"][0]).max() < 2e-4
    assert np.abs(it.get_measure("load") / 1e7 - ref["load"][0]).max() < 1e-4


def test_random_layouts_including_degenerate_ones():
    """Seeded random layouts: every kernel variant boundary (N = 4/5, 8/9, 12/13, 16/17, 32/33, 64/65, 80/81, 96/97),
    exactly aligned rows/columns, tight (1.5 D) and very wide spacing; plus unphysically packed layouts (rotors
    metres apart, co-located turbines) where the model divides by a rotor-averaged speed near zero: finite
    outputs are required there, parity is not meaningful."""
    from oracle import c_oracle

    rng = np.random.default_rng(2024)
    cases = []
    def spaced(n, span, dmin=1.5 * 126.0):
        pts = []
        while len(pts) < n:  # rejection sampling: no two rotors closer than 1.5 D
            p = np.array([rng.uniform(0, span), rng.uniform(-span / 3, span / 3)])
            if all(np.hypot(*(p - q)) >= dmin for q in pts):
                pts.append(p)
        pts = np.array(pts)
        return pts[:, 0], pts[:, 1]

    for n in (4, 5, 8, 9, 12, 13, 16, 17, 24, 25, 32, 33, 48, 49, 64, 65, 81, 96, 97, 100):
        cases.append(spaced(n, float(rng.choice([3000.0, 6000.0, 12000.0]))))
    g = np.arange(12)
    cases.append(((g % 4) * 630.0, (g // 4) * 378.0))                       # exact grid (ties at 270 and at 0/90/180)
    cases.append((np.arange(6) * 126.0 * 1.5, np.zeros(6)))                 # 1.5 D spacing: deep near-wake
    for x, y in cases:
        n = len(x)
        B = 24
        yaw = rng.uniform(-40, 40, (B, n)).astype(np.float32)
        ws = np.clip(8 * rng.weibull(8, B), 3, 28)
        wd = rng.uniform(0, 360, B)
        wd[:4] = [270.0, 0.0, 90.0, 180.0]
        out, info = _step(x, y, ws, wd, yaw)
        ref = _oracle(x, y, ws, wd, yaw)
        for v in out.values():
            assert np.isfinite(v).all(), n
        # directions that are exact multiples of 90 deg other than 270 make x' ties depend on 1e-13 rounding noise
        # of the rotation (also inside FLORIS itself): compare those farms on power only, loosely
        exact = slice(4, None)
        _check({k: v[exact] for k, v in out.items()}, {k: v[exact] for k, v in ref.items()}, max_flagged_frac=0.15)
        p = np.abs(out["power"][:4] - ref["power"][:4]) / np.maximum(ref["power"][:4], 1e3)
        assert p.max() < 5e-3, (n, p.max())
    packed = [(rng.uniform(0, 600, 13), rng.uniform(-200, 200, 13)),
              (np.array([0.0, 0.0, 500.0, 500.0]), np.array([0.0, 0.0, 100.0, 100.0]))]
    for x, y in packed:
        yaw = rng.uniform(-40, 40, (64, len(x))).astype(np.float32)
        out, _ = _step(x, y, np.clip(8 * rng.weibull(8, 64), 3, 28), rng.uniform(0, 360, 64), yaw)
        for v in out.values():
            assert np.isfinite(v).all()


def test_shared_wind_pair_table_path_and_its_fallbacks(layouts, monkeypatch):
    """Shared wind uses the pair-coefficient table (float64 precompute + LDS-DMA rows); per-farm wind, N > 128 and
    WF_NO_PAIR_TABLE use the on-the-fly transverse pass.  Both must agree with the oracle and with each other."""
    from oracle import c_oracle
    from oracle.floris_gch_numpy import ModelParams
    from wfcrl_env_amd.backend import WfStep

    rng = np.random.default_rng(77)
    l = layouts["HornsRev1_"]
    B = 64
    yaw = rng.uniform(-40, 40, (B, 80)).astype(np.float32)
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    w.set_wind(9.0, 281.0)
    assert w.kernel_info()["pair_table"] == 1
    a = _with_flags(w, w.step(yaw))
    # the same wind given per farm, as two spellings of the direction so that it does not register as shared:
    # on-the-fly path
    w.set_wind(np.full(B, 9.0), 281.0 + 360.0 * (np.arange(B) % 2))
    assert w.kernel_info()["pair_table"] == 0
    b = _with_flags(w, w.step(yaw))
    # one direction, a speed per farm (host arrays): geometry and pair table are shared, the speed is not
    ws_b = rng.uniform(5, 14, B)
    w.set_wind(ws_b, np.full(B, 281.0))
    assert w.kernel_info()["pair_table"] == 1
    cdir = _with_flags(w, w.step(yaw))
    w.close()
    _check(cdir, _oracle(l["xcoords"], l["ycoords"], ws_b, np.full(B, 281.0), yaw))
    ref = _oracle(l["xcoords"], l["ycoords"], 9.0, 281.0, yaw)
    _check(a, ref)
    _check(b, ref)
    both = (a["flags"] == 0) & (b["flags"] == 0)  # the two paths round differently: compare where neither is at risk
    assert np.abs(a["power"] / np.maximum(b["power"], 1e3) - 1)[both][b["power"][both] > 1e3].max() < 2e-5
    assert np.abs(a["load"] - b["load"])[both].max() < 2e-5
    # per-handle switch for A/B runs (wf_set_kernel_choice); the environment variable of earlier builds seeds it at wf_create
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B, kernel_choice=dict(pair_table=False))
    w.set_wind(9.0, 281.0)
    assert w.kernel_info()["pair_table"] == 0 and w.kernel_choice()["pair_table"] == 0
    w.close()
    monkeypatch.setenv("WF_NO_PAIR_TABLE", "1")
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    w.set_wind(9.0, 281.0)
    assert w.kernel_info()["pair_table"] == 0
    w.close()
    monkeypatch.delenv("WF_NO_PAIR_TABLE")
    # N > WF_PAIR_MAX_N with shared wind: no table
    x = (np.arange(130) % 13) * 700.0 + rng.uniform(-30, 30, 130)
    y = (np.arange(130) // 13) * 650.0 + rng.uniform(-30, 30, 130)
    yaw = rng.uniform(-30, 30, (6, 130)).astype(np.float32)
    w = WfStep(x, y, env_batch=6)
    w.set_wind(8.0, 265.0)
    assert w.kernel_info()["pair_table"] == 0
    out = _with_flags(w, w.step(yaw))
    w.close()
    _check(out, _oracle(x, y, 8.0, 265.0, yaw))
    # low hub (general mirror cores) through the table
    l16 = layouts["Turb16_Row5_"]
    yaw = rng.uniform(-35, 35, (40, 16)).astype(np.float32)
    w = WfStep(l16["xcoords"], l16["ycoords"], env_batch=40, model=dict(hub_height=70.0))
    w.set_wind(7.5, 270.0)
    assert w.kernel_info()["pair_table"] == 1
    out = _with_flags(w, w.step(yaw))
    # a model change invalidates the table
    w.set_model(dict(hub_height=90.0))
    out90 = _with_flags(w, w.step(yaw))
    w.close()
    _check(out, _oracle(l16["xcoords"], l16["ycoords"], 7.5, 270.0, yaw, ModelParams(HH=70.0)))
    _check(out90, _oracle(l16["xcoords"], l16["ycoords"], 7.5, 270.0, yaw))


_VARIANTS = [(g, s) for g, smax in ((4, 4), (8, 4), (16, 6), (32, 4), (64, 4)) for s in range(1, smax + 1)]


@pytest.mark.parametrize("G,S", _VARIANTS)
def test_every_kernel_variant_matches_the_oracle(G, S, monkeypatch):
    """Every (lanes per farm, slots per lane) instantiation — including the ones only large batches pick and the
    register-capped S <= 3 ones — on a ragged farm that does not fill the variant, on both transverse-pass paths."""
    import zlib

    from oracle import c_oracle
    from wfcrl_env_amd.backend import WfStep

    N = min(G * S, 256) - (1 if G * S > 1 else 0)
    rng = np.random.default_rng(zlib.crc32(f"variant/{G}x{S}".encode()))
    cols = 8
    x = (np.arange(N) // cols) * 700.0 + rng.uniform(-60, 60, N)
    y = (np.arange(N) % cols) * 560.0 + rng.uniform(-60, 60, N)
    B = 6
    yaw = rng.uniform(-30, 30, (B, N)).astype(np.float32)
    w = WfStep(x, y, env_batch=B, kernel_choice=dict(slot=(G, S)))
    info = w.kernel_info()
    assert (info["lanes_per_env"], info["slots_per_lane"]) == (G, S)
    w.set_wind(9.5, 263.0)
    shared = _with_flags(w, w.step(yaw))
    tab = w.kernel_info()["pair_table"]
    ws, wd = rng.uniform(6, 12, B), rng.uniform(250, 290, B)
    w.set_wind(ws, wd)
    assert w.kernel_info()["pair_table"] == 0
    per_farm = _with_flags(w, w.step(yaw))
    w.close()
    assert tab in (0, 1)  # the table path exists where its LDS slab fits (wf_kernels.hip: tab_fits)
    _check(shared, _oracle(x, y, 9.5, 263.0, yaw))
    _check(per_farm, _oracle(x, y, ws, wd, yaw))


@pytest.mark.parametrize("wdir", [270.0, 90.0, 0.0])
@pytest.mark.parametrize("gs,shape", [("4x4", (5, 3)), ("4x3", (3, 4)), ("8x2", (4, 4)), ("8x4", (4, 8)), ("16x2", (2, 12)),
                                      ("4x4", (1, 15)), ("16x5", (10, 8))])
def test_exact_x_ties_across_kernel_blocks(gs, shape, wdir, monkeypatch):
    """Axis-aligned grids at wd = 270 have exact x' ties: tied turbines exchange transverse
    velocities in both directions, also when the tie group straddles two (or more) lane-group blocks of the kernel,
    whose earlier block has been written out by the time the later sources run.  Both transverse-pass paths."""
    from oracle import c_oracle
    from wfcrl_env_amd.backend import WfStep

    # 5 D x 4 D grid: besides the ties, every third column sits exactly 15 D downstream — ON the reach threshold of the
    # wake-added TI, which FLORIS tests as x_t <= x_i + 15 D in float64: at wd = 90 / 0 the rounding of the rotation
    # (sin(pi) = 1.2e-16) decides it, and the kernel has to decide the same way
    ncol, nrow = shape  # ncol columns along the wind, nrow tied turbines per column
    x = np.repeat(np.arange(ncol) * 630.0, nrow)
    y = np.tile(np.arange(nrow) * 504.0, ncol)
    if wdir == 0.0:
        x, y = y, x
    N = x.size
    rng = np.random.default_rng(N * 131 + ncol)
    B = 5
    yaw = rng.uniform(-30, 30, (B, N)).astype(np.float32)
    w = WfStep(x, y, env_batch=B, kernel_choice=dict(slot=gs))
    info = w.kernel_info()
    assert f'{info["lanes_per_env"]}x{info["slots_per_lane"]}' == gs
    w.set_wind(8.0, wdir)
    shared = _with_flags(w, w.step(yaw))
    w.set_wind(np.full(B, 8.0), wdir + 360.0 * (np.arange(B) % 2))  # two spellings: not taken for a shared direction
    assert w.kernel_info()["pair_table"] == 0
    per_farm = _with_flags(w, w.step(yaw))
    w.close()
    ref = _oracle(x, y, 8.0, wdir, yaw)
    _check(shared, ref)
    _check(per_farm, ref)


def test_randomised_parity_fuzz_sample():
    """A fixed-seed sample of tests/tools/fuzz_parity.py (random regular / jittered / holed grids and clouds, axis-aligned and
    random wind directions, every kernel variant, shared and per-farm wind, default and non-default models): no run
    outside the parity tolerances except the bounded signature of a threshold flip, and few of those."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tests", "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    nflip, nbad = fz.run(120, 2024)
    assert nbad == 0  # every mismatch on a farm the kernel flagged itself, within the bounded signature
    assert nflip <= 3


def test_randomised_api_sequence_fuzz_sample():
    """A fixed-seed sample of tests/tools/fuzz_api.py: random sequences of layout / batch / model / wind (shared, per farm,
    device pointers, device sampling, series playback) / env calls on one handle, every step and fused env step
    checked against the oracle on the state the sequence should have produced (stale geometry, pair table, kernel
    variant or env state would show)."""
    import importlib.util
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    spec = importlib.util.spec_from_file_location("fuzz_api", os.path.join(ROOT, "tests", "tools", "fuzz_api.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    nflip, nbad = fz.run(14, 40, 7)
    assert nbad == 0


def test_integration_md_stub_runs_as_printed():
    """The ctypes binding printed in INTEGRATION.md section 2 (what a maintainer of the reference would paste) is extracted
    from the markdown and run against the reference's known-answer vector — and, as printed (it calls nothing but wf_create
    ... wf_step: the library's own default mode), against a farm whose overlap count float32 cannot decide: no flag is
    left and the result is the float64 oracle's (tests/golden/regime_cases.npz::overlap_flip)."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k != "WF_RISK_RESOLVE"}  # (this file's fixture: not for the stub)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "integration_stub_check.py")],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "stub: ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("forced", ["", "2x2", "4x2", "8", "4"])
def test_direction_groups_series_binned_and_explicit_shared_direction(layouts, forced, monkeypatch):
    """(every one-block kernel family forced in turn: a grouped launch lays its farm lists out in blocks of the kernel that
    will run them — the G = 2 kernel's 128-farm blocks are never used for it, also when WF_LL_G forces that kernel for
    the plain batch: found by the API fuzzer)
    Per-farm wind that is not really per-farm stays on the pair-table path:
    (a) a shared wind SERIES has only T distinct winds: farms are grouped by start row, one geometry + table per row
        (reference wfcrl/interface.py:503-524 playback), checked at every tick against the oracle and against the
        on-the-fly path of the same handle state;
    (b) device arrays with ONE direction and a speed per farm (wf_set_wind_counts: never read back, so stated explicitly);
    (c) binned reset sampling (build-defined): K grid directions, groups cached across resets."""
    import torch

    from wfcrl_env_amd.backend import WfStep

    choice = dict(one_block=forced) if forced else None
    l = layouts["HornsRev1_"]
    N, B, T = 80, 700, 7
    rng = np.random.default_rng(2025)
    x, y = l["xcoords"], l["ycoords"]
    # (a) series
    series = np.stack([rng.uniform(5, 14, T), rng.uniform(200, 340, T)], axis=1)
    start = rng.integers(0, T, B).astype(np.int32)
    start[:3] = [0, T - 1, T - 1]
    w = WfStep(x, y, env_batch=B, kernel_choice=choice)
    w.set_wind_series(series, start=start)
    info = w.kernel_info()
    assert info["pair_table"] == 1 and info["direction_groups"] == T and info["grid_blocks"] >= B // info["envs_per_block"]
    for t in range(T):
        if t:
            w.wind_series_step()
        ws, wd = w.get_wind()
        assert np.array_equal(ws, series[(start + t) % T, 0]) and np.array_equal(wd, series[(start + t) % T, 1])
        yaw = rng.uniform(-35, 35, (B, N)).astype(np.float32)
        got = _with_flags(w, w.step(yaw))
        _check(got, _oracle(x, y, ws, wd, yaw))
    with pytest.raises(ValueError, match="exhausted"):
        w.wind_series_step()
    # too many rows for the batch: falls back to a geometry per farm, same results contract
    w.set_batch(40)
    w.env_batch = 40
    series2 = np.stack([rng.uniform(5, 14, 30), rng.uniform(200, 340, 30)], axis=1)
    w.set_wind_series(series2, seed=5)
    assert w.kernel_info()["direction_groups"] == 0 and w.kernel_info()["pair_table"] == 0
    ws, wd = w.get_wind()
    yaw = rng.uniform(-35, 35, (40, N)).astype(np.float32)
    _check(_with_flags(w, w.step(yaw)), _oracle(x, y, ws, wd, yaw))
    # device-generated starts take the grouped path too
    w.set_batch(B)
    w.env_batch = B
    w.set_wind_series(series, seed=9)
    assert w.kernel_info()["direction_groups"] == T
    w.wind_series_step()
    ws, wd = w.get_wind()
    yaw = rng.uniform(-35, 35, (B, N)).astype(np.float32)
    _check(_with_flags(w, w.step(yaw)), _oracle(x, y, ws, wd, yaw))

    # (b) one direction on the device, a speed per farm
    ws_b = rng.uniform(5, 14, B)
    w.set_wind(torch.from_numpy(ws_b).cuda(), torch.tensor([281.0], dtype=torch.float64, device="cuda"))
    assert w.kernel_info()["pair_table"] == 1 and w.kernel_info()["direction_groups"] == 0
    got = _with_flags(w, w.step(yaw))
    _check(got, _oracle(x, y, ws_b, np.full(B, 281.0), yaw))
    ws2, wd2 = w.get_wind()
    assert np.array_equal(ws2, ws_b) and np.array_equal(wd2, np.full(B, 281.0))
    # per-farm device directions stay on the fly
    w.set_wind(torch.from_numpy(ws_b).cuda(), torch.from_numpy(np.full(B, 281.0)).cuda())
    assert w.kernel_info()["pair_table"] == 0
    got2 = _with_flags(w, w.step(yaw))
    both = (got["flags"] == 0) & (got2["flags"] == 0)
    assert np.abs(got["power"] / np.maximum(got2["power"], 1e3) - 1)[both].max() < 2e-5

    # (c) binned reset directions
    w.sample_wind(17, direction_step=5.0)
    info = w.kernel_info()
    if forced in ("8", "4", "4x2", "2x2"):  # 72 groups padded to blocks of 32 / 64 farms at 700 farms: does not pay,
        assert info["direction_groups"] == 0 and info["pair_table"] == 0  # the on-the-fly path serves (by design)
        w.close()
        return
    assert info["direction_groups"] == 72 and info["pair_table"] == 1
    ws, wd = w.get_wind()
    assert np.all(np.abs(wd / 5.0 - np.round(wd / 5.0)) < 1e-12) and len(np.unique(wd)) > 10 and ws.std() > 0.5
    _check(_with_flags(w, w.step(yaw)), _oracle(x, y, ws, wd, yaw))
    w.sample_wind(18, direction_step=5.0)  # next reset: cached geometry / tables, new grouping
    ws3, wd3 = w.get_wind()
    assert not np.array_equal(wd3, wd)
    _check(_with_flags(w, w.step(yaw)), _oracle(x, y, ws3, wd3, yaw))
    w.set_model(dict(ambient_ti=0.08))  # invalidates the cached tables, not the grouping
    from oracle.floris_gch_numpy import ModelParams

    _check(_with_flags(w, w.step(yaw)), _oracle(x, y, ws3, wd3, yaw, ModelParams(ambient_ti=0.08)))
    w.sample_wind(19)  # continuous directions again
    assert w.kernel_info()["direction_groups"] == 0 and w.kernel_info()["pair_table"] == 0
    w.close()


@pytest.mark.parametrize("model", [
    dict(enable_secondary_steering=False),
    dict(enable_yaw_added_recovery=False),
    dict(enable_transverse_velocities=False),
    dict(enable_secondary_steering=False, enable_yaw_added_recovery=False, enable_transverse_velocities=False),
    dict(defl_alpha=0.4, defl_beta=0.1, defl_ka=0.3, defl_kb=0.006),
    dict(alpha=0.5, ka=0.45, defl_alpha=0.58, defl_ka=0.38, enable_yaw_added_recovery=False),
])
def test_model_surface_of_the_case_yaml(layouts, model):
    """What the reference's case.yaml can express beyond its template values (case.yaml:46-59, 76-80): the three solver
    switches and separate gauss parameter sets for the deflection and the velocity model — on the pair-table path and on
    the fly, against the oracle with the same switches."""
    from oracle.floris_gch_numpy import ModelParams
    from wfcrl_env_amd.backend import WfStep

    l = layouts["Turb_TCRWP_"]
    rng = np.random.default_rng(len(str(model)))
    B, N = 300, 32
    yaw = rng.uniform(-35, 35, (B, N)).astype(np.float32)
    mp = ModelParams(**model)
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B, model=model)
    w.set_wind(9.0, 275.0)
    assert w.kernel_info()["pair_table"] == 1
    shared = _with_flags(w, w.step(yaw))
    _check(shared, _oracle(l["xcoords"], l["ycoords"], 9.0, 275.0, yaw, mp))
    if not model.get("enable_transverse_velocities", True):
        assert np.abs(shared["load"][..., 2:]).max() == 0.0  # V = W = 0: std v = std w = 0
        assert np.abs(shared["wind_direction"] - 275.0).max() < 1e-4
    ws, wd = rng.uniform(5, 14, B), rng.uniform(0, 360, B)
    w.set_wind(ws, wd)
    assert w.kernel_info()["pair_table"] == 0
    _check(_with_flags(w, w.step(yaw)), _oracle(l["xcoords"], l["ycoords"], ws, wd, yaw, mp))
    with pytest.raises(ValueError, match="unknown model parameter"):
        w.set_model(dict(enable_everything=True))
    w.close()


@pytest.mark.parametrize("G", ["4", "8", "16", "4x2", "2x2"])
@pytest.mark.parametrize("name,wdir", [("HornsRev1_", 270.0), ("HornsRev2_", 243.0), ("Turb_TCRWP_", 281.0), ("Ormonde_", 200.0)])
def test_one_block_at_a_time_kernel(layouts, name, wdir, G, monkeypatch):
    """wf_step_ll_kernel (csrc/wf_kernels_ll.hip: one target block in registers, earlier sources replayed from the source
    log) at every lane-group width, against the oracle and against wf_step_kernel on the same inputs: shared wind, one
    direction with a speed per farm, ragged batches, the fused env step."""
    from wfcrl_env_amd.backend import WfStep

    l = layouts[name]
    N = l["num_turbines"]
    lanes, slots = (int(v) for v in (G + "x1").split("x")[:2])
    if N <= lanes * slots:
        pytest.skip("single block")
    rng = np.random.default_rng(N * 10 + lanes + slots)
    B = 133  # not a multiple of the farms per block
    yaw = rng.uniform(-35, 35, (B, N)).astype(np.float32)
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B, kernel_choice=dict(one_block=G))
    w.set_wind(8.5, wdir)
    info = w.kernel_info()
    assert info["one_block_kernel"] == 1 and (info["lanes_per_env"], info["slots_per_lane"]) == (lanes, slots) and info["pair_table"] == 1
    a = _with_flags(w, w.step(yaw))
    _check(a, _oracle(l["xcoords"], l["ycoords"], 8.5, wdir, yaw))
    ws = rng.uniform(4, 16, B)
    w.set_wind(ws, np.full(B, wdir))
    b = _with_flags(w, w.step(yaw))
    _check(b, _oracle(l["xcoords"], l["ycoords"], ws, np.full(B, wdir), yaw))
    # fused env step through the same kernel
    w.env_config(load_coef=0.2)
    w.env_reset()
    act = rng.uniform(-5, 5, (B, N)).astype(np.float32)
    e = w.env_step(act)
    assert np.array_equal(e["yaw"], act)
    ref = _oracle(l["xcoords"], l["ycoords"], ws, np.full(B, wdir), act)
    r_ref = (ref["power"] / 1e6 * 1e3 / ws[:, None] ** 3).mean(axis=1) - 0.2 * np.abs(ref["load"]).reshape(B, -1).mean(axis=1)
    ok = w.risk_flags() == 0
    assert np.abs(e["reward"] - r_ref)[ok].max() < 5e-5 * np.abs(r_ref).max()
    w.close()
    # the register-slot kernel on the same inputs
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B, kernel_choice=dict(one_block=False))
    w.set_wind(8.5, wdir)
    assert w.kernel_info()["one_block_kernel"] == 0
    a0 = _with_flags(w, w.step(yaw))
    w.close()
    both = (a["flags"] == 0) & (a0["flags"] == 0)
    assert (np.abs(a["power"] - a0["power"]) / np.maximum(a0["power"], 1e3))[both].max() < 2e-5
    assert np.abs(a["wind_direction"] - a0["wind_direction"])[both].max() < 2e-4


@pytest.mark.parametrize("G", ["4", "8", "4x2", "2x2"])
def test_one_block_kernel_hands_cross_block_ties_back(layouts, G, monkeypatch):
    """Axis-aligned grids at wd = 270 have exact x' ties that straddle lane-group blocks: the device-side flag of the
    target-block table routes such a direction to wf_step_kernel (no host round trip); other directions of the same
    handle take the one-block kernel.  Same results contract either way."""
    from wfcrl_env_amd.backend import WfStep

    l = layouts["Turb32_Row5_"]
    rng = np.random.default_rng(len(G))
    B = 64
    yaw = rng.uniform(-30, 30, (B, 32)).astype(np.float32)
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B, kernel_choice=dict(one_block=G))
    for wdir in (270.0, 263.0, 270.0):
        w.set_wind(8.0, wdir)
        assert w.kernel_info()["one_block_kernel"] == 1
        _check(_with_flags(w, w.step(yaw)), _oracle(l["xcoords"], l["ycoords"], 8.0, wdir, yaw))
    w.close()


@pytest.mark.parametrize("G", ["4x2", "8", "4", "2x2"])
@pytest.mark.parametrize("name", ["HornsRev1_", "Turb32_Row5_", "Ormonde_"])
def test_one_block_kernel_on_the_fly_with_a_wind_per_farm(layouts, name, G, monkeypatch):
    """A wind per farm on the one-block kernel (transverse pass on the fly from each farm's own sorted geometry), against
    the oracle and against wf_step_kernel.  Some farms get wd = 270 exactly: on the grid layout their geometry has x' ties
    across the kernel's blocks, and those farms — only those — are served by wf_step_kernel behind it (per-farm device
    flags); the fused env step must advance every farm's state exactly once either way."""
    from wfcrl_env_amd.backend import WfStep

    l = layouts[name]
    N = l["num_turbines"]
    rng = np.random.default_rng(N + len(G))
    B = 150
    yaw = rng.uniform(-35, 35, (B, N)).astype(np.float32)
    ws = rng.uniform(4, 16, B)
    wd = rng.uniform(0, 360, B)
    wd[::7] = 270.0
    wd[3] = 90.0
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B, kernel_choice=dict(one_block=G))
    w.set_wind(ws, wd)
    info = w.kernel_info()
    assert info["one_block_kernel"] == 1 and info["pair_table"] == 0
    ref = _oracle(l["xcoords"], l["ycoords"], ws, wd, yaw)
    a = _with_flags(w, w.step(yaw))
    _check(a, ref)
    import torch

    w.set_wind(torch.from_numpy(ws).cuda(), torch.from_numpy(wd).cuda())  # device arrays: nothing is read back
    b = _with_flags(w, {k: v.cpu().numpy() for k, v in w.step(torch.from_numpy(yaw).cuda()).items()})
    for k in ("power", "wind_speed", "wind_direction", "load"):
        assert np.array_equal(a[k], b[k]), k
    w.env_config(load_coef=0.2, budget=float("inf"))  # no actuation gate: the transition alone is under test here
    w.env_reset()
    act = rng.uniform(-5, 5, (B, N)).astype(np.float32)
    e1 = w.env_step(act)
    e2 = w.env_step(act)
    assert np.array_equal(e1["yaw"], act) and np.array_equal(e2["yaw"], np.clip(2 * act, -40, 40).astype(np.float32))
    st = w.env_get_state()
    assert np.array_equal(st["moves"], np.full(B, 2)) and np.allclose(st["acc"], 2 * np.abs(act), rtol=1e-6)
    # the table path and the on-the-fly path of one handle share the source log; at G = 4 they use different block
    # sizes (one slot per lane on the table path, two on the fly): alternate them
    lanes = int(G.split("x")[0])
    k = w.kernel_info()  # on the fly: two slots at G = 4 whatever the table path uses; G = 2 x 2 as on the table path
    assert (k["lanes_per_env"], k["slots_per_lane"]) == ((lanes, 2) if lanes <= 4 else (lanes, 1))
    w.set_wind(9.0, 281.0)
    assert w.kernel_info()["pair_table"] == 1 and w.kernel_info()["one_block_kernel"] == (1 if N > lanes * (2 if "x2" in G else 1) else 0)
    _check(_with_flags(w, w.step(yaw)), _oracle(l["xcoords"], l["ycoords"], 9.0, 281.0, yaw))
    w.set_wind(ws, wd)
    c2 = _with_flags(w, w.step(yaw))
    for k in ("power", "wind_speed", "wind_direction", "load"):
        assert np.array_equal(a[k], c2[k]), k
    w.close()
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B, kernel_choice=dict(one_block=G, fly_one_block=False))
    w.set_wind(ws, wd)
    assert w.kernel_info()["one_block_kernel"] == 0
    a0 = _with_flags(w, w.step(yaw))
    w.close()
    both = (a["flags"] == 0) & (a0["flags"] == 0)
    assert (np.abs(a["power"] - a0["power"]) / np.maximum(a0["power"], 1e3))[both].max() < 2e-5


def _regime(name):
    d = np.load(os.path.join(ROOT, "tests", "golden", "regime_cases.npz"))
    inp = {k: d[f"{name}_{k}"] for k in ("x", "y", "ws", "wd", "yaw")}
    ref = {k[len(name) + 5:]: d[k] for k in d.files if k.startswith(name + "_ref_")}
    return inp, ref


@pytest.mark.parametrize("kernel", ["", "8", "4x2", "4", "2x2"])
def test_regime_cases_are_flagged_and_bounded(kernel, monkeypatch):
    """The two fuzzer-found regime farms (tests/golden/make_regime_cases.py) on every kernel family: the farm on the
    cut-in ramp of the thrust table raises WF_RISK_THRUST_RAMP, the farm 3.7e-7 from the overlap threshold raises
    WF_RISK_OVERLAP, both stay inside the bounded signature; at 9 m/s neither flag is raised and parity is strict."""
    import parity
    from wfcrl_env_amd.backend import WfStep

    for name, bit in (("thrust_ramp", parity.RISK_THRUST_RAMP), ("overlap_flip", parity.RISK_OVERLAP)):
        i, ref = _regime(name)
        B = 64
        w = WfStep(i["x"], i["y"], env_batch=B, kernel_choice=dict(one_block=kernel) if kernel else None)
        for per_farm in (False, True):  # pair-table path / on-the-fly path
            ws, wd = (np.repeat(i["ws"], B), np.repeat(i["wd"], B)) if per_farm else (float(i["ws"][0]), float(i["wd"][0]))
            w.set_wind(ws, wd)
            out = w.step(np.repeat(i["yaw"], B, axis=0).astype(np.float32))
            fl = w.risk_flags()
            assert (fl & bit).all(), (name, per_farm, fl[:4])
            refB = {k: np.repeat(v, B, axis=0) for k, v in ref.items()}
            s = parity.summarize(out, refB, fl)
            assert s["n_bad_flagged"] == 0 and s["n_bad_unflagged"] == 0, s
        w.set_wind(9.0, float(i["wd"][0]))
        yaw = np.repeat(i["yaw"], B, axis=0).astype(np.float32)
        out = _with_flags(w, w.step(yaw))
        if name == "thrust_ramp":
            assert not (out["flags"] & parity.RISK_THRUST_RAMP).any()
        _check(out, _oracle(i["x"], i["y"], 9.0, float(i["wd"][0]), yaw), max_flagged_frac=1.0)
        w.close()


def test_grouped_launch_follows_the_padded_farm_count(layouts):
    """HornsRev1 x 65536: the plain batch is exactly one full round of the G = 2 x 2 kernel (128 farms per block, two
    blocks per CU); grouped by 64 series rows the launch is padded per group, which that kernel's blocks would double:
    grouped launches use G = 4, with 1057 blocks of farm slots the one-slot kernel (three blocks per CU) rather than the
    two-slot one (a third round); back on a shared wind the first choice returns.  Parity on a sample of farms in each
    state."""
    import torch

    from wfcrl_env_amd.backend import WfStep

    l = layouts["HornsRev1_"]
    B, N = 65536, 80
    rng = np.random.default_rng(77)
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    yaw_t = (torch.rand((B, N), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)) * 60 - 30).float()
    yaw = yaw_t.cpu().numpy()
    idx = rng.choice(B, 192, replace=False)

    def check_sample():
        out = {k: v.cpu().numpy()[idx] for k, v in w.step(yaw_t).items()}
        ws, wd = w.get_wind()
        out["flags"] = w.risk_flags()[idx]
        _check(out, _oracle(l["xcoords"], l["ycoords"], ws[idx], wd[idx], yaw[idx]), max_flagged_frac=1.0)

    w.set_wind(8.0, 270.0)
    k = w.kernel_info()
    assert (k["one_block_kernel"], k["lanes_per_env"], k["slots_per_lane"]) == (1, 2, 2)
    check_sample()
    series = np.stack([rng.uniform(6, 12, 64), rng.uniform(0, 360, 64)], axis=1)
    w.set_wind_series(series, seed=3)
    w.wind_series_step()
    k = w.kernel_info()
    assert k["direction_groups"] == 64 and (k["one_block_kernel"], k["lanes_per_env"], k["slots_per_lane"]) == (1, 4, 1), k
    check_sample()
    w.wind_series_step()
    check_sample()
    w.set_wind(9.0, 255.0)
    k = w.kernel_info()
    assert (k["one_block_kernel"], k["lanes_per_env"], k["slots_per_lane"]) == (1, 2, 2)
    check_sample()
    w.close()


def test_two_threads_force_different_kernel_families_on_two_handles(layouts):
    """SURVEY §8(b) "Threading": the kernel choice is per handle (wf_set_kernel_choice), not process-global — two threads
    drive two handles concurrently, one forced onto the register-slot kernel and one onto wf_step_ll_kernel<4,2>, for
    many steps; each keeps its own kernel and both match the oracle (and each other)."""
    import threading

    from wfcrl_env_amd.backend import WfStep

    l = layouts["HornsRev1_"]
    x, y, N, B = l["xcoords"], l["ycoords"], 80, 192
    rng = np.random.default_rng(31)
    yaw = rng.uniform(-30, 30, (B, N)).astype(np.float32)
    ref = _oracle(x, y, 8.0, 263.0, yaw)
    choices = [dict(one_block=False, slot="16x5"), dict(one_block="4x2"), dict(one_block="8")]
    want = [(0, 16, 5), (1, 4, 2), (1, 8, 1)]
    results, errors = [None] * len(choices), []

    def work(k):
        try:
            w = WfStep(x, y, env_batch=B, kernel_choice=choices[k])
            w.set_wind(8.0, 263.0)
            info = w.kernel_info()
            assert (info["one_block_kernel"], info["lanes_per_env"], info["slots_per_lane"]) == want[k], info
            out = None
            for _ in range(40):
                out = w.step(yaw)
            assert w.kernel_info()["one_block_kernel"] == want[k][0]
            results[k] = _with_flags(w, out)
            w.close()
        except Exception as e:  # pragma: no cover
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(choices))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for r in results:
        _check(r, ref)
    assert np.abs(results[0]["power"] / results[1]["power"] - 1).max() < 2e-5


@pytest.mark.parametrize("hh,shear,veer", [(0.9, 0.0, 0.0), (0.56, 0.0, -6.0), (0.714, 0.12, 0.0)])
def test_on_the_fly_kernels_on_an_aligned_grid(hh, shear, veer):
    """An 8 x 14 grid at exactly 360 deg: every turbine has up to thirteen sources exactly upstream (lateral offset 1e-14
    m), where the rotation vortex' core factor 1 - exp(-r^2 / eps^2) has r^2 = 2e-6 m^2 and cancels in float32.  The
    on-the-fly kernels returned 0 for it and stood at wd 3e-4 .. 1e-3 deg / power 3e-4 here (fuzz seed 611 case 516, seed
    622 case 506, found with the re-solve on: not a flagged event; tests/tools/wd_error_probe.py, wd_error_small.py); with
    the series of the core factor they are where the table path (float64 coefficients) is.  Lateral offsets are taken
    from the float64 coordinates, so the layout may as well sit in UTM-like coordinates."""
    import torch

    import parity
    from oracle import c_oracle
    from oracle.floris_gch_numpy import ModelParams
    from wfcrl_env_amd.backend import WfStep

    D = 100.5
    gx, gy = np.meshgrid(np.arange(8) * 5 * D, np.arange(14) * 4 * D, indexing="ij")
    N, B = gx.size, 48
    rng = np.random.default_rng(7)
    yaw = rng.uniform(-35, 35, (B, N)).astype(np.float32)
    ws, wd = rng.uniform(5, 20, B), np.full(B, 360.0)
    model = dict(rotor_diameter=D, hub_height=hh * D, shear=shear, veer=veer)
    mp = ModelParams(D=D, HH=hh * D, shear=shear, veer=veer)
    for x0, y0 in ((0.0, 0.0), (512000.0, 6175000.0)):
        x, y = gx.ravel() + x0, gy.ravel() + y0
        ref = c_oracle.farm_step_batch(x, y, ws, wd, yaw.astype(np.float64), mp, margin=True)
        for choice, per_farm in ((dict(one_block=False, pair_table=False), False), (dict(one_block="4x2"), True), (dict(one_block=False), False)):
            w = WfStep(x, y, env_batch=B, model=model, kernel_choice=choice)
            if per_farm:  # device arrays: a direction per farm, the one-block kernel on the fly
                w.set_wind(torch.from_numpy(ws).cuda(), torch.from_numpy(wd).cuda())
            else:
                w.set_wind(ws, wd)
            assert w.kernel_info()["pair_table"] == (0 if (per_farm or "pair_table" in choice) else 1)
            got = w.step(yaw)
            got = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in got.items()}
            fl = w.risk_flags()
            parity.check({k: v.copy() for k, v in got.items()}, ref, fl, max_flagged_frac=0.2)
            e = parity.errors(got, ref)
            assert e["wd"][fl == 0].max() < 1e-4 and e["power"][fl == 0].max() < 3e-5, {k: float(v[fl == 0].max()) for k, v in e.items()}
            w.close()


def test_rounds_model_picks_the_16x1_family_at_three_blocks_per_cu(layouts):
    """12 288 farms of 80 turbines are one round of the G = 16 one-block kernel (three blocks per CU) and 1.5 of the
    register-slot kernel: the rounds model picks it (wf_dispatch.hip); a wind per farm on that handle runs the G = 8
    on-the-fly kernel.  Sampled farms against the oracle on both."""
    import parity
    from wfcrl_env_amd.backend import WfStep

    l = layouts["HornsRev1_"]
    x, y, N, B = l["xcoords"], l["ycoords"], 80, 12288
    rng = np.random.default_rng(3)
    yaw = rng.uniform(-30, 30, (B, N)).astype(np.float32)
    w = WfStep(x, y, env_batch=B)
    idx = np.arange(0, B, 97)
    w.set_wind(8.0, 263.0)
    info = w.kernel_info()
    assert (info["one_block_kernel"], info["lanes_per_env"], info["slots_per_lane"]) == (1, 16, 1)
    out = w.step(yaw)
    parity.check({k: v[idx] for k, v in out.items()}, _oracle(x, y, 8.0, 263.0, yaw[idx]), w.risk_flags()[idx], max_flagged_frac=0.1)
    ws, wd = np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360
    w.set_wind(ws, wd)
    info = w.kernel_info()
    assert (info["one_block_kernel"], info["lanes_per_env"], info["slots_per_lane"], info["pair_table"]) == (1, 8, 1, 0)
    out = w.step(yaw)
    parity.check({k: v[idx] for k, v in out.items()}, _oracle(x, y, ws[idx], wd[idx], yaw[idx]), w.risk_flags()[idx], max_flagged_frac=0.1)
    w.close()


def test_mixed_launch_for_a_batch_just_beyond_whole_rounds(layouts):
    """wf_kernel_choice::mixed (the envelope holes): 69 632 HornsRev1 farms are one round of the 2x2 kernel
    (65 536) plus 4 096 farms that would cost a second one.  The handle serves the whole rounds with the family and the
    remainder with wf_step_kernel behind it (WfGroupArgs::env_base / env_end): parity on farms of BOTH ranges and across the
    seam, flags of every farm written, the fused env step too; mixed=False is one launch; results of the two modes agree within
    the parity tolerances (two kernel families)."""
    import torch

    from wfcrl_env_amd.backend import WfStep

    l = layouts["HornsRev1_"]
    N, B = 80, 69632
    rng = np.random.default_rng(696)
    yaw = torch.from_numpy(rng.uniform(-30, 30, (B, N)).astype(np.float32)).cuda()
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B, kernel_choice=dict(one_block="2x2"))
    w.set_wind(8.0, 270.0)
    info = w.kernel_info()
    M = info["mixed_main_farms"]
    assert info["one_block_kernel"] == 1 and 0 < M < B and M % 65536 == 0, info
    out = {k: v.cpu().numpy() for k, v in w.step(yaw).items()}
    flags = w.risk_flags()
    idx = np.concatenate([rng.choice(M, 24, replace=False), np.arange(M - 4, M + 4), M + rng.choice(B - M, 24, replace=False), [B - 1]])
    ref = _oracle(l["xcoords"], l["ycoords"], 8.0, 270.0, yaw.cpu().numpy()[idx])
    _check(dict({k: v[idx] for k, v in out.items()}, flags=flags[idx]), ref)
    # fused env step through the same two launches: the transition of every farm, the reward of both ranges
    w.env_config(load_coef=0.1)
    w.env_reset()
    act = torch.from_numpy(rng.uniform(-5, 5, (B, N)).astype(np.float32)).cuda()
    e = w.env_step(act, want=("reward", "yaw", "power", "load"))
    assert torch.equal(e["yaw"], act.clamp(-5, 5))
    ref2 = _oracle(l["xcoords"], l["ycoords"], 8.0, 270.0, e["yaw"].cpu().numpy()[idx])
    r_ref = (ref2["power"] / 1e6 * 1e3 / 8.0 ** 3).mean(axis=1) - 0.1 * np.abs(ref2["load"]).mean(axis=(1, 2))
    ok = w.risk_flags()[idx] == 0
    assert np.abs(e["reward"].cpu().numpy()[idx] - r_ref)[ok].max() < 5e-5 * np.abs(r_ref).max()
    w.close()
    w1 = WfStep(l["xcoords"], l["ycoords"], env_batch=B, kernel_choice=dict(one_block="2x2", mixed=False))
    w1.set_wind(8.0, 270.0)
    assert w1.kernel_info()["mixed_main_farms"] == 0
    o1 = {k: v.cpu().numpy() for k, v in w1.step(yaw).items()}
    assert np.array_equal(o1["power"][:M], out["power"][:M])  # the family's share: the same kernel, the same bits
    assert np.abs(o1["power"][M:] / np.maximum(out["power"][M:], 1e3) - out["power"][M:] / np.maximum(out["power"][M:], 1e3)).max() < 2e-4
    w1.close()


def test_mixed_off_after_a_calibrated_mixed_handle_is_one_launch(layouts):
    """the process-wide calibration cache was keyed without `mixed`: a handle created with mixed=False AFTER a
    default handle of the same configuration had calibrated a mixed launch inherited the split (two launches, against the
    header's "0: always one launch").  Neither handle forces `one_block`, so both go through the calibration and its cache."""
    import torch

    from wfcrl_env_amd.backend import WfStep

    l = layouts["HornsRev1_"]
    N, B = 80, 69632
    rng = np.random.default_rng(697)
    yaw = torch.from_numpy(rng.uniform(-30, 30, (B, N)).astype(np.float32)).cuda()
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    w.set_wind(8.0, 270.0)
    w.step(yaw); w.sync()  # calibrates (and caches) this configuration
    first = w.calibration()
    info = w.kernel_info()
    w.close()
    w1 = WfStep(l["xcoords"], l["ycoords"], env_batch=B, kernel_choice=dict(mixed=False))
    w1.set_wind(8.0, 270.0)
    o1 = w1.step(yaw); w1.sync()
    assert w1.kernel_info()["mixed_main_farms"] == 0, (info, first, w1.kernel_info())
    assert w1.calibration()["mixed_main_farms"] == 0
    idx = np.concatenate([rng.choice(B, 32, replace=False), [65535, 65536, B - 1]])
    ref = _oracle(l["xcoords"], l["ycoords"], 8.0, 270.0, yaw.cpu().numpy()[idx])
    _check(dict({k: v.cpu().numpy()[idx] for k, v in o1.items()}, flags=w1.risk_flags()[idx]), ref)
    w1.close()


def test_one_slot_families_run_their_two_wave_build_when_no_third_block_per_cu(layouts):
    """Round 5: the one-slot one-block kernels exist in two builds — three waves per SIMD (168 registers: they spill since the
    hot records go through LDS) for launches that reach a third block per CU, two waves per SIMD (no spill, no private segment)
    for all others (wf_kernels_ll.hip: OCC2; launch_ll picks by the launch's blocks per CU).  kernel_info reports the build that
    runs; both are held to the oracle."""
    import parity
    from wfcrl_env_amd.backend import WfStep

    l = layouts["HornsRev1_"]
    x, y, N = l["xcoords"], l["ycoords"], 80
    for B, per_farm, want_scratch in ((8192, False, False), (12288, False, True), (8192, True, False)):
        rng = np.random.default_rng(B + per_farm)
        yaw = rng.uniform(-30, 30, (B, N)).astype(np.float32)
        w = WfStep(x, y, env_batch=B, kernel_choice=dict(one_block="16" if not per_farm else "8", calibrate=False))
        if per_farm:
            ws, wd = np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360
        else:
            ws, wd = 8.0, 263.0
        w.set_wind(ws, wd)
        info = w.kernel_info()
        assert info["one_block_kernel"] == 1 and info["slots_per_lane"] == 1 and info["pair_table"] == (0 if per_farm else 1)
        assert (info["scratch_bytes"] > 0) == want_scratch, info
        assert info["vgprs"] == 168 if want_scratch else 168 < info["vgprs"] <= 256, info
        out = w.step(yaw)
        idx = np.arange(0, B, 61)
        ref = _oracle(x, y, ws if not per_farm else ws[idx], wd if not per_farm else wd[idx], yaw[idx])
        parity.check({k: v[idx] for k, v in out.items()}, ref, w.risk_flags()[idx], max_flagged_frac=0.1)
        w.close()


def test_negative_rotor_speed_keeps_the_reference_turbulence_floor():
    """Round-5 fuzz, case 5041 / 1720 (tests/golden/negative_rotor_speed_case.npz: the case's inputs): a 7 x 3 farm at 1.6 D
    behind a thrust table clipped at 0.9999 drives one rotor's mean speed NEGATIVE; FLORIS keeps computing — the yaw-added
    mixing term I_tot / ubar turns negative there and takes the turbine's TI to -0.56, which the solver's
    maximum(sqrt(ti_added^2 + ambient^2), TI) over ALL turbines at the end of the same source step lifts back to ambient
    (oracle: 0.04).  Every kernel family stored TI + dTI unlifted.  Float32 kernels (both families), the flagged-farm re-solve
    and the float64 kernels on every farm against the oracle."""
    import parity
    from oracle.floris_gch_numpy import ModelParams
    from wfcrl_env_amd.backend import WfStep

    d = np.load(os.path.join(ROOT, "tests", "golden", "negative_rotor_speed_case.npz"))
    model = eval(str(d["model"]))
    ren = {"rotor_diameter": "D", "hub_height": "HH"}
    mp = ModelParams(**{ren.get(k, k): v for k, v in model.items()})
    x, y, yaw = d["x"], d["y"], d["yaw"]
    ws, wd = float(d["ws"][0]), float(d["wd"][0])
    ref = _oracle(x, y, ws, wd, yaw, mp)
    assert ref["wind_speed"].min() < -1.0 and np.isclose(ref["load"][..., 0].min(), 0.04)  # the regime, and FLORIS' floor
    for choice, mode in ((dict(slot=(32, 4), one_block="4"), 0), (dict(slot=(32, 4), one_block=False), 0), (dict(one_block="2x2"), 1), (None, 2)):
        w = WfStep(x, y, env_batch=yaw.shape[0], model=dict(model), kernel_choice=choice)
        w.set_risk_resolve(mode)
        w.set_wind(ws, wd)
        out = w.step(yaw)
        if mode == 0:
            parity.check(out, ref, w.risk_flags(), max_flagged_frac=1.0)
        else:
            # (TOL also for the float64 kernels: the rotor with the negative mean is the difference of two free-stream-sized
            # numbers, and the kernels' combined deficit sits 6e-7 from the oracle's — parity._ws_scale)
            parity.check_strict(out, ref, parity.TOL)
        assert np.abs(np.asarray(out["load"])[..., 0] - ref["load"][..., 0]).max() < 1e-5
        w.close()
