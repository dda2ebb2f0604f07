"""GPU: several turbine definitions per farm (include/wfstep.h: wf_set_turbine_types; csrc/wf_resolve_mt.hip).

farm.turbine_type of a FLORIS case is a list (reference wfcrl/simulators/floris/inputs/template/case.yaml:27-28; the
template writes one entry, FLORIS 3.5 takes one per turbine).  The product serves definitions that share the rotor by
solving every farm with the float64 kernels; the checker is the CPU oracle with the same definitions
(oracle/floris_gch_numpy.py: ModelParams.turbine_defs) — the device float64 kernel is never compared with itself.
"""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _definitions():
    """Three definitions on the nrel_5MW rotor: the library turbine, a derated one (own tables, TSR, pP, efficiency,
    reference density), a coarse five-point table."""
    from oracle.floris_gch_numpy import ModelParams

    base = ModelParams()
    derated = dict(table_ws=list(base.table_ws), table_ct=[0.9 * c for c in base.table_ct],
                   table_cp=[0.8 * c for c in base.table_cp], tsr=7.0, pP=2.0, gen_eff=0.95, ref_density=1.2)
    coarse = dict(table_ws=[0.0, 3.0, 9.0, 12.0, 25.0, 25.5], table_ct=[0.0, 0.85, 0.8, 0.45, 0.1, 0.0],
                  table_cp=[0.0, 0.25, 0.46, 0.4, 0.05, 0.0], tsr=8.5, pP=1.7)
    return [{}, derated, coarse]


def _oracle_params(defs, type_of):
    from oracle.floris_gch_numpy import ModelParams

    ren = {"tsr": "TSR"}
    return ModelParams(turbine_defs=[{ren.get(k, k): v for k, v in d.items()} for d in defs], turbine_type_of=list(type_of))


def _oracle(l, ws, wd, yaw, mp):
    from oracle import c_oracle

    return c_oracle.farm_step_batch(l["xcoords"], l["ycoords"], ws, wd, np.asarray(yaw, dtype=np.float64), mp, margin=True)


# (B <= 512 farms: the four-wave kernel; beyond: the one-wave kernel — csrc/wf_resolve.hip: wfk_launch_resolve)
CASES = [("Turb3_Row1_", 64), ("Ablaincourt_", 300), ("Ablaincourt_", 700), ("Turb_TCRWP_", 200), ("Turb16_Row5_", 640),
         ("Ormonde_", 96), ("HornsRev1_", 80), ("HornsRev1_", 560)]


@pytest.mark.parametrize("name,B", CASES)
@pytest.mark.parametrize("mode", ["shared", "per_env"])
def test_mixed_farm_is_the_oracle(layouts, name, B, mode):
    import parity
    from wfcrl_env_amd.backend import WfStep

    l = layouts[name]
    N = l["num_turbines"]
    rng = np.random.default_rng(zlib.crc32(f"types/{name}/{B}/{mode}".encode()))
    defs = _definitions()
    type_of = rng.integers(0, len(defs), N)
    yaw = rng.uniform(-35, 35, (B, N)).astype(np.float32)
    if mode == "shared":
        ws, wd = np.array([8.0]), np.array([270.0])
    else:
        ws, wd = np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B, model=dict(turbine_defs=defs, turbine_type_of=type_of))
    assert w.turbine_types() == 3 and w.risk_resolve() == 2
    w.set_wind(ws, wd)
    out = w.step(yaw)
    assert w.resolve_stats()["n_resolved"] == B and not w.risk_flags().any()
    ref = _oracle(l, ws, wd, yaw, _oracle_params(defs, type_of))
    parity.check_strict(out, ref, parity.TOL_F64)
    # ... and the definitions matter: the plain farm's powers are somewhere else
    plain = _oracle(l, ws, wd, yaw, None)
    assert np.abs(plain["power"] - ref["power"]).max() > 1e4
    w.close()


def test_one_definition_equal_to_the_model_is_the_plain_float64_solve(layouts):
    """The kernels of wf_resolve_mt.hip are wf_resolve.hip's with a table lookup per turbine: with the model's own table as
    the only definition they return the bits of mode 2 on a plain handle (both kernels: 200 and 700 farms)."""
    from wfcrl_env_amd.backend import WfStep

    l = layouts["Turb16_Row5_"]
    N = l["num_turbines"]
    for B in (200, 700):
        rng = np.random.default_rng(B)
        yaw = rng.uniform(-35, 35, (B, N)).astype(np.float32)
        ws, wd = np.clip(8 * rng.weibull(8, B), 3, 28), rng.normal(270, 20, B) % 360
        a = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
        a.set_risk_resolve(2)
        a.set_wind(ws, wd)
        oa = {k: np.array(v) for k, v in a.step(yaw).items()}
        a.close()
        b = WfStep(l["xcoords"], l["ycoords"], env_batch=B, model=dict(turbine_defs=[{}], turbine_type_of=[0] * N))
        b.set_wind(ws, wd)
        ob = b.step(yaw)
        for k in oa:
            assert np.array_equal(oa[k], np.asarray(ob[k])), k
        b.close()


def test_definitions_can_be_cleared_and_mode_zero_is_refused(layouts):
    import parity
    from wfcrl_env_amd.backend import WfStep

    l = layouts["Ablaincourt_"]
    N = l["num_turbines"]
    B = 128
    rng = np.random.default_rng(3)
    yaw = rng.uniform(-30, 30, (B, N)).astype(np.float32)
    defs = _definitions()[:2]
    type_of = [0, 1, 1, 0, 1, 0, 1]
    w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
    w.set_risk_resolve(1)
    w.set_wind(8.0, 270.0)
    plain = {k: np.array(v) for k, v in w.step(yaw).items()}
    w.set_turbine_types(defs, type_of)
    assert w.turbine_types() == 2 and w.risk_resolve() == 2
    with pytest.raises(ValueError, match="mode 0 cannot be served"):
        w.set_risk_resolve(0)
    mixed = w.step(yaw)
    parity.check_strict(mixed, _oracle(l, np.array([8.0]), np.array([270.0]), yaw, _oracle_params(defs, type_of)), parity.TOL_F64)
    w.set_turbine_types(None, None)
    assert w.turbine_types() == 0 and w.risk_resolve() == 1
    again = w.step(yaw)
    for k in plain:
        assert np.array_equal(plain[k], np.asarray(again[k])), k
    # a bad index / too many definitions / a definition index per turbine of another layout: loud
    with pytest.raises(ValueError, match="outside 0..n_types-1"):
        w.set_turbine_types(defs, [0, 1, 2, 0, 0, 0, 0])
    with pytest.raises(ValueError, match="WF_MAX_TURBINE_TYPES"):
        w.set_turbine_types([{}] * 5, [0] * N)
    with pytest.raises(ValueError, match="cannot differ"):
        w.set_turbine_types([{}, {"rotor_diameter": 100.0}], [0] * N)
    w.close()


def test_case_yaml_with_two_turbine_types_through_the_interface(tmp_path):
    """farm.turbine_type with one entry per turbine, through `HipFlorisInterface.from_yaml` (the reference's constructor path,
    reference wfcrl/interface.py:462-479 reads the case) against the oracle with the same definitions."""
    import yaml

    from oracle.floris_gch_numpy import ModelParams, farm_step
    from wfcrl_env_amd.environments.data_cases import named_cases_dictionary
    from wfcrl_env_amd.interface import HipFlorisInterface
    from wfcrl_env_amd.simul_utils import case_config

    case = named_cases_dictionary["Ablaincourt_"][1]
    cfg = case_config(case.dict())
    derated = {"turbine_type": "derated", "rotor_diameter": 126.0, "hub_height": 90.0, "TSR": 7.0, "pP": 2.0, "pT": 1.88,
               "generator_efficiency": 0.95, "ref_density_cp_ct": 1.225,
               "power_thrust_table": {"wind_speed": [0.0, 3.0, 9.0, 12.0, 25.0, 25.5], "thrust": [0.0, 0.85, 0.8, 0.45, 0.1, 0.0],
                                      "power": [0.0, 0.25, 0.46, 0.4, 0.05, 0.0]}}
    type_of = [0, 1, 0, 1, 1, 0, 1]
    cfg["farm"]["turbine_type"] = ["nrel_5MW" if t == 0 else derated for t in type_of]
    cfg["flow_field"]["wind_speeds"], cfg["flow_field"]["wind_directions"] = [9.0], [265.0]
    path = tmp_path / "case.yaml"
    with open(path, "w") as fp:
        yaml.safe_dump(cfg, fp)
    fi = HipFlorisInterface.from_yaml(str(path))
    assert fi.fi.turbine_types() == 2
    yaw = np.array([12.0, -8.0, 0.0, 20.0, -15.0, 5.0, -25.0])
    fi.update_command(yaw=yaw)
    d = dict(table_ws=derated["power_thrust_table"]["wind_speed"], table_ct=derated["power_thrust_table"]["thrust"],
             table_cp=derated["power_thrust_table"]["power"], TSR=7.0, pP=2.0, gen_eff=0.95, ref_density=1.225)
    mp = ModelParams(turbine_defs=[{}, d], turbine_type_of=type_of)
    ref = farm_step(np.array(case.xcoords), np.array(case.ycoords), 9.0, 265.0, yaw, mp)
    got = np.asarray(fi.avg_powers(), dtype=np.float64)
    assert np.abs(got - ref["power"]).max() <= 2e-6 * np.abs(ref["power"]).max()
    # definitions that do not share the rotor are refused where the case is read
    from wfcrl_env_amd.simul_utils import UnsupportedCaseError, load_case_yaml

    cfg["farm"]["turbine_type"][1] = dict(derated, hub_height=100.0)
    with pytest.raises(UnsupportedCaseError, match="share hub_height"):
        load_case_yaml(cfg)


def test_definitions_with_layouts_of_different_turbine_counts():
    """The definition index is per turbine SLOT, shared by the layouts of a batch (include/wfstep.h); the placeholders of a
    shorter layout (wf_set_layouts_counts) receive wakes and give none whatever their slot's definition.  Checker: the oracle
    on the UNPADDED layouts with the definitions of their slots; fused env step: the reward over the real turbines."""
    import parity
    from oracle import c_oracle
    from wfcrl_env_amd.backend import WfStep

    rng = np.random.default_rng(77)
    counts, B = (12, 9, 7), 96
    N, K = max(counts), len(counts)
    X = [list(rng.uniform(0, 2500, n)) for n in counts]
    Y = [list(rng.uniform(0, 1500, n)) for n in counts]
    layout_of = rng.integers(0, K, B).astype(np.int32)
    defs = _definitions()
    type_of = rng.integers(0, len(defs), N)
    yaw = rng.uniform(-30, 30, (B, N)).astype(np.float32)
    ws, wd = np.clip(8 * rng.weibull(8, B), 4, 20), rng.uniform(0, 360, B)
    w = WfStep(X, Y, env_batch=B, layout_of=layout_of, model=dict(turbine_defs=defs, turbine_type_of=type_of))
    assert w.turbine_types() == 3 and list(w.turbine_counts) == list(counts)
    w.set_wind(ws, wd)
    out = {k: np.asarray(v) for k, v in w.step(yaw).items()}
    for l in range(K):
        idx = np.flatnonzero(layout_of == l)
        n = counts[l]
        ref = c_oracle.farm_step_batch(np.array(X[l]), np.array(Y[l]), ws[idx], wd[idx], yaw[idx, :n].astype(np.float64),
                                       _oracle_params(defs, type_of[:n]), margin=True)
        got = {k: v[idx] for k, v in out.items()}
        for k, v in got.items():
            assert not v[:, n:].any(), (l, k)
        parity.check_strict({k: v[:, :n] for k, v in got.items()}, ref, parity.TOL_F64)
    w.close()
