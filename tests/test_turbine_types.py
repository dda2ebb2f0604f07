"""CPU: several turbine definitions per farm — the two oracles against each other and the case.yaml ingestion.

FLORIS 3.5 evaluates the thrust / power tables, TSR, pP and ref_density_cp_ct per turbine through turbine_type_map
(farm.turbine_type of reference wfcrl/simulators/floris/inputs/template/case.yaml:27-28 is a list).  PARITY UNPINNED: the
reference holds no mixed farm; the restatement follows FLORIS' published per-type evaluation (oracle/floris_gch_numpy.py:
ModelParams.turbine_defs).  The HIP side of it: tests/test_turbine_types_gpu.py.
"""
import numpy as np
import pytest


def _defs():
    from oracle.floris_gch_numpy import ModelParams

    base = ModelParams()
    derated = dict(table_ct=[0.9 * c for c in base.table_ct], table_cp=[0.8 * c for c in base.table_cp], TSR=7.0, pP=2.0,
                   gen_eff=0.95, ref_density=1.2)
    coarse = dict(table_ws=[0.0, 3.0, 9.0, 12.0, 25.0, 25.5], table_ct=[0.0, 0.85, 0.8, 0.45, 0.1, 0.0],
                  table_cp=[0.0, 0.25, 0.46, 0.4, 0.05, 0.0], TSR=8.5)
    return [{}, derated, coarse]


def test_numpy_and_c_oracle_agree_on_a_mixed_farm(layouts):
    from oracle import c_oracle
    from oracle import floris_gch_numpy as fn

    for name, B in (("Ablaincourt_", 6), ("Turb16_Row5_", 4)):
        l = layouts[name]
        N = l["num_turbines"]
        rng = np.random.default_rng(N)
        p = fn.ModelParams(turbine_defs=_defs(), turbine_type_of=list(rng.integers(0, 3, N)))
        yaw = rng.uniform(-30, 30, (B, N))
        ws, wd = rng.uniform(5, 14, B), rng.uniform(0, 360, B)
        a = fn.farm_step_batch(l["xcoords"], l["ycoords"], ws, wd, yaw, p)
        b = c_oracle.farm_step_batch(l["xcoords"], l["ycoords"], ws, wd, yaw, p)
        for k in a:
            assert np.abs(a[k] - b[k]).max() <= 1e-11 * np.abs(a[k]).max(), (name, k)
        plain = c_oracle.farm_step_batch(l["xcoords"], l["ycoords"], ws, wd, yaw)
        assert np.abs(plain["power"] - b["power"]).max() > 1e4  # the definitions matter


def test_a_single_definition_equal_to_the_model_changes_nothing(layouts):
    from oracle import c_oracle
    from oracle import floris_gch_numpy as fn

    l = layouts["Turb6_Row2_"]
    rng = np.random.default_rng(1)
    yaw = rng.uniform(-30, 30, (5, 6))
    one = fn.ModelParams(turbine_defs=[{}], turbine_type_of=[0] * 6)
    for mod in (fn, c_oracle):
        a = mod.farm_step_batch(l["xcoords"], l["ycoords"], 8.0, 270.0, yaw)
        b = mod.farm_step_batch(l["xcoords"], l["ycoords"], 8.0, 270.0, yaw, one)
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    with pytest.raises(ValueError, match="cannot differ"):
        fn.ModelParams(turbine_defs=[{"D": 100.0}], turbine_type_of=[0] * 6).definition(0)


def test_case_yaml_with_mixed_turbine_types():
    from wfcrl_env_amd.backend import default_model
    from wfcrl_env_amd.environments.data_cases import named_cases_dictionary
    from wfcrl_env_amd.simul_utils import UnsupportedCaseError, case_config, load_case_yaml

    case = named_cases_dictionary["Ablaincourt_"][1]
    cfg = case_config(case.dict())
    custom = {"turbine_type": "derated", "rotor_diameter": 126.0, "hub_height": 90.0, "TSR": 7.0, "pP": 2.0,
              "generator_efficiency": 0.95,
              "power_thrust_table": {"wind_speed": [0.0, 3.0, 10.0, 25.0, 26.0], "thrust": [0.0, 0.8, 0.7, 0.2, 0.0],
                                     "power": [0.0, 0.3, 0.45, 0.1, 0.0]}}
    cfg["farm"]["turbine_type"] = [custom, "nrel_5MW", custom, custom, "nrel_5MW", "nrel_5MW", custom]
    m = load_case_yaml(cfg)["model"]
    assert m["turbine_type_of"] == [0, 1, 0, 0, 1, 1, 0] and len(m["turbine_defs"]) == 2
    d = default_model()
    # definition 0 is the custom one; the library name stands for FLORIS' nrel_5MW in EVERY per-definition field — not for
    # whatever definition 0 says there
    assert (m["turbine_defs"][0]["tsr"], m["turbine_defs"][0]["gen_eff"], m["turbine_defs"][0]["table_ws"][2]) == (7.0, 0.95, 10.0)
    assert (m["turbine_defs"][1]["tsr"], m["turbine_defs"][1]["pP"], m["turbine_defs"][1]["gen_eff"]) == (d["tsr"], d["pP"], d["gen_eff"])
    assert m["turbine_defs"][1]["table_ct"] == d["table_ct"]
    # one distinct entry repeated per turbine is the plain single-definition case
    cfg["farm"]["turbine_type"] = [custom] * 7
    assert "turbine_defs" not in load_case_yaml(cfg)["model"]
    for bad, msg in ((dict(custom, rotor_diameter=120.0), "share rotor_diameter"), (dict(custom, hub_height=95.0), "share hub_height")):
        cfg["farm"]["turbine_type"] = ["nrel_5MW", bad] + ["nrel_5MW"] * 5
        with pytest.raises(UnsupportedCaseError, match=msg):
            load_case_yaml(cfg)
    cfg["farm"]["turbine_type"] = ["nrel_5MW", custom]
    with pytest.raises(UnsupportedCaseError, match="one per turbine"):
        load_case_yaml(cfg)
