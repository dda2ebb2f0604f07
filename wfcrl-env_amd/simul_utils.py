"""Case-file helper.  The reference writes a FLORIS case.yaml on every make() (simul_utils.py:34-48,
interface.py:533-538); the HIP backend needs no file, so this is an optional dump for inspection
(SURVEY Appendix C10)."""
from __future__ import annotations

from pathlib import Path

from .backend import default_model


def case_config(case: dict) -> dict:
    """The content the reference's template (case.yaml:1-89) would have for this case."""
    m = default_model()
    cfg = {
        "name": "GCH", "floris_version": "v3.0.0",
        "solver": {"type": "turbine_grid", "turbine_grid_points": 3},
        "farm": {"layout_x": list(case["xcoords"]), "layout_y": list(case["ycoords"]), "turbine_type": ["nrel_5MW"]},
        "flow_field": {"air_density": m["air_density"], "reference_wind_height": -1,
                       "turbulence_intensity": m["ambient_ti"], "wind_directions": [270.0],
                       "wind_shear": m["shear"], "wind_speeds": [8.0], "wind_veer": m["veer"]},
        "wake": {
            "model_strings": {"combination_model": "sosfs", "deflection_model": "gauss",
                              "turbulence_model": "crespo_hernandez", "velocity_model": "gauss"},
            "enable_secondary_steering": True, "enable_yaw_added_recovery": True,
            "enable_transverse_velocities": True,
            "wake_deflection_parameters": {"gauss": {k: m[k] for k in ("ad", "alpha", "bd", "beta", "dm", "ka", "kb")}},
            "wake_velocity_parameters": {"gauss": {k: m[k] for k in ("alpha", "beta", "ka", "kb")}},
            "wake_turbulence_parameters": {"crespo_hernandez": {"initial": m["ch_initial"], "constant": m["ch_constant"],
                                                                 "ai": m["ch_ai"], "downstream": m["ch_downstream"]}},
        },
    }
    if case.get("direction") is not None:
        cfg["flow_field"]["wind_directions"] = [case["direction"]]
    if case.get("speed") is not None:
        cfg["flow_field"]["wind_speeds"] = [case["speed"]]
    return cfg


def dump_case_yaml(case: dict, output_dir) -> str:
    import yaml

    out = Path(output_dir)
    out.mkdir(parents=True, exist_ok=True)
    path = out / "case.yaml"
    with open(path, "w") as fp:
        yaml.safe_dump(case_config(case), fp)
    return str(path)


def create_floris_case(case: dict, output_dir=None) -> str:
    """Name and signature of reference wfcrl/simul_utils.py:34-48: writes `<output_dir>/case.yaml`."""
    return dump_case_yaml(case, "__simul__/floris/case" if output_dir is None else output_dir)


# ---------------------------------------------------------------------------------------------------
# FLORIS-YAML ingestion: read a (possibly user-modified) case.yaml into what the C ABI takes.
# ---------------------------------------------------------------------------------------------------
_SUPPORTED_MODELS = {"combination_model": "sosfs", "deflection_model": "gauss", "turbulence_model": "crespo_hernandez",
                     "velocity_model": "gauss"}


class UnsupportedCaseError(ValueError):
    """The case selects a FLORIS option the HIP kernels do not implement."""


def _turbine_fields(t) -> dict:
    if isinstance(t, str):
        if t != "nrel_5MW":
            raise UnsupportedCaseError(f"turbine library entry {t!r}: only 'nrel_5MW' is built in; pass the turbine "
                                       "as an inline dict (FLORIS turbine yaml content) instead")
        return {}
    out = {}
    ren = {"rotor_diameter": "rotor_diameter", "hub_height": "hub_height", "TSR": "tsr", "pP": "pP", "pT": "pT",
           "generator_efficiency": "gen_eff", "ref_density_cp_ct": "ref_density"}
    for k, v in ren.items():
        if k in t:
            out[v] = float(t[k])
    if "ref_tilt_cp_ct" in t and "tilt_angle" in t and float(t["ref_tilt_cp_ct"]) != float(t.get("tilt_angle", t["ref_tilt_cp_ct"])):
        raise UnsupportedCaseError("tilt_angle != ref_tilt_cp_ct (tilt corrections) is not implemented")
    tab = t.get("power_thrust_table")
    if tab is not None:
        out["table_ws"] = [float(v) for v in tab["wind_speed"]]
        out["table_ct"] = [float(v) for v in tab["thrust"]]
        out["table_cp"] = [float(v) for v in tab["power"]]
    return out


def load_case_yaml(path_or_dict) -> dict:
    """Parse a FLORIS v3 input (file path or already-loaded dict) into
    {"xcoords", "ycoords", "speed", "direction", "model": {...wf_model_params fields...}}.
    Raises UnsupportedCaseError for anything outside the reference template's model family
    (reference wfcrl/simulators/floris/inputs/template/case.yaml:14-89)."""
    if isinstance(path_or_dict, dict):
        cfg = path_or_dict
    else:
        import yaml

        with open(path_or_dict) as fp:
            cfg = yaml.safe_load(fp)
    solver = cfg.get("solver", {})
    if solver.get("type", "turbine_grid") != "turbine_grid" or int(solver.get("turbine_grid_points", 3)) != 3:
        raise UnsupportedCaseError("only solver.type = turbine_grid with turbine_grid_points = 3 is implemented")
    farm, flow, wake = cfg["farm"], cfg["flow_field"], cfg["wake"]
    for key, want in _SUPPORTED_MODELS.items():
        got = wake["model_strings"].get(key)
        if got != want:
            raise UnsupportedCaseError(f"wake.model_strings.{key} = {got!r} is not implemented (only {want!r})")
    if len(flow.get("wind_speeds", [0])) != 1 or len(flow.get("wind_directions", [0])) != 1:
        raise UnsupportedCaseError("exactly one wind speed and one wind direction per case (as the reference uses)")
    if flow.get("heterogenous_inflow_config") or flow.get("heterogeneous_inflow_config"):
        raise UnsupportedCaseError("heterogeneous inflow is not implemented")
    # farm.turbine_type is a list (case.yaml:27-28): one entry for every turbine, or one entry per turbine.  Distinct entries
    # become the backend's turbine definitions (include/wfstep.h: wf_set_turbine_types) as long as they share the rotor.
    ttypes = farm.get("turbine_type", ["nrel_5MW"])
    n_turb = len(farm["layout_x"])
    if len(ttypes) not in (1, n_turb):
        raise UnsupportedCaseError("farm.turbine_type needs one entry, or one per turbine")
    distinct = []
    type_of = []
    for t in ttypes:
        if t not in distinct:
            distinct.append(t)
        type_of.append(distinct.index(t))
    fields = [_turbine_fields(t) for t in distinct]
    model = dict(fields[0])
    if len(distinct) > 1:
        per_def = ("table_ws", "table_ct", "table_cp", "tsr", "pP", "gen_eff", "ref_density")
        nrel = {"rotor_diameter": 126.0, "hub_height": 90.0, "pT": 1.88}  # what a library name stands for in the shared fields
        for k, f in enumerate(fields[1:], 1):
            for name in ("rotor_diameter", "hub_height"):
                if f.get(name, nrel[name]) != fields[0].get(name, nrel[name]):
                    raise UnsupportedCaseError(f"turbine definitions of one farm must share {name} (definition {k} differs): "
                                               "the rotor grid and the vortex geometry are per farm")
        if len(distinct) > 4:
            raise UnsupportedCaseError("at most 4 distinct turbine definitions per farm")
        # a definition that leaves a per-definition field out means FLORIS' nrel_5MW value there, not definition 0's (plain
        # Python data: parsing a case file needs no native library)
        from ._nrel5mw import NREL_5MW_DEFINITION as dflt

        model["turbine_defs"] = [{k: f.get(k, dflt[k]) for k in per_def} for f in fields]
        model["turbine_type_of"] = type_of if len(ttypes) > 1 else [0] * n_turb
    ref_h = float(flow.get("reference_wind_height", -1))
    hub = model.get("hub_height", 90.0)
    if ref_h != -1 and ref_h != hub:
        raise UnsupportedCaseError("reference_wind_height must be -1 (hub height) or equal to the hub height")
    model.update(air_density=float(flow["air_density"]), ambient_ti=float(flow["turbulence_intensity"]),
                 shear=float(flow["wind_shear"]), veer=float(flow.get("wind_veer", 0.0)))
    gd = wake["wake_deflection_parameters"]["gauss"]
    gv = wake["wake_velocity_parameters"]["gauss"]
    for k in ("alpha", "beta", "ka", "kb"):  # each model has its own set (case.yaml:55-59 and 76-80)
        model[k] = float(gv[k])
        model["defl_" + k] = float(gd[k])
    model.update(ad=float(gd.get("ad", 0.0)), bd=float(gd.get("bd", 0.0)), dm=float(gd.get("dm", 1.0)))
    # solver switches (case.yaml:46-50); FLORIS defaults a missing key to false
    for flag in ("enable_secondary_steering", "enable_yaw_added_recovery", "enable_transverse_velocities"):
        model[flag] = bool(wake.get(flag, False))
    ch = wake["wake_turbulence_parameters"]["crespo_hernandez"]
    model.update(ch_initial=float(ch["initial"]), ch_constant=float(ch["constant"]), ch_ai=float(ch["ai"]),
                 ch_downstream=float(ch["downstream"]))
    return {"xcoords": [float(v) for v in farm["layout_x"]], "ycoords": [float(v) for v in farm["layout_y"]],
            "speed": float(flow["wind_speeds"][0]), "direction": float(flow["wind_directions"][0]), "model": model}
