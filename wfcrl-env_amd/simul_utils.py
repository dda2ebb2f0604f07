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
