"""ctypes binding of libwfstep.so (C ABI: include/wfstep.h).

This is the binding a maintainer of the reference would add next to wfcrl/interface.py to replace
the `floris.tools.FlorisInterface` calls (INTEGRATION.md shows it in place).  There is no fallback:
a missing or unloadable library raises, and wf_create raises without a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "libwfstep.so"

WF_OK = 0
WF_RISK_OVERLAP = 1     # a deficit within the guard band of the overlap threshold (include/wfstep.h)
WF_RISK_POWER_KNEE = 2  # a turbine on a steep segment (cut-in / cut-out) of the power table
WF_RISK_THRUST_RAMP = 4  # a turbine on the cut-in ramp / cut-out drop of the thrust table (v |dCt/dv| > 5)
WF_RISK_THRUST_UNITY = 8  # a thrust coefficient above 0.995 (user tables): always re-solved in float64, only seen in the raw flags
WF_RISK_NEGATIVE_SPEED = 16  # a rotor-grid speed that is not positive (an unphysically tight farm)
WF_E = {-1: "WF_E_INVALID", -2: "WF_E_UNSUPPORTED", -3: "WF_E_NODEVICE", -4: "WF_E_HIP", -5: "WF_E_NOMEM"}

_MODEL_DOUBLES = (
    "air_density", "ambient_ti", "shear", "veer",
    "rotor_diameter", "hub_height", "tsr", "pP", "pT", "gen_eff", "ref_density",
    "alpha", "beta", "ka", "kb", "ad", "bd", "dm",
    "ch_initial", "ch_constant", "ch_ai", "ch_downstream",
    "eps_gain", "num_eps", "kappa", "gch_gain", "overlap_thresh", "near_wake_c",
    "defl_alpha", "defl_beta", "defl_ka", "defl_kb",
)
_MODEL_SWITCHES = ("enable_secondary_steering", "enable_yaw_added_recovery", "enable_transverse_velocities")


class ModelParams(C.Structure):
    """Mirror of `struct wf_model_params`."""

    _fields_ = [(n, C.c_double) for n in _MODEL_DOUBLES] + [
        ("n_table", C.c_int),
        ("table_ws", C.POINTER(C.c_double)),
        ("table_ct", C.POINTER(C.c_double)),
        ("table_cp", C.POINTER(C.c_double)),
    ] + [(n, C.c_int) for n in _MODEL_SWITCHES]


class TurbineDef(C.Structure):
    """Mirror of `struct wf_turbine_def`."""

    _fields_ = [("n_table", C.c_int), ("table_ws", C.POINTER(C.c_double)), ("table_ct", C.POINTER(C.c_double)),
                ("table_cp", C.POINTER(C.c_double))] + [(n, C.c_double) for n in ("tsr", "pP", "gen_eff", "ref_density")]


class EnvParams(C.Structure):
    """Mirror of `struct wf_env_params`."""

    _fields_ = [(n, C.c_float) for n in ("yaw_lo", "yaw_hi", "yaw_step", "actuator_rate", "dt", "budget", "load_coef")] + [
        ("discrete", C.c_int)]


class WindDist(C.Structure):
    """Mirror of `struct wf_wind_dist`."""

    _fields_ = [(n, C.c_double) for n in ("ws_scale", "ws_shape", "ws_lo", "ws_hi", "wd_mean", "wd_std", "wd_lo", "wd_hi")]


class KernelInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "lanes_per_env", "slots_per_lane", "envs_per_block", "threads_per_block", "grid_blocks",
        "vgprs", "lds_bytes", "scratch_bytes", "pair_table", "direction_groups", "mixed_main_farms", "one_block_kernel")]


class KernelChoice(C.Structure):
    """Mirror of `struct wf_kernel_choice`."""

    _fields_ = [(n, C.c_int) for n in ("slot_G", "slot_S", "one_block", "ll_G", "ll_S", "pair_table", "fly_one_block", "far_skip", "calibrate", "mixed")]


# every symbol include/wfstep.h declares: name -> (restype, argtypes)
_P = C.c_void_p
ABI = {
    "wf_version": (C.c_int, []),
    "wf_default_model": (C.c_int, [C.POINTER(ModelParams)]),
    "wf_turbine_table": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_double)),
                                   C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_double))]),
    "wf_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "wf_destroy": (C.c_int, [_P]),
    "wf_set_stream": (C.c_int, [_P, _P, C.c_int]),
    "wf_get_stream": (_P, [_P]),
    "wf_set_model": (C.c_int, [_P, C.POINTER(ModelParams)]),
    "wf_set_turbine_types": (C.c_int, [_P, C.c_int, C.POINTER(TurbineDef), C.POINTER(C.c_int)]),
    "wf_get_turbine_types": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "wf_set_layout": (C.c_int, [_P, C.c_int, _P, _P]),
    "wf_set_batch": (C.c_int, [_P, C.c_int]),
    "wf_set_layouts": (C.c_int, [_P, C.c_int, _P, _P, _P]),
    "wf_set_layouts_counts": (C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
    "wf_set_wind": (C.c_int, [_P, _P, _P, C.c_int, C.c_int]),
    "wf_set_wind_counts": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_int]),
    "wf_wind_sample_binned": (C.c_int, [_P, C.c_ulonglong, C.POINTER(WindDist), C.c_double]),
    "wf_step": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int]),
    "wf_sync": (C.c_int, [_P]),
    "wf_set_risk_guard": (C.c_int, [_P, C.c_double]),
    "wf_get_risk_flags": (C.c_int, [_P, _P, C.c_int]),
    "wf_set_risk_resolve": (C.c_int, [_P, C.c_int]),
    "wf_get_risk_resolve": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "wf_get_resolve_stats": (C.c_int, [_P, C.POINTER(C.c_int), _P, C.c_int]),
    "wf_wind_sample": (C.c_int, [_P, C.c_ulonglong, C.POINTER(WindDist)]),
    "wf_wind_series": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_ulonglong]),
    "wf_wind_series_step": (C.c_int, [_P]),
    "wf_get_wind": (C.c_int, [_P, _P, _P, C.c_int]),
    "wf_env_config": (C.c_int, [_P, C.POINTER(EnvParams)]),
    "wf_env_set_power_unit": (C.c_int, [_P, C.c_int]),
    "wf_env_reset": (C.c_int, [_P]),
    "wf_env_state": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int]),
    "wf_env_set_prev_wind": (C.c_int, [_P, _P, C.c_int]),
    "wf_env_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int]),
    "wf_timing_begin": (C.c_int, [_P]),
    "wf_timing_end": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "wf_get_kernel_info": (C.c_int, [_P, C.POINTER(KernelInfo)]),
    "wf_set_kernel_choice": (C.c_int, [_P, C.POINTER(KernelChoice)]),
    "wf_get_kernel_choice": (C.c_int, [_P, C.POINTER(KernelChoice)]),
    "wf_get_calibration": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "wf_get_fly_calibration": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "wf_get_mixed_launch": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "wf_calibrate": (C.c_int, [_P]),
    "wf_set_calibration": (C.c_int, [_P, C.c_int, C.c_int]),
    "wf_last_error": (C.c_char_p, [_P]),
}

_lib = None


def build(force: bool = False) -> Path:
    """Compile csrc/ into libwfstep.so with hipcc for gfx950 (cross-compiles without a GPU)."""
    srcs = sorted((PKG_DIR / "csrc").glob("*.hip")) + sorted((PKG_DIR / "csrc").glob("*.h"))
    srcs.append(PKG_DIR.parent / "include" / "wfstep.h")
    stale = (not LIB_PATH.exists()) or any(s.stat().st_mtime > LIB_PATH.stat().st_mtime for s in srcs)
    if force or stale:
        subprocess.run(["make", "-j4", "-C", str(PKG_DIR / "csrc")] + (["-B"] if force else []), check=True)
    return LIB_PATH


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64 (SONAME libamdhip64.so.7).  If libwfstep.so were
    loaded first it would bind /opt/rocm's copy and torch would then load a SECOND runtime (its NEEDED
    entry is the unversioned file name), leaving two HIP runtimes in one process: torch fails to see
    the GPU and torch device pointers are meaningless to our kernels.  Importing torch first makes the
    dynamic loader resolve our `libamdhip64.so.7` to the copy torch already mapped."""
    import importlib.util
    import sys

    if os.environ.get("WFSTEP_NO_TORCH"):
        return
    if "torch" in sys.modules or importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            if os.environ.get("WFSTEP_NO_AUTOBUILD"):
                raise OSError(f"{LIB_PATH} is missing; run __graft_entry__.build() (there is no CPU fallback)")
            build()
        _share_hip_runtime_with_torch()
        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in ABI.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


class WfError(RuntimeError):
    pass


def check(rc: int, handle=None):
    if rc != WF_OK:
        msg = load().wf_last_error(handle)
        text = f"{WF_E.get(rc, rc)}: {msg.decode() if msg else ''}"
        if rc in (-1, -2):
            raise ValueError(text)
        raise WfError(text)
