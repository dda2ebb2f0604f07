"""TEST INFRASTRUCTURE — ctypes loader for oracle/libwforacle.so (the C float64 restatement).

Checker / cpu_baseline only; see oracle/floris_gch.c for what it restates and how it is pinned.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from .floris_gch_numpy import ModelParams

_HERE = Path(__file__).resolve().parent
_LIB = None


class _Type(C.Structure):  # wfo_type
    _fields_ = [("TSR", C.c_double), ("pP", C.c_double), ("ref_density", C.c_double), ("n_table", C.c_int),
                ("table_ws", C.POINTER(C.c_double)), ("table_ct", C.POINTER(C.c_double)), ("table_pow", C.POINTER(C.c_double))]


class _Params(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "air_density", "ambient_ti", "shear", "veer",
        "D", "HH", "TSR", "pP", "pT", "gen_eff", "ref_density",
        "alpha", "beta", "ka", "kb", "ad", "bd", "dm",
        "ch_initial", "ch_constant", "ch_ai", "ch_downstream",
        "eps_gain", "num_eps", "kappa", "gch_gain", "overlap_thresh", "near_wake_c",
        "defl_alpha", "defl_beta", "defl_ka", "defl_kb",
    )] + [
        ("n_table", C.c_int),
        ("table_ws", C.POINTER(C.c_double)),
        ("table_ct", C.POINTER(C.c_double)),
        ("table_pow", C.POINTER(C.c_double)),
        ("enable_secondary_steering", C.c_int),
        ("enable_yaw_added_recovery", C.c_int),
        ("enable_transverse_velocities", C.c_int),
        ("n_types", C.c_int),
        ("types", C.POINTER(_Type)),
        ("type_of", C.POINTER(C.c_int)),
    ]


def build(force: bool = False) -> Path:
    so = _HERE / "libwforacle.so"
    src = _HERE / "floris_gch.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B" if force else "-s"], check=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(str(build()))
        _LIB.wfo_step_batch.restype = C.c_int
        _LIB.wfo_step_batch_ex.restype = C.c_int
        _LIB.wfo_max_threads.restype = C.c_int
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class OracleResult(dict):
    """Output arrays of a batch, plus the inputs they belong to as attributes (`yaw` (B, N) float64, `model`)."""

    yaw = None
    model = None


def farm_step_batch(x, y, ws, wd, yaw, p: ModelParams | None = None, nthreads: int = 0, margin: bool = False,
                    tie_reverse: bool = False):
    """Same contract as floris_gch_numpy.farm_step_batch, evaluated by the C restatement.
    margin=True adds out["margin"] (B,): the smallest relative distance of any relevant deficit to the overlap
    threshold (floris_gch.c: farm_step_one).  tie_reverse: exact x' ties in descending instead of ascending original
    index (FLORIS' argsort leaves that order implementation-defined)."""
    p = p or ModelParams()
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    yaw = np.ascontiguousarray(np.atleast_2d(yaw), dtype=np.float64)
    B, N = yaw.shape
    assert x.shape == (N,) and y.shape == (N,)
    ws = np.atleast_1d(np.asarray(ws, dtype=np.float64))
    wd = np.atleast_1d(np.asarray(wd, dtype=np.float64))
    if ws.size == 1 and wd.size == 1:
        stride = 0
    else:
        stride = 1
        ws = np.broadcast_to(ws, (B,))
        wd = np.broadcast_to(wd, (B,))
    ws = np.ascontiguousarray(ws)
    wd = np.ascontiguousarray(wd)
    tws = np.ascontiguousarray(p.table_ws, dtype=np.float64)
    tct = np.ascontiguousarray(p.table_ct, dtype=np.float64)
    tpw = np.ascontiguousarray(p.power_table(), dtype=np.float64)
    cp = _Params()
    for name, _ in _Params._fields_[:32]:
        setattr(cp, name, float(getattr(p, name)))
    for name in ("enable_secondary_steering", "enable_yaw_added_recovery", "enable_transverse_velocities"):
        setattr(cp, name, int(bool(getattr(p, name))))
    cp.n_table = len(tws)
    cp.table_ws, cp.table_ct, cp.table_pow = _dp(tws), _dp(tct), _dp(tpw)
    keep = []  # (the arrays the definitions point into)
    if p.turbine_defs:  # several turbine definitions per farm (ModelParams.turbine_defs)
        types = (_Type * len(p.turbine_defs))()
        for k in range(len(p.turbine_defs)):
            pk = p.definition(k)
            cols = [np.ascontiguousarray(c, dtype=np.float64) for c in (pk.table_ws, pk.table_ct, pk.power_table())]
            keep.append(cols)
            types[k].TSR, types[k].pP, types[k].ref_density, types[k].n_table = pk.TSR, pk.pP, pk.ref_density, len(cols[0])
            types[k].table_ws, types[k].table_ct, types[k].table_pow = (_dp(c) for c in cols)
        type_of = np.ascontiguousarray(p.turbine_type_of, dtype=np.int32)
        assert type_of.shape == (N,) and type_of.min() >= 0 and type_of.max() < len(p.turbine_defs)
        keep.append(type_of)
        cp.n_types, cp.types, cp.type_of = len(p.turbine_defs), types, type_of.ctypes.data_as(C.POINTER(C.c_int))
    out = {
        "power": np.empty((B, N)),
        "wind_speed": np.empty((B, N)),
        "wind_direction": np.empty((B, N)),
        "load": np.empty((B, N, 4)),
    }
    if margin:
        out["margin"] = np.empty(B)
    rc = lib().wfo_step_batch_ex(
        C.byref(cp), C.c_int(N), _dp(x), _dp(y), C.c_int(B), _dp(ws), _dp(wd), C.c_int(stride), _dp(yaw),
        _dp(out["power"]), _dp(out["wind_speed"]), _dp(out["wind_direction"]), _dp(out["load"]), C.c_int(nthreads),
        _dp(out["margin"]) if margin else None, C.c_int(int(bool(tie_reverse))),
    )
    if rc != 0:
        raise RuntimeError(f"wfo_step_batch failed: {rc}")
    if margin:  # what the parity contract needs to judge the kernel's knee / ramp flags (tests/parity.py): attributes,
        out = OracleResult(out)  # not items — the dict holds output arrays only
        out.yaw, out.model = yaw, p
    return out


def max_threads() -> int:
    return int(lib().wfo_max_threads())
