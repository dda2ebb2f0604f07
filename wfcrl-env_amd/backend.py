"""`WfStep` — thin object wrapper over the C ABI handle (one handle = one device + stream).

Accepts NumPy arrays (host path: staged through pinned buffers inside the library) or torch CUDA
tensors (device path: pointers handed over as-is, call is asynchronous on the handle's stream).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import EnvParams, KernelChoice, KernelInfo, ModelParams, WindDist, check


def _is_torch(a) -> bool:
    return type(a).__module__.startswith("torch")


def default_model() -> dict:
    """Reference defaults (case.yaml + nrel_5MW) as a plain dict incl. the power/thrust table."""
    p = ModelParams()
    check(_lib.load().wf_default_model(C.byref(p)))
    d = {n: getattr(p, n) for n in _lib._MODEL_DOUBLES}
    d.update({n: bool(getattr(p, n)) for n in _lib._MODEL_SWITCHES})
    n = p.n_table
    d["table_ws"] = [p.table_ws[i] for i in range(n)]
    d["table_ct"] = [p.table_ct[i] for i in range(n)]
    d["table_cp"] = [p.table_cp[i] for i in range(n)]
    return d


def turbine_table(name: str = "nrel_5MW_floris3") -> dict:
    """A named power/thrust table shipped with the library (include/wfstep.h: wf_turbine_table) as the three model keys
    `table_ws`, `table_ct`, `table_cp` — e.g. WfStep(..., model=turbine_table("nrel_5MW_survey_a5"))."""
    n = C.c_int(0)
    dp = C.POINTER(C.c_double)
    ws, ct, cp = dp(), dp(), dp()
    check(_lib.load().wf_turbine_table(name.encode(), C.byref(n), C.byref(ws), C.byref(ct), C.byref(cp)))
    return {"table_ws": [ws[i] for i in range(n.value)], "table_ct": [ct[i] for i in range(n.value)],
            "table_cp": [cp[i] for i in range(n.value)]}


class WfStep:
    def __init__(self, xcoords, ycoords, env_batch: int = 1, device_id: int = 0, model: dict | None = None,
                 kernel_choice: dict | None = None, layout_of=None):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        check(self._lib.wf_create(int(device_id), C.byref(self._h)))
        self.device_id = int(device_id)
        self.num_turbines = 0
        self.env_batch = 0
        if model is not None:
            self.set_model(model)
        ragged = (not isinstance(xcoords, np.ndarray) and len(xcoords) > 0 and hasattr(xcoords[0], "__len__")
                  and len({len(r) for r in xcoords}) > 1)
        if ragged:  # layouts of different turbine counts: the handle holds the largest, shorter rows are padded (set_layouts)
            n = max(len(r) for r in xcoords)
            big = max(range(len(xcoords)), key=lambda l: len(xcoords[l]))
            self.set_layout(np.asarray(xcoords[big], np.float64), np.asarray(ycoords[big], np.float64))
            self.set_batch(env_batch)
            self.set_layouts(xcoords, ycoords, layout_of)
            self._apply_turbine_defs()
            if kernel_choice:
                self.set_kernel_choice(**kernel_choice)
            return
        xy = np.asarray(xcoords, dtype=np.float64), np.asarray(ycoords, dtype=np.float64)
        if xy[0].ndim == 2:  # several layouts in the batch: [n_layouts][n_turbines], layout_of[env_batch] (set_layouts)
            self.set_layout(xy[0][0], xy[1][0])
            self.set_batch(env_batch)
            self.set_layouts(xy[0], xy[1], layout_of)
        else:
            if layout_of is not None:
                raise ValueError("layout_of needs 2-D coordinates [n_layouts][num_turbines]")
            self.set_layout(*xy)
            self.set_batch(env_batch)
        self._apply_turbine_defs()
        if kernel_choice:  # keyword arguments of set_kernel_choice: which kernels may serve this handle
            self.set_kernel_choice(**kernel_choice)

    # -- configuration ---------------------------------------------------------------------------
    def set_model(self, model: dict):
        """Model constants + the power_thrust_table (keys of default_model()).  Two more keys describe a farm of SEVERAL
        turbine definitions (set_turbine_types): `turbine_defs`, `turbine_type_of`."""
        base = default_model()
        model = dict(model)
        self._turbine_defs = model.pop("turbine_defs", None)
        self._turbine_type_of = model.pop("turbine_type_of", None)
        if (self._turbine_defs is None) != (self._turbine_type_of is None):
            raise ValueError("turbine_defs and turbine_type_of come together")
        for k in ("alpha", "beta", "ka", "kb"):  # one gauss set given: the deflection model follows it (as in the template)
            if k in model and "defl_" + k not in model:
                model["defl_" + k] = model[k]
        unknown = set(model) - set(base)
        if unknown:
            raise ValueError(f"unknown model parameter(s): {sorted(unknown)}")
        base.update(model)
        p = ModelParams()
        for n in _lib._MODEL_DOUBLES:
            setattr(p, n, float(base[n]))
        for n in _lib._MODEL_SWITCHES:
            setattr(p, n, int(bool(base[n])))
        tws = np.ascontiguousarray(base["table_ws"], dtype=np.float64)
        tct = np.ascontiguousarray(base["table_ct"], dtype=np.float64)
        tcp = np.ascontiguousarray(base["table_cp"], dtype=np.float64)
        if not (len(tws) == len(tct) == len(tcp)):
            raise ValueError("power_thrust_table columns must have equal length")
        p.n_table = len(tws)
        dp = C.POINTER(C.c_double)
        p.table_ws, p.table_ct, p.table_cp = tws.ctypes.data_as(dp), tct.ctypes.data_as(dp), tcp.ctypes.data_as(dp)
        check(self._lib.wf_set_model(self._h, C.byref(p)), self._h)
        self._model = base
        if self.num_turbines:
            self._apply_turbine_defs()

    def _apply_turbine_defs(self):
        if getattr(self, "_turbine_defs", None) is not None:
            self.set_turbine_types(self._turbine_defs, self._turbine_type_of)
        elif self.turbine_types():
            self.set_turbine_types(None, None)

    def set_turbine_types(self, defs, type_of):
        """Several turbine definitions per farm (include/wfstep.h: wf_set_turbine_types; farm.turbine_type of a FLORIS case is
        a list): `defs` = up to 4 dicts with any of table_ws / table_ct / table_cp / tsr / pP / gen_eff / ref_density (missing:
        the model's value), `type_of[num_turbines]` the definition of each turbine.  The definitions share the rotor.  Every
        farm is then solved by the float64 kernels on every step (risk_resolve() reports 2).  None / [] clears them."""
        if not defs:
            check(self._lib.wf_set_turbine_types(self._h, 0, None, None), self._h)
            return
        base = getattr(self, "_model", None) or default_model()
        allowed = {"table_ws", "table_ct", "table_cp", "tsr", "pP", "gen_eff", "ref_density"}
        arr = (_lib.TurbineDef * len(defs))()
        keep = []
        dp = C.POINTER(C.c_double)
        for k, d in enumerate(defs):
            bad = set(d) - allowed
            if bad:
                raise ValueError(f"turbine definition {k}: {sorted(bad)} cannot differ between the definitions of one farm "
                                 f"(allowed: {sorted(allowed)})")
            cols = [np.ascontiguousarray(d.get(c, base[c]), dtype=np.float64) for c in ("table_ws", "table_ct", "table_cp")]
            if not (len(cols[0]) == len(cols[1]) == len(cols[2])):
                raise ValueError("power_thrust_table columns must have equal length")
            keep.append(cols)
            arr[k].n_table = len(cols[0])
            arr[k].table_ws, arr[k].table_ct, arr[k].table_cp = (c.ctypes.data_as(dp) for c in cols)
            for n in ("tsr", "pP", "gen_eff", "ref_density"):
                setattr(arr[k], n, float(d.get(n, base[n])))
        t = np.ascontiguousarray(type_of, dtype=np.int32)
        if t.shape != (self.num_turbines,):
            raise ValueError("turbine_type_of: one definition index per turbine of the layout")
        check(self._lib.wf_set_turbine_types(self._h, len(defs), arr, t.ctypes.data_as(C.POINTER(C.c_int))), self._h)

    def turbine_types(self) -> int:
        """Number of turbine definitions in force (0: the model's single one)."""
        n = C.c_int(0)
        check(self._lib.wf_get_turbine_types(self._h, C.byref(n)), self._h)
        return int(n.value)

    def set_layout(self, xcoords, ycoords):
        x = np.ascontiguousarray(xcoords, dtype=np.float64)
        y = np.ascontiguousarray(ycoords, dtype=np.float64)
        if x.ndim != 1 or x.shape != y.shape:
            raise ValueError("xcoords and ycoords layout coordinates must have the same length")
        check(self._lib.wf_set_layout(self._h, x.size, x.ctypes.data, y.ctypes.data), self._h)
        self.num_turbines = int(x.size)

    def set_layouts(self, xcoords, ycoords, layout_of=None, counts=None):
        """Several layouts in one batch (include/wfstep.h: wf_set_layouts): xcoords / ycoords [n_layouts][n_turbines],
        layout_of [env_batch] the layout of each farm (None: n_layouts == env_batch, farm b has layout b).  After
        set_batch; the wind has to be set again.
        Layouts of DIFFERENT turbine counts (wf_set_layouts_counts): `counts[l]` turbines of row l are real, or pass
        ragged lists of coordinates — rows are padded to the handle's turbine count; the outputs of the padding are 0."""
        if counts is None and not isinstance(xcoords, np.ndarray) and len({len(r) for r in xcoords}) > 1:
            counts = [len(r) for r in xcoords]  # ragged lists
            n = self.num_turbines
            if max(counts) > n or any(len(a) != len(b) for a, b in zip(xcoords, ycoords)):
                raise ValueError("every layout needs as many x as y coordinates, at most the handle's turbine count")
            xcoords = [list(r) + [0.0] * (n - len(r)) for r in xcoords]
            ycoords = [list(r) + [0.0] * (n - len(r)) for r in ycoords]
        x = np.ascontiguousarray(xcoords, dtype=np.float64)
        y = np.ascontiguousarray(ycoords, dtype=np.float64)
        if x.ndim != 2 or x.shape != y.shape or x.shape[1] != self.num_turbines:
            raise ValueError("xcoords and ycoords must both be [n_layouts][num_turbines]")
        cn = None
        if counts is not None:
            cn = np.ascontiguousarray(counts, dtype=np.int32)
            if cn.shape != (x.shape[0],):
                raise ValueError("counts must have one entry per layout")
        lo = None
        if layout_of is not None:
            lo = np.ascontiguousarray(layout_of, dtype=np.int32)
            if lo.shape != (self.env_batch,):
                raise ValueError("layout_of must have one entry per env")
        check(self._lib.wf_set_layouts_counts(self._h, x.shape[0], x.ctypes.data, y.ctypes.data, cn.ctypes.data if cn is not None else None,
                                              lo.ctypes.data if lo is not None else None), self._h)
        self.turbine_counts = None if cn is None else cn.copy()  # per layout; None: every layout has num_turbines

    def set_batch(self, env_batch: int):
        check(self._lib.wf_set_batch(self._h, int(env_batch)), self._h)
        self.env_batch = int(env_batch)

    def set_stream(self, hip_stream: int | None, external: bool = True):
        """Adopt an external HIP stream (int handle; 0/None = the null stream) or, with external=False,
        return to the handle's own stream."""
        check(self._lib.wf_set_stream(self._h, C.c_void_p(hip_stream or 0), int(external)), self._h)
        self._stream_key = (int(hip_stream or 0), bool(external))

    def _follow_torch_stream(self):
        """Device tensors are produced/consumed on torch's current stream: run there too, so that no
        cross-stream synchronisation is needed around our kernels."""
        import torch

        s = torch.cuda.current_stream(self.device_id).cuda_stream
        if getattr(self, "_stream_key", None) != (int(s), True):
            self.set_stream(s, True)

    def set_wind(self, wind_speed, wind_direction):
        """Scalar (shared by the batch) or one value per env; NumPy/float or torch float64 CUDA tensors.  A speed per
        env with ONE direction (size-1 `wind_direction`) keeps the shared geometry and the pair-table path."""
        if _is_torch(wind_speed):
            ws, wd = wind_speed.contiguous().reshape(-1), wind_direction.contiguous().reshape(-1)
            assert ws.dtype == wd.dtype and str(ws.dtype) == "torch.float64" and ws.is_cuda and wd.is_cuda
            self._follow_torch_stream()
            check(self._lib.wf_set_wind_counts(self._h, ws.data_ptr(), ws.numel(), wd.data_ptr(), wd.numel(), 1), self._h)
            return
        ws = np.ascontiguousarray(np.atleast_1d(wind_speed), dtype=np.float64)
        wd = np.ascontiguousarray(np.atleast_1d(wind_direction), dtype=np.float64)
        if ws.shape != wd.shape and wd.size != 1:
            raise ValueError("wind_speed and wind_direction must have the same shape (or one direction for all)")
        check(self._lib.wf_set_wind_counts(self._h, ws.ctypes.data, ws.size, wd.ctypes.data, wd.size, 0), self._h)

    def sample_wind(self, seed: int, dist: dict | None = None, direction_step: float | None = None):
        """On-device per-farm reset sampling (reference distributions by default, mdp.py:237-258).  `direction_step`
        (degrees, must divide 360; build-defined): directions rounded to that grid, farms grouped by direction, every
        step on the pair-table path (include/wfstep.h: wf_wind_sample_binned)."""
        d = None
        if dist is not None:
            base = dict(ws_scale=8.0, ws_shape=8.0, ws_lo=3.0, ws_hi=28.0, wd_mean=270.0, wd_std=20.0, wd_lo=0.0, wd_hi=360.0)
            base.update(dist)
            d = C.byref(WindDist(**base))
        s = C.c_ulonglong(int(seed) & (2**64 - 1))
        if direction_step:
            check(self._lib.wf_wind_sample_binned(self._h, s, d, float(direction_step)), self._h)
        else:
            check(self._lib.wf_wind_sample(self._h, s, d), self._h)

    def set_wind_series(self, series, start=None, seed: int = 0):
        """series: (T, 2) [speed, direction]; start: (B,) ints or None (random per farm from `seed`)."""
        ts = np.ascontiguousarray(series, dtype=np.float64)
        if ts.ndim != 2 or ts.shape[1] < 2:
            raise ValueError("wind series must have shape (T, 2): speed, direction")
        ws, wd = np.ascontiguousarray(ts[:, 0]), np.ascontiguousarray(ts[:, 1])
        st = None if start is None else np.ascontiguousarray(start, dtype=np.int32)
        if st is not None and st.shape != (self.env_batch,):
            raise ValueError("start must have one entry per farm")
        check(self._lib.wf_wind_series(self._h, ts.shape[0], ws.ctypes.data, wd.ctypes.data,
                                       None if st is None else st.ctypes.data, C.c_ulonglong(int(seed) & (2**64 - 1))), self._h)

    def wind_series_step(self):
        check(self._lib.wf_wind_series_step(self._h), self._h)

    def get_wind(self, as_torch: bool = False):
        """Current (ws, wd) of every farm: two float64 arrays of length B."""
        B = self.env_batch
        if as_torch:
            import torch

            self._follow_torch_stream()
            ws = torch.empty(B, dtype=torch.float64, device=f"cuda:{self.device_id}")
            wd = torch.empty_like(ws)
            check(self._lib.wf_get_wind(self._h, ws.data_ptr(), wd.data_ptr(), 1), self._h)
            return ws, wd
        ws, wd = np.empty(B), np.empty(B)
        check(self._lib.wf_get_wind(self._h, ws.ctypes.data, wd.ctypes.data, 0), self._h)
        return ws, wd

    # -- the step ----------------------------------------------------------------------------------
    def step(self, yaw, out: dict | None = None):
        """yaw: (B, N) absolute degrees.  Returns dict(power, wind_speed, wind_direction, load)."""
        B, N = self.env_batch, self.num_turbines
        if _is_torch(yaw):
            import torch

            assert yaw.is_cuda and yaw.dtype == torch.float32 and yaw.numel() == B * N
            yaw = yaw.contiguous()
            self._follow_torch_stream()
            if out is None:
                out = {
                    "power": torch.empty((B, N), device=yaw.device, dtype=torch.float32),
                    "wind_speed": torch.empty((B, N), device=yaw.device, dtype=torch.float32),
                    "wind_direction": torch.empty((B, N), device=yaw.device, dtype=torch.float32),
                    "load": torch.empty((B, N, 4), device=yaw.device, dtype=torch.float32),
                }
            check(self._lib.wf_step(self._h, yaw.data_ptr(), out["power"].data_ptr(), out["wind_speed"].data_ptr(),
                                    out["wind_direction"].data_ptr(), out["load"].data_ptr(), 1), self._h)
            return out
        yaw = np.ascontiguousarray(yaw, dtype=np.float32).reshape(B, N)
        if out is None:
            out = {
                "power": np.empty((B, N), np.float32),
                "wind_speed": np.empty((B, N), np.float32),
                "wind_direction": np.empty((B, N), np.float32),
                "load": np.empty((B, N, 4), np.float32),
            }
        check(self._lib.wf_step(self._h, yaw.ctypes.data, out["power"].ctypes.data, out["wind_speed"].ctypes.data,
                                out["wind_direction"].ctypes.data, out["load"].ctypes.data, 0), self._h)
        return out

    # -- fused env step ---------------------------------------------------------------
    def env_config(self, yaw_lo=-40.0, yaw_hi=40.0, yaw_step=5.0, actuator_rate=0.3, dt=60.0, budget=0.1,
                   load_coef=0.1, discrete=False, power_mw=False):
        """power_mw: the `power` output of env_step in MW (include/wfstep.h: wf_env_set_power_unit) instead of W."""
        p = EnvParams(yaw_lo, yaw_hi, yaw_step, actuator_rate, dt, budget, load_coef, int(bool(discrete)))
        check(self._lib.wf_env_config(self._h, C.byref(p)), self._h)
        check(self._lib.wf_env_set_power_unit(self._h, int(bool(power_mw))), self._h)

    def env_reset(self):
        check(self._lib.wf_env_reset(self._h), self._h)

    def env_set_prev_wind(self, wind_speed):
        """Free-stream speed (B,) of the state before the coming env step, when it differs from the current wind
        (include/wfstep.h: wf_env_set_prev_wind); used once."""
        ws = np.ascontiguousarray(np.broadcast_to(np.asarray(wind_speed, np.float64), (self.env_batch,)))
        check(self._lib.wf_env_set_prev_wind(self._h, ws.ctypes.data, 0), self._h)

    def env_get_state(self, as_torch: bool = False) -> dict:
        """Copy of the device-resident env state: yaw (B, N), acc (B, N), moves (B,) — host NumPy arrays, or torch CUDA
        tensors (device-to-device, asynchronous on torch's current stream) with as_torch=True."""
        B, N = self.env_batch, self.num_turbines
        if as_torch:
            import torch

            self._follow_torch_stream()
            dev = f"cuda:{self.device_id}"
            st = {"yaw": torch.empty((B, N), dtype=torch.float32, device=dev), "acc": torch.empty((B, N), dtype=torch.float32, device=dev),
                  "moves": torch.empty(B, dtype=torch.int32, device=dev)}
            check(self._lib.wf_env_state(self._h, st["yaw"].data_ptr(), st["acc"].data_ptr(), st["moves"].data_ptr(), 0, 1), self._h)
            return st
        st = {"yaw": np.empty((B, N), np.float32), "acc": np.empty((B, N), np.float32), "moves": np.empty(B, np.int32)}
        check(self._lib.wf_env_state(self._h, st["yaw"].ctypes.data, st["acc"].ctypes.data, st["moves"].ctypes.data, 0, 0), self._h)
        return st

    def env_set_state(self, state: dict):
        B, N = self.env_batch, self.num_turbines
        yaw = np.ascontiguousarray(state["yaw"], np.float32).reshape(B, N)
        acc = np.ascontiguousarray(state["acc"], np.float32).reshape(B, N)
        moves = np.ascontiguousarray(state["moves"], np.int32).reshape(B)
        check(self._lib.wf_env_state(self._h, yaw.ctypes.data, acc.ctypes.data, moves.ctypes.data, 1, 0), self._h)

    def env_step(self, action=None, want=("reward", "yaw", "power", "wind_speed", "wind_direction", "load"), out=None):
        """One fused env step.  `action` (B, N) torch CUDA float32 tensor or NumPy array, or None for a solve
        at the current yaw (reset warm-up).  `want` selects which outputs are produced at all."""
        B, N = self.env_batch, self.num_turbines
        shapes = {"reward": (B,), "yaw": (B, N), "power": (B, N), "wind_speed": (B, N), "wind_direction": (B, N),
                  "load": (B, N, 4)}
        order = ("reward", "yaw", "power", "wind_speed", "wind_direction", "load")
        use_torch = _is_torch(action) or (action is None and out is not None and any(_is_torch(v) for v in out.values()))
        if use_torch:
            import torch

            dev = action.device if action is not None else next(iter(out.values())).device
            self._follow_torch_stream()
            if action is not None:
                assert action.is_cuda and action.dtype == torch.float32 and action.numel() == B * N
                action = action.contiguous()
            out = dict(out or {})
            for k in want:
                if k not in out:
                    out[k] = torch.empty(shapes[k], device=dev, dtype=torch.float32)
            ptrs = [out[k].data_ptr() if k in want else None for k in order]
            check(self._lib.wf_env_step(self._h, action.data_ptr() if action is not None else None, *ptrs, 1), self._h)
            return {k: out[k] for k in want}
        if action is not None:
            action = np.ascontiguousarray(action, dtype=np.float32).reshape(B, N)
        out = dict(out or {})
        for k in want:
            if k not in out:
                out[k] = np.empty(shapes[k], np.float32)
        ptrs = [out[k].ctypes.data if k in want else None for k in order]
        check(self._lib.wf_env_step(self._h, action.ctypes.data if action is not None else None, *ptrs, 0), self._h)
        return {k: out[k] for k in want}

    def sync(self):
        check(self._lib.wf_sync(self._h), self._h)

    def set_risk_guard(self, rel_band: float):
        """Relative half-width of the guard band around the overlap threshold (include/wfstep.h, WF_RISK_OVERLAP)."""
        check(self._lib.wf_set_risk_guard(self._h, float(rel_band)), self._h)

    def set_risk_resolve(self, mode: int | bool = 1):
        """Float64 re-solve of the farms the float32 kernels flag (include/wfstep.h: wf_set_risk_resolve): 1 / True the
        flagged farms (the DEFAULT of a new handle: every farm then meets the parity tolerances and its flag is 0), 2 every
        farm, 0 / False off (float32 only: flagged farms within the per-flag bounds of include/wfstep.h)."""
        check(self._lib.wf_set_risk_resolve(self._h, int(mode)), self._h)

    def risk_resolve(self) -> int:
        """The handle's re-solve mode (wf_get_risk_resolve): 0 off, 1 flagged farms, 2 every farm."""
        m = C.c_int(0)
        check(self._lib.wf_get_risk_resolve(self._h, C.byref(m)), self._h)
        return int(m.value)

    def resolve_stats(self) -> dict:
        """{"n_resolved": farms the last step solved in float64, "raw_flags": int32 (B,) flags before they were cleared}.
        Only meaningful after a step with the re-solve on: with it off (set_risk_resolve(0)) nothing is recorded and
        this raises instead of returning stale flags — read risk_flags() there."""
        if not self.risk_resolve():
            raise RuntimeError("resolve_stats() needs the float64 re-solve on (set_risk_resolve(1), the default); with it off "
                               "the flags of the last step are risk_flags()")
        n = C.c_int(0)
        raw = np.empty(self.env_batch, np.int32)
        check(self._lib.wf_get_resolve_stats(self._h, C.byref(n), raw.ctypes.data, 0), self._h)
        return {"n_resolved": int(n.value), "raw_flags": raw}

    def risk_flags(self, as_torch: bool = False):
        """WF_RISK_* bits of every farm for the last step: int32 (B,).  0 = every float64 decision of the reference
        was reproduced with a margin; see include/wfstep.h."""
        B = self.env_batch
        if as_torch:
            import torch

            self._follow_torch_stream()
            f = torch.empty(B, dtype=torch.int32, device=f"cuda:{self.device_id}")
            check(self._lib.wf_get_risk_flags(self._h, f.data_ptr(), 1), self._h)
            return f
        f = np.empty(B, np.int32)
        check(self._lib.wf_get_risk_flags(self._h, f.ctypes.data, 0), self._h)
        return f

    def timing_begin(self):
        check(self._lib.wf_timing_begin(self._h), self._h)

    def timing_end(self) -> float:
        ms = C.c_float()
        check(self._lib.wf_timing_end(self._h, C.byref(ms)), self._h)
        return float(ms.value)

    def set_kernel_choice(self, slot=None, one_block=None, pair_table=None, fly_one_block=None, far_skip=None, calibrate=None, mixed=None):
        """Which kernels may serve THIS handle (include/wfstep.h: wf_set_kernel_choice); None = automatic.
          slot=(G, S) or "16x5"      wf_step_kernel<G,S>
          one_block=False            never wf_step_ll_kernel;  one_block=(G, S) / "4x2" / "8": always, with that shape
          pair_table=False           everything on the fly
          fly_one_block=False        a wind per farm stays on wf_step_kernel
          far_skip=False             wf_step_ll_kernel evaluates every (source, target) pair (no far-source / far-pair skip)
          calibrate=False            the rounds model's guess stands: no timing of the kernel families before the first step
          mixed=False                always ONE launch per step (no whole-rounds + remainder split of the batch)
        Drops the current wind: set it again before the next step."""
        def gs(v, default_s=1):
            if isinstance(v, bool):
                raise ValueError("a kernel shape is (G, S), 'GxS' or G — not a bool (one_block=False disables the one-block "
                                 "kernel, None leaves the choice to the rounds model)")
            if isinstance(v, str):
                parts = v.lower().split("x")
                return int(parts[0]), (int(parts[1]) if len(parts) > 1 else default_s)
            if isinstance(v, int):
                return v, default_s
            return int(v[0]), int(v[1])

        c = KernelChoice(0, 0, -1, 0, 0, -1, -1, -1, -1, -1)
        if slot:
            c.slot_G, c.slot_S = gs(slot)
        if one_block is not None:
            if one_block is False or (not isinstance(one_block, bool) and one_block == 0):
                c.one_block = 0
            else:
                c.one_block = 1
                c.ll_G, c.ll_S = gs(one_block)
        if pair_table is not None:
            c.pair_table = int(bool(pair_table)) if pair_table is False else -1
        if fly_one_block is not None:
            c.fly_one_block = 0 if fly_one_block is False else -1
        if far_skip is not None:
            c.far_skip = 0 if far_skip is False or far_skip == 0 else -1
        if calibrate is not None:
            c.calibrate = 0 if calibrate is False or calibrate == 0 else -1
        if mixed is not None:
            c.mixed = 0 if mixed is False or mixed == 0 else -1
        check(self._lib.wf_set_kernel_choice(self._h, C.byref(c)), self._h)

    def kernel_choice(self) -> dict:
        c = KernelChoice()
        check(self._lib.wf_get_kernel_choice(self._h, C.byref(c)), self._h)
        return {n: getattr(c, n) for n, _ in KernelChoice._fields_}

    def calibration(self) -> dict:
        """What the per-handle kernel calibration found (include/wfstep.h: wf_get_calibration): {"shape": "2x2" / "16x5-slot
        kernel" ... or None when it has not run, "family_ms": {family: ms per launch} of the families it timed; "on_the_fly":
        which kernel the timing on the on-the-fly path (a wind per farm) kept, None before it ran}."""
        code = C.c_int(-1)
        ms = (C.c_float * 6)()
        check(self._lib.wf_get_calibration(self._h, C.byref(code), ms), self._h)
        names = ("slot", "8x1", "4x2", "4x1", "2x2", "16x1")
        shape = None if code.value < 0 else ("slot" if code.value == 0 else f"{code.value >> 4}x{code.value & 15}")
        fly, fms = C.c_int(0), (C.c_float * 2)()
        check(self._lib.wf_get_fly_calibration(self._h, C.byref(fly), fms), self._h)
        mixn, mixms = C.c_int(0), C.c_float(0.0)
        check(self._lib.wf_get_mixed_launch(self._h, C.byref(mixn), C.byref(mixms)), self._h)
        return {"shape": shape, "family_ms": {n: float(m) for n, m in zip(names, ms) if m > 0.0},
                "on_the_fly": (None, "one_block", "slot")[fly.value], "on_the_fly_ms": {n: float(m) for n, m in zip(("one_block", "slot"), fms) if m > 0.0},
                "mixed_main_farms": int(mixn.value), "mixed_ms": float(mixms.value)}

    def calibrate(self):
        """Time the kernel families NOW for the current layout / batch / wind (include/wfstep.h: wf_calibrate; synchronises) —
        otherwise the first step of a configuration does it (or takes the process-wide cached result)."""
        check(self._lib.wf_calibrate(self._h), self._h)

    _FAMILY_CODES = {"slot": 0, "8x1": (8 << 4) | 1, "4x2": (4 << 4) | 2, "4x1": (4 << 4) | 1, "2x2": (2 << 4) | 2, "16x1": (16 << 4) | 1}

    def set_calibration(self, shape=None, on_the_fly=None):
        """Take a saved calibration as is — `shape` and `on_the_fly` as calibration() returned them (None: leave that one
        to the timing); nothing is timed for this configuration afterwards (wf_set_calibration)."""
        code = -1 if shape is None else self._FAMILY_CODES[shape]
        fly = {None: 0, "one_block": 1, "slot": 2}[on_the_fly]
        check(self._lib.wf_set_calibration(self._h, code, fly), self._h)

    def kernel_info(self) -> dict:
        k = KernelInfo()
        check(self._lib.wf_get_kernel_info(self._h, C.byref(k)), self._h)
        return {n: getattr(k, n) for n, _ in KernelInfo._fields_}

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.wf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
