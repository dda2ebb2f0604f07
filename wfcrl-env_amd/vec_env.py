"""`VecWindFarmEnv` — B independent wind-farm envs stepped by ONE fused kernel launch.

Batched restatement of the reference's centralised env: reset wind sampling (wfcrl/mdp.py:233-271),
actuation budget (simple_env.py:64-72), yaw transition (mdp.py:291-319), reward (simple_env.py:78-85),
truncation bookkeeping (interface.py:578-586, mdp.py:261-262).  Env state (yaw, actuation accumulators,
move counters) lives on the device; actions and observations are torch CUDA tensors (NumPy accepted).
All B envs share the horizon, so they truncate together and are reset together.

Differences from running B reference envs, by construction of a batch: `reset` draws all winds from ONE
`default_rng(seed)` (vectorised weibull then normal draws; for B = 1 this is exactly the reference's
stream), and constrained actions are not zeroed in the caller's array (the gate is applied on the device).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from ._compat import spaces
from .backend import WfStep
from .mdp import WD_MEAN, WD_STD, WEIBULL_SCALE, WEIBULL_SHAPE, WindFarmMDP
from .rewards import DoNothingReward, ReferencePercentage, StepPercentage

_OBS_KEYS = ("yaw", "freewind_measurements", "wind_speed", "wind_direction")


class VecWindFarmEnv:
    metadata = {"name": "vectorized-centralized-windfarm"}

    def __init__(self, farm_case, controls: dict = None, env_batch: int = 1, continuous_control: bool = True,
                 reward_shaper=None, start_iter: int = 0, max_num_steps: int = 500, load_coef: float = 0.1,
                 device_id: int = 0, model: dict = None, return_torch: bool = True, backend=None,
                 wind_sampling: str = "host", reuse_buffers: bool = True, wind_direction_step: float = None,
                 actuation_budget: float = 0.1, kernel_choice: dict = None, risk_resolve: bool = None,
                 layouts: dict = None):
        controls = {"yaw": (-40, 40, 5)} if controls is None else dict(controls)
        if list(controls) != ["yaw"]:
            raise ValueError(f"Cannot control {list(controls)}. Interface HipFlorisInterface only allows for the "
                             "following: ['yaw']")
        spec = controls["yaw"]
        if not (len(spec) in (2, 3) and spec[0] < spec[1]):
            raise ValueError("Wrong bounds for actuator yaw: ensure that lower_bound < upper_bound")
        if len(spec) == 2:
            spec = tuple(spec) + (1,)
        self.controls = {"yaw": tuple(spec)}
        self.farm_case = farm_case.clone() if hasattr(farm_case, "clone") else farm_case
        self.num_envs = int(env_batch)
        self.num_turbines = self.farm_case.num_turbines
        self.continuous_control = continuous_control
        self.max_num_steps = max_num_steps
        self.start_iter = start_iter
        self.load_coef = load_coef
        self.dt = self.farm_case.dt
        self.reward_shaper = DoNothingReward() if reward_shaper is None else reward_shaper
        self.return_torch = return_torch
        self.farm_case.max_iter = start_iter + max_num_steps
        p = self.farm_case.simul_params
        self.fi = backend if backend is not None else WfStep(p["xcoords"], p["ycoords"], env_batch=self.num_envs,
                                                              device_id=device_id, model=model, kernel_choice=kernel_choice)
        # risk_resolve: every farm the float32 kernel flags is solved again in float64 behind each step (wf_set_risk_resolve)
        # — the reference computes every step in float64 (interface.py:564), and north_star's 1e-4 holds on EVERY farm only
        # with it.  ON unless told otherwise (it is the default of the handle itself, ABI 6); a caller's own `backend=` is left as
        # it was configured unless risk_resolve is given — one set up for mode 2 (every farm in float64) stays in mode 2.  Nothing measurable where no
        # farm is flagged (a shared 270 deg wind on HornsRev1); about +1 ms per step where a wind per farm flags ~1 % of the
        # batch.  risk_resolve=False is the opt-out for throughput runs that accept the per-flag bounds of
        # include/wfstep.h on flagged farms.
        if risk_resolve is not None or backend is None:
            self.fi.set_risk_resolve(0 if risk_resolve is False else 1)
        # layouts: several layouts in the batch — dict(xcoords=[K][N], ycoords=[K][N], layout_of=[env_batch] or None for
        # K == env_batch[, counts=[K]]), each with the case's number of turbines or — with `counts`, or ragged lists of
        # coordinates — fewer (backend.WfStep.set_layouts: observations, powers and loads of the missing turbines are 0,
        # the reward averages over the real ones); the case's own layout is then only the default the handle returns to
        if layouts is not None:
            self.fi.set_layouts(layouts["xcoords"], layouts["ycoords"], layouts.get("layout_of"), layouts.get("counts"))
        self.fi.env_config(yaw_lo=spec[0], yaw_hi=spec[1], yaw_step=spec[2],
                           actuator_rate=WindFarmMDP.ACTUATORS_RATE["yaw"], dt=self.dt, budget=actuation_budget,
                           load_coef=load_coef, discrete=not continuous_control, power_mw=True)
        n = self.num_turbines
        # per-env spaces, identical to the reference's (mdp.py:108-153)
        if continuous_control:
            self.single_action_space = spaces.Dict({"yaw": spaces.Box(-spec[2], spec[2], shape=(n,))})
        else:
            self.single_action_space = spaces.Dict({"yaw": spaces.MultiDiscrete([3] * n)})
        ones = np.ones(n, dtype=np.float32)
        b = WindFarmMDP.DEFAULT_BOUNDS
        self.single_observation_space = spaces.Dict(OrderedDict([
            ("yaw", spaces.Box(ones * spec[0], ones * spec[1], shape=(n,))),
            ("freewind_measurements", spaces.Box(np.array([b["wind_speed"][0], b["wind_direction"][0]], np.float32),
                                                 np.array([b["wind_speed"][1], b["wind_direction"][1]], np.float32),
                                                 shape=(2,))),
            ("wind_speed", spaces.Box(ones * b["wind_speed"][0], ones * b["wind_speed"][1], shape=(n,))),
            ("wind_direction", spaces.Box(ones * b["wind_direction"][0], ones * b["wind_direction"][1], shape=(n,))),
        ]))
        self.action_space = self.single_action_space
        self.observation_space = self.single_observation_space
        if wind_sampling not in ("host", "device"):
            raise ValueError("wind_sampling must be 'host' (NumPy stream of the reference) or 'device' (Philox on the GPU)")
        self.wind_sampling = wind_sampling
        # build-defined option of device sampling: reset directions rounded to a grid of that many degrees (must divide
        # 360), so that every step stays on the pair-table path (backend.WfStep.sample_wind)
        self.wind_direction_step = wind_direction_step
        ts = self.farm_case.wind_time_series
        has_series = ts is not None and (ts.size > 0 if isinstance(ts, np.ndarray) else bool(ts))
        if has_series:
            from .interface import _load_time_series

            self._series = np.ascontiguousarray(_load_time_series(ts)[:, :2], dtype=np.float64)
        else:
            self._series = None
        self._num_iter = 0
        self._freewind = None
        self._shaper_ref = None
        # reuse_buffers=True (the DEFAULT since round 6; torch path): the output tensors of `step` / `step_light` are two
        # preallocated sets used alternately — what step t returned stays valid until step t + 2 overwrites it: enough for
        # (obs, next_obs) pairs; ALIASING: anything that must live longer (rollout storage, loggers, a list of observations)
        # has to be copied (`.clone()`) by the caller, or the env built with reuse_buffers=False, which allocates fresh
        # tensors every step as B reference envs would (+3 % per step on the headline batch).  The logged / AEC flavours
        # (environments.make(..., log=True), Dec_*) keep what they return and therefore build their inner env without reuse.
        self.reuse_buffers = bool(reuse_buffers) and return_torch
        self._bufs = [{}, {}]
        self._flip = 0
        self._const = {}

    # -- helpers ------------------------------------------------------------------------------------
    def _sample_wind(self, seed, options):
        rng = np.random.default_rng(seed)
        B = self.num_envs
        lo, hi = self.single_observation_space["freewind_measurements"].low, \
            self.single_observation_space["freewind_measurements"].high
        if options is not None and "wind_speed" in options:
            ws = np.broadcast_to(np.asarray(options["wind_speed"], np.float64), (B,)).copy()
        elif self.farm_case.set_wind_speed:
            ws = np.full(B, float(self.farm_case.simul_params["speed"]))
        else:
            ws = np.clip(WEIBULL_SCALE * rng.weibull(WEIBULL_SHAPE, B), lo[0], hi[0])
        if options is not None and "wind_direction" in options:
            wd = np.broadcast_to(np.asarray(options["wind_direction"], np.float64), (B,)).copy()
        elif self.farm_case.set_wind_direction:
            wd = np.full(B, float(self.farm_case.simul_params["direction"]))
        else:
            wd = np.clip(rng.normal(WD_MEAN, WD_STD, B) % 360, lo[1], hi[1])
        return ws, wd % 360

    def _refresh_freewind(self):
        if self.return_torch:
            import torch

            ws, wd = self.fi.get_wind(as_torch=True)
            self._freewind = torch.stack([ws, wd], dim=1)
        else:
            ws, wd = self.fi.get_wind()
            self._freewind = np.stack([ws, wd], axis=1)

    def _obs(self, out):
        obs = OrderedDict()
        obs["yaw"] = out["yaw"]
        obs["freewind_measurements"] = self._freewind
        obs["wind_speed"] = out["wind_speed"]
        obs["wind_direction"] = out["wind_direction"]
        return obs

    def _next_buf(self):
        if not self.reuse_buffers:
            return None
        self._flip ^= 1
        return self._bufs[self._flip]

    def _to_device(self, a):
        if not self.return_torch:
            return np.ascontiguousarray(a, dtype=np.float32)
        import torch

        if isinstance(a, torch.Tensor):
            return a.to(device=f"cuda:{self.fi.device_id}", dtype=torch.float32)
        return torch.as_tensor(np.asarray(a, dtype=np.float32), device=f"cuda:{self.fi.device_id}")

    def _shape(self, r):
        """Batched reward shaping with the semantics of reference wfcrl/rewards.py:4-46, per env.  Dispatch is by class
        name, so a shaper instance built from the `wfcrl.rewards` alias of this package is recognised as well."""
        s = self.reward_shaper
        kind = type(s).__name__
        if kind == "DoNothingReward":
            return r
        if kind == "ReferencePercentage":
            return (r - s.reference) / s.reference
        if kind == "StepPercentage":
            # per-env previous reward, seeded from the shaper's own `reference` (constructor / reset value); a previous
            # reward of exactly 0 yields 0.0 (rewards.py:36-37), and the shaper object keeps the state like the
            # reference's does
            ref = self._shaper_ref
            if ref is None:
                base = s.reference
                ref = base if np.ndim(base) > 0 else r * 0 + float(base)
            zero = ref == 0
            safe = ref + zero  # 1 where the previous reward is 0: no division by zero, result replaced below
            shaped = (r - ref) / safe
            shaped = shaped * (~zero) if hasattr(shaped, "clone") else np.where(zero, 0.0, shaped)
            self._shaper_ref = r.clone() if hasattr(r, "clone") else r.copy()
            s.reference = self._shaper_ref
            return shaped
        return s(r)

    # -- episode ------------------------------------------------------------------------------------
    def reset(self, seed=None, options=None):
        """Samples one wind per env, zeroes the env state, runs the warm-up solve(s) at yaw = 0 and returns
        the start observation (clipped to the observation space like the reference's, mdp.py:263-266)."""
        if self._series is not None:
            # time-series mode: requested / sampled wind is ignored (interface.py:588-600); every farm starts at a
            # random row (interface.py:516-518), then each solve consumes one row
            s = seed if seed is not None else int(np.random.randint(0, 2**31 - 1))
            self.fi.set_wind_series(self._series, start=None, seed=s)
        elif self.wind_sampling == "device" and not (options and ("wind_speed" in options or "wind_direction" in options)) \
                and not (self.farm_case.set_wind_speed or self.farm_case.set_wind_direction):
            s = seed if seed is not None else int(np.random.randint(0, 2**31 - 1))
            self.fi.sample_wind(s, direction_step=self.wind_direction_step)
        else:
            ws, wd = self._sample_wind(seed, options)
            if np.all(ws == ws[0]) and np.all(wd == wd[0]):
                self.fi.set_wind(ws[0], wd[0])  # one wind for the whole batch: shared-wind (pair table) path
            else:
                self.fi.set_wind(ws, wd)
        self.fi.env_reset()
        self.reward_shaper.reset()
        self._shaper_ref = None
        seed_out = None
        if self.return_torch:
            import torch

            seed_out = {"yaw": torch.empty((self.num_envs, self.num_turbines), device=f"cuda:{self.fi.device_id}")}
        self._num_iter = 0
        out = None
        for _ in range(self.start_iter + 1):
            if self._series is not None:
                self.fi.wind_series_step()
            out = self.fi.env_step(None, want=("yaw", "wind_speed", "wind_direction"), out=seed_out)
            self._num_iter += 1
        self._refresh_freewind()
        obs = self._obs(out)
        sp = self.single_observation_space
        for k in ("wind_speed", "wind_direction"):
            lo, hi = float(sp[k].low[0]), float(sp[k].high[0])
            obs[k] = obs[k].clamp(lo, hi) if self.return_torch else np.clip(obs[k], lo, hi)
        # the whole start state is clipped to the observation bounds (mdp.py:263-266), the free wind included; the first
        # reward is normalised by that clipped speed (simple_env.py:78-80 reads the state before the step)
        lo, hi = sp["freewind_measurements"].low, sp["freewind_measurements"].high
        fw = obs["freewind_measurements"]
        if self.return_torch:
            import torch

            clipped = torch.minimum(torch.maximum(fw, torch.as_tensor(lo, dtype=fw.dtype, device=fw.device)),
                                    torch.as_tensor(hi, dtype=fw.dtype, device=fw.device))
            changed = bool((clipped[:, 0] != fw[:, 0]).any()) if self._series is None else False
        else:
            clipped = np.clip(fw, lo, hi)
            changed = bool((clipped[:, 0] != fw[:, 0]).any()) if self._series is None else False
        obs["freewind_measurements"] = clipped
        if changed:
            self.fi.env_set_prev_wind(clipped[:, 0].cpu().numpy() if self.return_torch else clipped[:, 0])
        return obs

    def step(self, actions):
        """actions: {"yaw": (B, N)} or the (B, N) array itself.  Returns (obs, reward[B], terminated[B],
        truncated[B], info) with info["power"] in MW and info["load"] (B, N, 4) — units of the reference."""
        a = actions["yaw"] if isinstance(actions, dict) else actions
        if self._series is not None:
            self.fi.wind_series_step()  # ValueError("wind series exhausted") ~ the reference's StopIteration
            self._refresh_freewind()
        buf = self._next_buf()
        out = self.fi.env_step(self._to_device(a), out=buf)
        self._num_iter += 1
        truncated = self._num_iter == self.farm_case.max_iter
        reward = self._shape(out["reward"])
        if self.return_torch:
            import torch

            if buf is not None:
                buf.update(out)  # keep the tensors env_step allocated on first use
            info = {"power": out["power"], "load": out["load"]}  # (MW out of the kernel: wf_env_set_power_unit)
            if not self._const:
                dev = out["reward"].device
                self._const = {False: torch.zeros(self.num_envs, dtype=torch.bool, device=dev),
                               True: torch.ones(self.num_envs, dtype=torch.bool, device=dev)}
            trunc, term = self._const[bool(truncated)], self._const[False]
        else:
            info = {"power": out["power"], "load": out["load"]}
            trunc = np.full(self.num_envs, bool(truncated))
            term = np.zeros(self.num_envs, bool)
        return self._obs(out), reward, term, trunc, info

    def step_light(self, actions):
        """Learner-facing variant: only reward + local wind observations leave the kernel (no power/load
        arrays are written: 2N+1 instead of 7N floats per env)."""
        a = actions["yaw"] if isinstance(actions, dict) else actions
        if self._series is not None:
            self.fi.wind_series_step()
            self._refresh_freewind()
        buf = self._next_buf()
        out = self.fi.env_step(self._to_device(a), want=("reward", "yaw", "wind_speed", "wind_direction"), out=buf)
        if buf is not None:
            buf.update(out)
        self._num_iter += 1
        return self._obs(out), self._shape(out["reward"]), self._num_iter == self.farm_case.max_iter

    # -- checkpoint / resume (the env state is tiny; FLORIS itself is stateless between steps) ------
    def get_state(self) -> dict:
        """Everything needed to resume the batch: device env state, the per-farm wind, the step counter."""
        if self._series is not None:
            raise NotImplementedError("checkpointing a time-series episode is not supported")
        ws, wd = self.fi.get_wind()
        return {**self.fi.env_get_state(), "wind_speed": ws, "wind_direction": wd, "num_iter": self._num_iter,
                "shaper_ref": None if self._shaper_ref is None else np.asarray(
                    self._shaper_ref.cpu() if hasattr(self._shaper_ref, "cpu") else self._shaper_ref).copy()}

    def set_state(self, state: dict):
        self.fi.set_wind(state["wind_speed"], state["wind_direction"])
        self.fi.env_set_state(state)
        self._num_iter = int(state["num_iter"])
        ref = state.get("shaper_ref")
        self._shaper_ref = None if ref is None else self._to_device(ref)
        self._refresh_freewind()

    def close(self):
        self.fi.close()


def make_vec(env_id: str, env_batch: int, controls=("yaw",), **kw):
    """`make(env_id, env_batch=B)` shortcut."""
    from .environments import make

    return make(env_id, controls=list(controls) if not isinstance(controls, dict) else controls, env_batch=env_batch,
                **kw)
