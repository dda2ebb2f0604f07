"""GPU box: randomised fuzz of the handle's state machine — long random sequences of wf_set_layout / wf_set_batch /
wf_set_model / wf_set_wind (shared, per farm, device pointers) / wf_wind_sample / wf_wind_series(+_step) /
wf_env_reset / wf_set_turbine_types interleaved with wf_step (host and device buffers) and wf_env_step, every result checked against the
float64 oracle evaluated on the state the sequence should have produced (stale geometry, a stale pair table, a stale
kernel variant or stale env state all show up as a mismatch).
usage: python tests/tools/fuzz_api.py [n_sessions] [ops_per_session] [seed]
FUZZ_API_BIG=1: farms of 33+ turbines in batches of 4200-9000 — beyond the latency regime, so that the handle's own timing of
the kernel families (wf_kernel_choice::calibrate: before the first launch of a configuration, table path and on the fly) fires
inside the sessions, also inside fused env steps and after every reconfiguration; a sample of 192 farms per check goes to the
oracle (tools/round_close.sh runs one such seed on every round's HEAD)."""
import os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np

from fuzz_parity import classify, make_layout, parity, worst  # noqa: E402  (same directory)


def mdp_step_f32(st, action, p):
    """One float32 MDP transition as the reference performs it (simple_env.py:64-72, mdp.py:291-319)."""
    yaw, acc, moves = st["yaw"].copy(), st["acc"].copy(), st["moves"].copy()
    moves_new = moves + 1
    f32 = np.float32
    frac = ((acc / f32(p["actuator_rate"])) / moves_new[:, None].astype(f32)) / f32(p["dt"])
    a = np.where(frac >= f32(p["budget"]), f32(0), action.astype(f32))
    if p["discrete"]:
        a = (a - f32(1)) * f32(p["yaw_step"])
    else:
        a = np.clip(a, -f32(p["yaw_step"]), f32(p["yaw_step"]))
    yaw_new = np.clip(yaw + a, f32(p["yaw_lo"]), f32(p["yaw_hi"])).astype(f32)
    return yaw_new


def run(n_sessions, n_ops, seed):
    import torch

    from oracle import c_oracle
    from oracle.floris_gch_numpy import ModelParams
    from wfcrl_env_amd.backend import WfStep

    rng = np.random.default_rng(seed)
    nbad = nflip = nchecks = 0
    BIG = bool(os.environ.get("FUZZ_API_BIG"))

    def new_layout():
        while True:
            xx, yy = make_layout(rng)
            if not BIG or 33 <= xx.size <= 100:
                return xx, yy

    def new_batch(small_hi=20):
        return int(rng.integers(4200, 9000)) if BIG else int(rng.integers(1, small_hi))

    ncal = ntyped = 0
    for sess in range(n_sessions):
        x, y = new_layout()
        B = new_batch()
        sub = rng.choice(B, 192, replace=False) if BIG else None  # farms whose results go to the oracle (None: all)
        # every other session forces the one-block-at-a-time kernel (table path and on the fly) where the farm has more
        # than one of its blocks; the batch sizes of this fuzzer would never pick it by themselves
        llg = str(rng.choice(["", "", "8", "4x2", "4", "2x2"]))
        if BIG:
            llg = str(rng.choice(["", "", "", "4x2"]))  # mostly the handle's own pick + calibration
        w = WfStep(x, y, env_batch=B, kernel_choice=dict(one_block=llg) if llg else None)
        resolve_on = False
        mp, model = None, {}
        typed = None  # (definitions as the oracle takes them, definition of each turbine) while wf_set_turbine_types is in force
        envp = dict(yaw_lo=-40.0, yaw_hi=40.0, yaw_step=5.0, actuator_rate=0.3, dt=60.0, budget=0.1, load_coef=0.1, discrete=False)
        w.env_config(**envp)
        w.set_wind(8.0, 270.0)
        w.env_reset()
        log = []

        multi = None  # (X [K][N], Y, layout_of [B]) after wf_set_layouts; None: the one layout (x, y)

        def S(a):
            return a if sub is None else a[sub]

        def oracle(yaw64):
            """reference of the farms `sub` (all of them unless FUZZ_API_BIG)"""
            ws, wd = w.get_wind()
            mp = cur_mp()
            ws, wd, yaw64 = S(ws), S(wd), S(yaw64)
            if multi is None:
                return c_oracle.farm_step_batch(x, y, ws, wd, yaw64, mp, margin=True), ws
            out = None
            for l in range(multi[0].shape[0]):
                idx = np.flatnonzero(S(multi[2]) == l)
                if idx.size == 0:
                    continue
                r = c_oracle.farm_step_batch(multi[0][l], multi[1][l], ws[idx], wd[idx], yaw64[idx], mp, margin=True)
                if out is None:
                    out = {k: np.zeros((yaw64.shape[0],) + np.asarray(v).shape[1:], np.asarray(v).dtype) for k, v in r.items()}
                for k, v in r.items():
                    out[k][idx] = v
            return out, ws

        def cur_mp():
            if typed is None:
                return mp
            import dataclasses

            return dataclasses.replace(mp or ModelParams(), turbine_defs=typed[0], turbine_type_of=list(typed[1]))

        last_yaw = None

        def check(tag, got, ref):
            nonlocal nbad, nflip, nchecks, ntyped
            nchecks += 1
            ntyped += typed is not None
            got = {k: S(v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in got.items()}
            flags_now = S(w.risk_flags())
            r = worst(got, ref, flags_now)
            k = classify(r)
            if resolve_on or typed is not None:  # the float64 re-solve is on (several turbine definitions: it solves every farm): every farm strict, no flag left — anything else is a violation
                e = parity.errors(got, ref)
                if not parity.within(e, parity.TOL, x.size).all() or flags_now.any():
                    k = "BAD"
                elif k == "flip":
                    k = "ok"
            nflip += k == "flip"
            if k == "BAD":
                nbad += 1
                print("BAD", dict(session=sess, seed=seed, N=x.size, B=B, tag=tag, info=w.kernel_info()), r, "\n   ops:", log[-12:], flush=True)
                if os.environ.get("WF_FUZZ_DUMP"):  # the whole case, for a post-mortem against the oracle on a CPU
                    ws_, wd_ = w.get_wind()
                    np.savez(os.path.join(os.environ["WF_FUZZ_DUMP"], f"bad_{seed}_{sess}_{nchecks}.npz"), x=x, y=y, ws=ws_, wd=wd_,
                             yaw=last_yaw, flags=w.risk_flags(), model=np.array(repr(model)), typed=np.array(repr(typed)),
                             **{"got_" + k: v for k, v in got.items()})

        for _ in range(n_ops):
            op = rng.choice(["step", "step", "step_torch", "wind_shared", "wind_per_farm", "wind_device", "wind_sample", "series",
                             "series_step", "batch", "model", "layout", "env_step", "env_step", "env_reset", "env_config",
                             "resolve", "kernel_choice", "layouts", "turbine_types"])
            log.append(str(op))
            N = x.size
            if op in ("layout", "batch", "layouts"):
                multi = None  # (wf_set_batch returns the handle to the first layout)
            if op in ("layout", "batch", "layouts", "wind_shared", "wind_per_farm", "wind_device", "wind_sample"):
                w._series_left = 0  # any other way of setting the wind leaves series mode
                w._ws_prev = None
            if op in ("layout", "model") and typed is not None:  # another turbine count invalidates the definitions; a model
                w.set_turbine_types(None, None)                   # without them clears them (backend.set_model)
                w.set_risk_resolve(1 if resolve_on else 0)
                typed = None
            if op == "turbine_types":  # several turbine definitions per farm on / off (wf_set_turbine_types)
                if typed is None:
                    base = ModelParams()
                    k = float(rng.uniform(0.7, 0.95))
                    d1 = dict(table_ct=[k * c for c in base.table_ct], table_cp=[0.8 * c for c in base.table_cp], TSR=float(rng.choice([7.0, 8.5])),
                              pP=float(rng.choice([1.7, 2.0])), gen_eff=0.95)
                    d2 = dict(table_ws=[0.0, 3.0, 9.0, 12.0, 25.0, 25.5], table_ct=[0.0, 0.85, 0.8, 0.45, 0.1, 0.0],
                              table_cp=[0.0, 0.25, 0.46, 0.4, 0.05, 0.0], ref_density=1.2)
                    defs = [{}, d1, d2][: int(rng.integers(2, 4))]
                    tof = rng.integers(0, len(defs), N)
                    w.set_turbine_types([{("tsr" if kk == "TSR" else kk): v for kk, v in d.items()} for d in defs], tof)
                    typed = (defs, tof)
                else:
                    w.set_turbine_types(None, None)
                    w.set_risk_resolve(1 if resolve_on else 0)
                    typed = None
            elif op == "resolve" and typed is not None:
                log[-1] = "resolve(skipped)"  # (mode 0 is refused while definitions are set)
            elif op == "resolve":  # float64 re-solve of the flagged farms on / off (wf_set_risk_resolve)
                resolve_on = not resolve_on
                w.set_risk_resolve(1 if resolve_on else 0)
            elif op == "kernel_choice":  # another kernel family for this handle; the wind has to be set again
                fam = str(rng.choice(["", "8", "4x2", "4", "2x2", "off"]))
                w.set_kernel_choice(one_block=(False if fam == "off" else (fam or None)))
                w._series_left = 0
                w._ws_prev = None
                w.set_wind(float(rng.uniform(4, 15)), float(rng.choice([270.0, rng.uniform(0, 360)])))
            elif op == "layout":
                x, y = new_layout()
                w.set_layout(x, y)
                B = new_batch()
                sub = rng.choice(B, 192, replace=False) if BIG else None
                w.set_batch(B)
                w.env_batch = B
                w.set_wind(float(rng.uniform(4, 15)), float(rng.choice([270.0, 90.0, rng.uniform(0, 360)])))
                w.env_reset()
            elif op == "layouts":  # several layouts in the batch (wf_set_layouts): shifted / jittered copies of the current one
                K = B if (rng.random() < 0.3 and B <= 64) else int(rng.integers(1, min(B, 4) + 1))
                if BIG:
                    K = int(rng.integers(1, 4))  # (the oracle runs layout by layout)
                X = np.repeat(x[None, :], K, axis=0) + rng.choice([0.0, 126.0, -378.0], (K, 1))
                Y = np.repeat(y[None, :], K, axis=0) + rng.choice([0.0, 252.0, -126.0], (K, 1))
                jit = rng.random(K) < 0.5
                X[jit] += rng.uniform(-50, 50, (int(jit.sum()), N))
                Y[jit] += rng.uniform(-50, 50, (int(jit.sum()), N))
                lof = None if K == B and rng.random() < 0.5 else rng.integers(0, K, B).astype(np.int32)
                if K == 1:
                    lof = None
                w.set_layouts(X, Y, lof)
                x, y = X[0].copy(), Y[0].copy()
                multi = None if K == 1 else (X, Y, np.arange(B) if lof is None else lof)
                w.set_wind(float(rng.uniform(4, 15)), float(rng.choice([270.0, 90.0, rng.uniform(0, 360)])))
            elif op == "batch":
                B = new_batch() if BIG else int(rng.choice([1, rng.integers(1, 40), rng.integers(200, 3000)]))
                sub = rng.choice(B, 192, replace=False) if BIG else None
                w.set_batch(B)
                w.env_batch = B
                w.set_wind(float(rng.uniform(4, 15)), float(rng.uniform(0, 360)))
                w.env_reset()
            elif op == "model":
                if rng.random() < 0.4:
                    model, mp = {}, None
                else:
                    D = 126.0
                    model = dict(hub_height=float(rng.choice([70.0, 90.0, 110.0])), ambient_ti=float(rng.choice([0.06, 0.09])),
                                 shear=float(rng.choice([0.12, 0.0])))
                    mp = ModelParams(HH=model["hub_height"], ambient_ti=model["ambient_ti"], shear=model["shear"])
                w.set_model(dict(model))
            elif op == "wind_shared":
                w.set_wind(float(rng.uniform(4, 20)), float(rng.choice([270.0, 0.0, 90.0, rng.uniform(0, 360)])))
            elif op == "wind_per_farm":
                wdv = rng.uniform(0, 360, B) if rng.random() < 0.6 else np.full(B, float(rng.choice([270.0, 90.0, rng.uniform(0, 360)])))
                w.set_wind(rng.uniform(4, 20, B), wdv)  # 40 %: one direction, a speed per farm (shared geometry + table)
            elif op == "wind_device":
                n = B if rng.random() < 0.5 else 1
                w.set_wind(torch.from_numpy(rng.uniform(4, 20, n)).cuda(), torch.from_numpy(rng.uniform(0, 360, n)).cuda())
            elif op == "wind_sample":
                w.sample_wind(int(rng.integers(0, 2**31)))
            elif op == "series":
                T = int(rng.integers(2, 12))
                series = np.stack([rng.uniform(4, 20, T), rng.uniform(0, 360, T)], axis=1)
                w.set_wind_series(series, start=rng.integers(0, T, B).astype(np.int32) if rng.random() < 0.5 else None,
                                  seed=int(rng.integers(0, 1000)))
                w._series_left = T - 1
                w._ws_prev = None
            elif op == "series_step":
                if getattr(w, "_series_left", 0) > 0:
                    w._ws_prev = w.get_wind()[0]  # the reward is normalised by the free wind BEFORE the step (a9)
                    w.wind_series_step()
                    w._series_left -= 1
                else:
                    log[-1] = "series_step(skipped)"
            elif op == "env_reset":
                w.env_reset()
            elif op == "env_config":
                envp = dict(yaw_lo=-float(rng.choice([30, 40])), yaw_hi=float(rng.choice([30, 40])), yaw_step=float(rng.choice([4, 5])),
                            actuator_rate=0.3, dt=60.0, budget=float(rng.choice([0.1, 0.02])), load_coef=float(rng.choice([0.1, 0.5])),
                            discrete=bool(rng.random() < 0.4))
                w.env_config(**envp)
                w.env_reset()
            elif op in ("step", "step_torch"):
                yaw = rng.uniform(-35, 35, (B, N)).astype(np.float32)
                got = w.step(torch.from_numpy(yaw).cuda()) if op == "step_torch" else w.step(yaw)
                ref, _ = oracle(yaw.astype(np.float64))
                last_yaw = yaw
                check(op, got, ref)
                cal = w.calibration()
                ncal += (cal["shape"] is not None) + (cal["on_the_fly"] is not None)
            elif op == "env_step":
                st = w.env_get_state()
                act = (rng.integers(0, 3, (B, N)) if envp["discrete"] else rng.uniform(-8, 8, (B, N))).astype(np.float32)
                use_torch = rng.random() < 0.5
                got = w.env_step(torch.from_numpy(act).cuda() if use_torch else act)
                got = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in got.items()}
                yaw_new = mdp_step_f32(st, act, envp)
                if not np.array_equal(got["yaw"], yaw_new):
                    nbad += 1
                    print("BAD yaw transition", dict(session=sess, seed=seed, N=N, B=B, envp=envp), np.abs(got["yaw"] - yaw_new).max(), log[-12:], flush=True)
                    where = np.argwhere(got["yaw"] != yaw_new)
                    print("   (farm, turbine) got / expected / state before / action:",
                          [(int(b), int(t), float(got["yaw"][b, t]), float(yaw_new[b, t]), float(st["yaw"][b, t]), float(act[b, t])) for b, t in where[:8]],
                          "of", len(where), w.kernel_info(), flush=True)
                    st2 = w.env_get_state()
                    b0 = int(where[0][0])
                    print("   farm", b0, "acc before", st["acc"][b0][:12], "after", st2["acc"][b0][:12], "moves", int(st["moves"][b0]), "->", int(st2["moves"][b0]),
                          "yaw after", st2["yaw"][b0][:12], flush=True)
                    if os.environ.get("FUZZ_DUMP"):
                        ws_, wd_ = w.get_wind()
                        np.savez(os.environ["FUZZ_DUMP"], x=x, y=y, ws=ws_, wd=wd_, yaw=st["yaw"], acc=st["acc"], moves=st["moves"], act=act,
                                 got_yaw=got["yaw"], got_power=got["power"], got_wd=got["wind_direction"], envp=np.array(repr(envp)), llg=np.array(llg))
                ref, ws = oracle(yaw_new.astype(np.float64))
                last_yaw = yaw_new
                check("env_step", got, ref)
                # the wind of the state before the step: the one before the last series tick, ONCE (a second env step without
                # a new tick starts from the current wind; include/wfstep.h: wf_env_set_prev_wind)
                prev = getattr(w, "_ws_prev", None)
                wsn = S(prev) if prev is not None else ws  # (ws: the sample's already)
                w._ws_prev = None
                r_ref = (ref["power"] / 1e6 * 1e3 / wsn[:, None] ** 3).mean(axis=1) - envp["load_coef"] * np.abs(ref["load"]).mean(axis=(1, 2))
                bad_r = np.abs(S(got["reward"]) - r_ref) > 5e-5 * np.abs(r_ref) + 1e-7
                if (bad_r & (S(w.risk_flags()) == 0)).any():  # a reward may only differ where the kernel flagged the farm
                    nbad += 1
                    print("BAD reward", dict(session=sess, seed=seed, N=N, B=B, envp=envp), float(np.abs(S(got["reward"]) / r_ref - 1).max()), log[-12:], flush=True)
        w.close()
    print(f"api fuzz{' (big batches: ' + str(ncal) + ' checks behind a kernel calibration)' if BIG else ''}: {n_sessions} sessions x {n_ops} ops, {nchecks} oracle checks ({ntyped} on farms of several turbine definitions): {nflip} threshold flips, {nbad} violations")
    return nflip, nbad


if __name__ == "__main__":
    a = sys.argv
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    _, bad = run(int(a[1]) if len(a) > 1 else 30, int(a[2]) if len(a) > 2 else 40, int(a[3]) if len(a) > 3 else 1)
    sys.exit(1 if bad else 0)
