"""GPU box: randomised parity fuzz of the HIP path against the float64 C oracle — random layouts (regular grids with
exact ties, jittered grids, random clouds with a minimum spacing), wind directions including the axis-aligned ones,
every kernel variant that can hold the farm, shared and per-farm wind, plain step and fused env step outputs.
usage: python tests/tools/fuzz_parity.py [n_cases] [seed]      (exit code 1 on the first violation)"""
import os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
sys.path.insert(0, os.getcwd())
import numpy as np

# free-stream speeds; FUZZ_WS=lo,hi widens them past cut-in / cut-out and the ends of the Ct / power table
WS_RANGE = tuple(float(v) for v in os.environ.get("FUZZ_WS", "4,20").split(","))
VARIANTS = [(g, s) for g, smax in ((4, 4), (8, 4), (16, 6), (32, 4), (64, 4)) for s in range(1, smax + 1)]


def make_layout(rng):
    kind = rng.integers(0, 4)
    if os.environ.get("WF_FUZZ_SKIP"):  # dense farms (2-3 D spacing): 6.12 sigma_y reaches the neighbours of a wake
        nc, nr = rng.integers(2, 14), rng.integers(2, 10)
        sx, sy = rng.choice([252.0, 315.0, 378.0, 504.0]), rng.choice([252.0, 290.0, 378.0])
        x = np.repeat(np.arange(nc) * sx, nr) + np.tile(np.arange(nr) * rng.choice([0.0, 31.0, 90.0]), nc)
        y = np.tile(np.arange(nr) * sy, nc)
        if rng.random() < 0.5:
            x = x + rng.uniform(-40, 40, x.size)
            y = y + rng.uniform(-40, 40, y.size)
        return x, y
    if rng.random() < float(os.environ.get("FUZZ_BIG", "0.03")):  # up to the 256-turbine limit of the ABI
        nc, nr = rng.integers(8, 17), rng.integers(8, 17)
        x = np.repeat(np.arange(nc) * 630.0, nr) + (rng.uniform(-50, 50, nc * nr) if rng.random() < 0.5 else 0.0)
        y = np.tile(np.arange(nr) * 504.0, nc)
        keep = np.arange(x.size) < 256
        return x[keep], y[keep]
    if kind == 0:  # regular grid, exact ties at axis-aligned directions
        nc, nr = rng.integers(1, 12), rng.integers(1, 10)
        x = np.repeat(np.arange(nc) * rng.choice([504.0, 630.0, 882.0]), nr)
        y = np.tile(np.arange(nr) * rng.choice([378.0, 504.0, 756.0]), nc)
    elif kind == 1:  # jittered grid
        nc, nr = rng.integers(1, 12), rng.integers(1, 10)
        x = np.repeat(np.arange(nc) * 700.0, nr) + rng.uniform(-80, 80, nc * nr)
        y = np.tile(np.arange(nr) * 560.0, nc) + rng.uniform(-80, 80, nc * nr)
    elif kind == 2:  # grid with holes (ties, ragged)
        nc, nr = rng.integers(2, 12), rng.integers(2, 10)
        x = np.repeat(np.arange(nc) * 630.0, nr)
        y = np.tile(np.arange(nr) * 504.0, nc)
        keep = rng.random(x.size) < 0.7
        keep[0] = True
        x, y = x[keep], y[keep]
    else:  # random cloud, >= 2.5 D apart
        n = rng.integers(1, 90)
        pts = []
        while len(pts) < n:
            p = rng.uniform(0, 6000, 2)
            if all(np.hypot(*(p - q)) >= 2.5 * 126 for q in pts):
                pts.append(p)
        x, y = np.array(pts)[:, 0], np.array(pts)[:, 1]
    return x, y


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import parity  # noqa: E402  tests/parity.py: the per-farm contract (strict on unflagged farms)


def worst(got, ref, flags):
    return parity.summarize(got, ref, flags)


def classify(s):
    """'ok': every farm inside the strict tolerances; 'flip': every mismatch sits on a farm the kernel flagged itself
    (WF_RISK_*: a deficit inside the guard band of the overlap threshold in THIS farm, confirmed by the oracle's own
    margin; or a knee of the power table) and stays inside the bounded signature of that event; 'BAD' otherwise — an
    unflagged farm outside the tolerances, a flagged one outside the bound, or a spurious flag."""
    k = parity.classify(s)
    return "flip" if k == "flagged" else k


def run(n_cases, seed, only=-1, resolve=False):
    """resolve: the same cases with the float64 re-solve on (wf_set_risk_resolve) — every farm must then be inside TOL:
    a farm outside it counts as a violation, and 'flip' cannot occur."""
    from oracle import c_oracle
    from oracle.floris_gch_numpy import ModelParams
    from wfcrl_env_amd.backend import WfStep

    rng = np.random.default_rng(seed)  # only >= 0: replay one case of the sequence, verbosely
    nbad = nflip = 0
    for case in range(n_cases):
        x, y = make_layout(rng)
        N = x.size
        fits = [v for v in VARIANTS if v[0] * v[1] >= N]
        G, S = fits[rng.integers(0, len(fits))]
        choice = dict(slot=(G, S))
        # every other case also forces the one-block-at-a-time kernel (csrc/wf_kernels_ll.hip) at a random lane-group
        # width: it serves the table-path modes of farms with more than one block, wf_step_kernel the rest
        llg = str(rng.choice(["0", "0", "4", "8", "16", "4x2", "4x2", "2x2"]))
        skip_leg = bool(os.environ.get("WF_FUZZ_SKIP"))  # the far-pair skip of the table-path one-block kernels
        if skip_leg:
            llg = str(rng.choice(["2x2", "2x2", "4x2", "4"]))
        if llg != "0" and N > eval(llg.replace("x", "*")):
            choice["one_block"] = llg
        B = int(rng.integers(1, 9))
        yaw = rng.uniform(-40, 40, (B, N)).astype(np.float32) if skip_leg else rng.uniform(-35, 35, (B, N)).astype(np.float32)
        wd0 = float(rng.choice([0.0, 90.0, 180.0, 270.0, 360.0, rng.uniform(0, 360), rng.uniform(250, 290)]))
        if skip_leg:  # mostly oblique: the axis-aligned directions tie across blocks and leave the one-block kernel
            wd0 = float(rng.choice([270.0, rng.uniform(0, 360), rng.uniform(250, 290), rng.uniform(0, 360)]))
        ws0 = float(rng.uniform(*WS_RANGE))
        # every fourth case: a non-default model (low hub = general mirror cores, other D: 15 D no longer an integer
        # number of grid steps, other ambient TI / shear / deflection offsets)
        model, mp = {}, None
        all_veer = bool(os.environ.get("WF_FUZZ_VEER"))  # every case a veer model (the VEER kernel instantiations)
        if rng.random() < 0.25 or all_veer:
            D = float(rng.choice([126.0, 100.5, 150.0]))
            model = dict(rotor_diameter=D, hub_height=float(rng.choice([0.56, 0.714, 0.9]) * D),
                         ambient_ti=float(rng.choice([0.06, 0.1])), shear=float(rng.choice([0.12, 0.0, 0.2])),
                         ad=float(rng.choice([0.0, 0.01])), bd=float(rng.choice([0.0, -0.002])),
                         veer=float(rng.choice([4.0, -6.0, 12.0] if all_veer else [0.0, 0.0, 4.0, -6.0])))
            if os.environ.get("WF_FUZZ_GCH"):  # also the GCH internals FLORIS exposes: vortex core size, recovery gain
                model.update(eps_gain=float(rng.choice([0.2, 0.1, 0.35])), gch_gain=float(rng.choice([2.0, 1.0])))
            mp = ModelParams(D=model["rotor_diameter"], HH=model["hub_height"], ambient_ti=model["ambient_ti"],
                             shear=model["shear"], ad=model["ad"], bd=model["bd"], veer=model["veer"],
                             eps_gain=model.get("eps_gain", 0.2), gch_gain=model.get("gch_gain", 2.0))
            x, y = x * (D / 126.0), y * (D / 126.0)  # keeps the grids on the thresholds
        if skip_leg:
            # models that move what the skip's bound rests on: wake growth ka / kb x {0.25, 1, 4}, a deflection set of its
            # own (sigma ratios < 1: the log argument of the far-wake deflection below 1), deflection offsets ad / bd,
            # rotor sizes 60-220 m, a user thrust table reaching Ct 0.9999
            D = float(rng.choice([60.0, 90.0, 126.0, 170.0, 220.0]))
            kf, df = float(rng.choice([0.25, 1.0, 4.0])), float(rng.choice([0.25, 1.0, 4.0]))
            model = dict(rotor_diameter=D, hub_height=float(rng.choice([0.6, 0.714, 0.9]) * D),
                         ambient_ti=float(rng.choice([0.04, 0.06, 0.12])), ka=0.38 * kf, kb=0.004 * kf,
                         defl_ka=0.38 * df, defl_kb=0.004 * df, defl_alpha=float(rng.choice([0.58, 0.3, 1.2])),
                         defl_beta=float(rng.choice([0.077, 0.03, 0.2])), alpha=float(rng.choice([0.58, 0.4, 0.9])),
                         beta=float(rng.choice([0.077, 0.05, 0.15])), ad=float(rng.choice([0.0, 0.02, -0.05])) * D / 126.0,
                         bd=float(rng.choice([0.0, -0.01, 0.006])), dm=float(rng.choice([1.0, 1.3])))
            mp = ModelParams(D=D, HH=model["hub_height"], ambient_ti=model["ambient_ti"], ka=model["ka"], kb=model["kb"],
                             defl_ka=model["defl_ka"], defl_kb=model["defl_kb"], defl_alpha=model["defl_alpha"],
                             defl_beta=model["defl_beta"], alpha=model["alpha"], beta=model["beta"], ad=model["ad"],
                             bd=model["bd"], dm=model["dm"])
            if rng.random() < 0.3:  # thrust table up to the clip
                tct = np.clip(np.asarray(mp.table_ct) * 1.35, 0.0, 0.9999)
                model["table_ws"], model["table_ct"], model["table_cp"] = list(mp.table_ws), list(tct), list(mp.table_cp)
                mp.table_ct = list(tct)
            x, y = x * (D / 126.0), y * (D / 126.0)
        run = only < 0 or case == only
        if run:
            w = WfStep(x, y, env_batch=B, model=dict(model) if model else None, kernel_choice=choice)
            if resolve:
                w.set_risk_resolve(1)
            info = w.kernel_info()
            assert (info["lanes_per_env"], info["slots_per_lane"]) == (G, S)
        for mode in (("shared", "shared_dir", "shared_dir") if skip_leg else ("shared", "per_farm", "shared_dir")):
            if mode == "shared":
                ws, wd = np.array([ws0]), np.array([wd0])
            elif mode == "shared_dir":  # a speed per farm under one direction: table path with per-farm speeds
                ws, wd = rng.uniform(*WS_RANGE, B), np.full(B, wd0)
            else:
                ws = rng.uniform(*WS_RANGE, B)
                wd = np.where(rng.random(B) < 0.5, wd0, rng.uniform(0, 360, B))
            if not run:
                continue
            w.set_wind(ws0 if mode == "shared" else ws, wd0 if mode == "shared" else wd)
            got = w.step(yaw)
            flags = w.risk_flags()
            ref = c_oracle.farm_step_batch(x, y, ws, wd, yaw.astype(np.float64), mp, margin=True)
            r = worst(got, ref, flags)
            k = classify(r)
            if resolve:  # flags are cleared by the re-solve: 'ok' means every farm strict; anything else is a violation
                e = parity.errors(got, ref)
                if not parity.within(e, parity.TOL, N).all() or flags.any():
                    k = "BAD"
            nflip += k == "flip"
            if k != "ok":
                nbad += k == "BAD"
                print(k, dict(case=case, N=N, G=G, S=S, LL=w.kernel_info()["one_block_kernel"] and llg, B=B, mode=mode, wd0=wd0, ws0=ws0,
                              model={a: (b if not isinstance(b, list) else f"<{len(b)} values, max {max(b):.4f}>") for a, b in model.items()},
                              table=w.kernel_info()["pair_table"]), r, flush=True)
            if only >= 0 and os.environ.get("WF_FUZZ_DUMP"):  # the whole case, for a post-mortem (tools/replay_fuzz_case.py)
                np.savez(os.path.join(os.environ["WF_FUZZ_DUMP"], f"case_{seed}_{case}_{mode}_{int(ws[-1] * 1000)}.npz"), x=x, y=y, ws=ws, wd=wd, yaw=yaw,
                         model=np.array(repr(model)), choice=np.array(repr(choice)), flags=flags, **{"got_" + k_: np.asarray(v_) for k_, v_ in got.items()})
            if only >= 0:
                np.set_printoptions(linewidth=220, precision=5, suppress=True)
                p = np.abs(got["power"].astype(np.float64) - ref["power"]) / np.maximum(ref["power"], 1e3)
                print(mode, "x", x, "\ny", y, "\nws", ws, "wd", wd)
                b = int(np.argmax(p.max(axis=1)))
                print("flags", flags[b], "margin", ref["margin"][b], "worst farm", b, "yaw", yaw[b], "\nperr", p[b], "\ngot P", got["power"][b], "\nref P", ref["power"][b],
                      "\ngot ws", got["wind_speed"][b], "\nref ws", ref["wind_speed"][b], "\ngot TI", got["load"][b, :, 0], "\nref TI", ref["load"][b, :, 0])
        if run:
            w.close()
    print(f"fuzz{' (float64 re-solve on)' if resolve else ''}: {n_cases} cases x 3 wind modes: {nflip} threshold flips, {nbad} violations")
    return nflip, nbad


if __name__ == "__main__":
    a = sys.argv
    _, bad = run(int(a[1]) if len(a) > 1 else 200, int(a[2]) if len(a) > 2 else 1, int(a[3]) if len(a) > 3 else -1,
                 resolve=bool(os.environ.get("WF_FUZZ_RESOLVE")))
    sys.exit(1 if bad else 0)
