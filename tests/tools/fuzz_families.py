"""GPU box: CROSS-FAMILY differential fuzz — no oracle in the loop.

Every kernel family that can hold a random farm solves the SAME case: register-slot variants wf_step_kernel<G,S>, the
one-block families wf_step_ll_kernel<G,S> (4x1, 8x1, 16x1, 4x2, 2x2) on the pair-table path and on the fly, and the float64
kernels (mode 2: the four-wave kernel with level stages and its helper waves, the same without levels on four waves, the
one-wave kernel at a batch beyond the four-wave residency).  Then they are compared WITH EACH OTHER:
  * float32 family against float32 family on the farms neither flags: 2 x TOL (each is within TOL of the truth);
  * every float32 family against the float64 kernel on the farms it does not flag: TOL;
  * the float64 kernel with and without level stages: bit for bit; the one-wave float64 kernel against the four-wave one: 2e-6.
A defect shared by ALL families (round 5's TI floor) cannot be seen this way — the oracle legs are for that; one that is not
shared is seen without the oracle's own blind spots (its restatement of the model).
usage: python tests/tools/fuzz_families.py [n_cases] [seed]     (exit code 1 on a violation)"""
import os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import parity
from fuzz_parity import make_layout, WS_RANGE

ONE_BLOCK = [("4", 4), ("8", 8), ("16", 16), ("4x2", 8), ("2x2", 4)]
SLOTS = [(g, s) for g, smax in ((4, 4), (8, 4), (16, 6), (32, 4), (64, 4)) for s in range(1, smax + 1)]


def as64(o):
    return {k: np.asarray(v, dtype=np.float64) for k, v in o.items() if k in ("power", "wind_speed", "wind_direction", "load")}


def run(n_cases, seed):
    from wfcrl_env_amd import _lib
    from wfcrl_env_amd.backend import WfStep

    lib = _lib.load()
    rng = np.random.default_rng(seed)
    nbad = npairs = nfam = 0
    for case in range(n_cases):
        x, y = make_layout(rng)
        N = x.size
        B = int(rng.integers(2, 7))
        yaw = rng.uniform(-35, 35, (B, N)).astype(np.float32)
        wd0 = float(rng.choice([0.0, 90.0, 270.0, rng.uniform(0, 360), rng.uniform(250, 290)]))
        ws0 = float(rng.uniform(*WS_RANGE))
        model = {}
        if rng.random() < 0.25:
            D = float(rng.choice([126.0, 100.5, 150.0]))
            model = dict(rotor_diameter=D, hub_height=float(rng.choice([0.56, 0.714, 0.9]) * D), ambient_ti=float(rng.choice([0.06, 0.1])),
                         shear=float(rng.choice([0.12, 0.0, 0.2])), ad=float(rng.choice([0.0, 0.01])), bd=float(rng.choice([0.0, -0.002])))
            x, y = x * (D / 126.0), y * (D / 126.0)
        mode = str(rng.choice(["shared", "shared_dir", "per_farm"]))
        if mode == "shared":
            ws, wd = ws0, wd0
        elif mode == "shared_dir":
            ws, wd = rng.uniform(*WS_RANGE, B), np.full(B, wd0)
        else:
            ws, wd = rng.uniform(*WS_RANGE, B), np.where(rng.random(B) < 0.5, wd0, rng.uniform(0, 360, B))
        fits = [v for v in SLOTS if v[0] * v[1] >= N]
        pick = [fits[i] for i in rng.choice(len(fits), size=min(2, len(fits)), replace=False)]
        fams = [(f"slot{g}x{s}", dict(slot=(g, s))) for g, s in pick]
        fams += [(f"ll{n}", dict(slot=fits[0], one_block=n)) for n, gs in ONE_BLOCK if N > gs]
        res = {}
        for name, choice in fams:
            w = WfStep(x, y, env_batch=B, model=dict(model) if model else None, kernel_choice=choice)
            w.set_wind(ws, wd)
            o = w.step(yaw)
            res[name] = (as64(o), w.risk_flags().copy())
            w.close()
        # float64: four-wave kernel with levels, without, and the one-wave kernel (a batch beyond the four-wave residency: the
        # case's farms repeated)
        f64 = {}
        for name, lv, rep in (("f64_levels", 1, 1), ("f64_sequential", 0, 1), ("f64_one_wave", 1, (1100 + B - 1) // B)):
            lib.wfk_set_resolve_levels(lv)
            lib.wfk_set_resolve_helpers(2 if name == "f64_levels" else 0)  # (levels with the helper waves at work / four waves only)
            lib.wfk_set_resolve_policy(1 if name == "f64_one_wave" else 0)  # ("both": mode 2 beyond a residency on the one-wave kernel)
            Bt = B * rep
            w = WfStep(x, y, env_batch=Bt, model=dict(model) if model else None)
            w.set_risk_resolve(2)
            w.set_wind(np.tile(np.broadcast_to(ws, (B,)), rep) if rep > 1 else ws, np.tile(np.broadcast_to(wd, (B,)), rep) if rep > 1 else wd)
            o = w.step(np.tile(yaw, (rep, 1)))
            f64[name] = {k: np.asarray(v)[:B].copy() for k, v in o.items()}
            w.close()
        lib.wfk_set_resolve_levels(1)
        lib.wfk_set_resolve_helpers(1)
        lib.wfk_set_resolve_policy(0)
        bad = []
        for k in f64["f64_levels"]:
            if not np.array_equal(f64["f64_levels"][k].view(np.uint32), f64["f64_sequential"][k].view(np.uint32)):
                bad.append(("f64 levels vs sequential: bits differ", k))
            d = np.abs(f64["f64_levels"][k].astype(np.float64) - f64["f64_one_wave"][k])
            if d.max() > 2e-6 * max(1.0, np.abs(f64["f64_levels"][k]).max()):
                bad.append(("f64 four-wave vs one-wave", k, float(d.max())))
        ref = as64(f64["f64_levels"])
        names = list(res)
        for a in names:
            oa, fa = res[a]
            e = parity.errors(oa, ref)
            ok = parity.within(e, parity.TOL, N) | (fa != 0)
            if not ok.all():
                bad.append((a, "vs float64", {k: float(v[~ok].max()) for k, v in e.items()}))
            for b in names[names.index(a) + 1:]:
                ob, fb = res[b]
                e2 = parity.errors(oa, ob)
                tol2 = {k: 2 * v for k, v in parity.TOL.items()}
                ok2 = parity.within(e2, tol2, N) | (fa != 0) | (fb != 0)
                npairs += 1
                if not ok2.all():
                    bad.append((a, b, {k: float(v[~ok2].max()) for k, v in e2.items()}))
        nfam += len(names) + 3
        if bad:
            nbad += 1
            print("BAD", dict(case=case, N=N, B=B, mode=mode, wd0=wd0, ws0=ws0, model=model), bad[:4], flush=True)
    print(f"family fuzz: {n_cases} cases, {nfam} kernel-family solves, {npairs} float32 pairs compared at 2 x TOL, every family against "
          f"the float64 kernel at TOL, float64 levels / sequential bit for bit: {nbad} violations")
    return nbad


if __name__ == "__main__":
    a = sys.argv
    sys.exit(1 if run(int(a[1]) if len(a) > 1 else 100, int(a[2]) if len(a) > 2 else 1) else 0)
