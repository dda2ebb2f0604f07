"""GPU box: post-mortem of a fuzz case dumped by tests/tools/fuzz_parity.py (WF_FUZZ_DUMP=dir, `only` set): the same inputs
under resolve mode 0 / 1 / 2 and with the register-slot kernel, per-farm errors against the C oracle, raw flags."""
import glob, os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import parity
from oracle import c_oracle
from oracle.floris_gch_numpy import ModelParams
from wfcrl_env_amd.backend import WfStep
np.set_printoptions(linewidth=220, precision=6, suppress=True)
ren = {"rotor_diameter": "D", "hub_height": "HH", "tsr": "TSR"}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "case_*.npz"))):
    d = np.load(f)
    model, choice = eval(str(d["model"])), eval(str(d["choice"]))
    mp = ModelParams(**{ren.get(k, k): v for k, v in model.items()})
    x, y, ws, wd, yaw = d["x"], d["y"], d["ws"], d["wd"], d["yaw"]
    ref = c_oracle.farm_step_batch(x, y, ws, wd, yaw.astype(np.float64), mp, margin=True)
    print("==", os.path.basename(f), "N", x.size, "B", yaw.shape[0], "choice", choice, "ws", ws, "wd", wd[:1])
    for label, ch, mode in (("as fuzzed, mode 0", choice, 0), ("mode 1", choice, 1), ("mode 2", choice, 2), ("slot kernel, mode 0", dict(slot=choice["slot"], one_block=False), 0)):
        w = WfStep(x, y, env_batch=yaw.shape[0], model=dict(model), kernel_choice=ch)
        w.set_risk_resolve(mode)
        w.set_wind(ws if ws.size > 1 else float(ws[0]), wd if ws.size > 1 else float(wd[0]))
        got = w.step(yaw)
        e = parity.errors(got, ref)
        st = w.resolve_stats() if mode else None
        raw = w.resolve_stats().get("raw_flags") if mode else None
        print(f"  {label:22s} flags {w.risk_flags()}  resolved {None if st is None else st['n_resolved']}  per farm: power err {e['power']}  TI abs err {e['ti']}  ws err {e['ws']}")
        w.close()
    b = int(np.argmax(parity.errors({k[4:]: d[k] for k in d.files if k.startswith('got_')}, ref)["ti"]))
    print("  worst farm", b, "yaw", yaw[b], "\n  got TI", d["got_load"][b, :, 0], "\n  ref TI", ref["load"][b, :, 0], "\n  ref ws", ref["wind_speed"][b])
