"""Writes tests/golden/regime_cases.npz: single farms found by the fuzzers (tests/tools/fuzz_api.py with WF_FUZZ_DUMP set)
that pin down a regime of the model, with the C oracle's float64 outputs.

  thrust_ramp   56 turbines at a free wind of 5.44 m/s: 48 waked turbines sit at 2.8-3.5 m/s, on the cut-in ramp of the
                nrel_5MW thrust table (Ct 0 -> 0.99 between 2.5 and 3 m/s).  fuzz_api seed 512, session 56, farm 1107.
  overlap_flip  51 turbines whose float64 margin to the overlap threshold "deficit * Uinit > 0.05" is 3.7e-7: a float32
                evaluation may count one grid point the other way.  fuzz_api seed 513, session 62, farm 1626.

usage: python tests/golden/make_regime_cases.py <dump_512_56.npz> <dump_513_62.npz>
       python tests/golden/make_regime_cases.py            (re-evaluate the stored inputs: after an oracle / table change)
Inputs are stored next to the outputs, so the fixture is self-contained.
Round 3: re-evaluated with the default turbine table nrel_5MW_floris3 (oracle/floris_gch_numpy.py).
"""
import os, sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import c_oracle  # noqa: E402

out = {}
HERE = os.path.join(os.path.dirname(__file__), "regime_cases.npz")
if len(sys.argv) < 3:
    d = np.load(HERE)
    for name in ("thrust_ramp", "overlap_flip"):
        inp = {k: d[f"{name}_{k}"] for k in ("x", "y", "ws", "wd", "yaw")}
        ref = c_oracle.farm_step_batch(inp["x"], inp["y"], inp["ws"], inp["wd"], inp["yaw"], None, margin=True)
        out.update({f"{name}_{k}": v for k, v in inp.items()})
        out.update({f"{name}_ref_{k}": v for k, v in ref.items()})
    np.savez_compressed(HERE, **out)
    print({k: v.shape for k, v in out.items()})
    sys.exit(0)
for name, path, b in (("thrust_ramp", sys.argv[1], 1107), ("overlap_flip", sys.argv[2], 1626)):
    d = np.load(path)
    assert str(d["model"]) == "{}"  # default model
    x, y = d["x"], d["y"]
    ws, wd, yaw = d["ws"][b:b + 1], d["wd"][b:b + 1], d["yaw"][b:b + 1].astype(np.float64)
    ref = c_oracle.farm_step_batch(x, y, ws, wd, yaw, None, margin=True)
    out.update({f"{name}_x": x, f"{name}_y": y, f"{name}_ws": ws, f"{name}_wd": wd, f"{name}_yaw": yaw})
    out.update({f"{name}_ref_{k}": v for k, v in ref.items()})
np.savez_compressed(os.path.join(os.path.dirname(__file__), "regime_cases.npz"), **out)
print({k: v.shape for k, v in out.items()})
