"""Generate tests/golden/oracle_goldens.npz — ORACLE-PINNED vectors (NOT reference-pinned).

The reference cannot be imported here or anywhere (FLORIS==3.5 is absent), so beyond the notebook
known-answer vector (kat1_demo_notebook.json) the goldens are produced by the float64 NumPy oracle
(oracle/floris_gch_numpy.py) once it has passed that KAT.  They freeze the oracle's behaviour for
yaw != 0, powers and load proxies so that regressions in either oracle or in the HIP path show up.
Run from the repo root:  python tests/golden/make_goldens.py
Round 3: regenerated with the default turbine table nrel_5MW_floris3 (six-decimal Cp, 5 MW plateau; the evidence is in
oracle/floris_gch_numpy.py and DESIGN.md §2 — "recollection, not reference-held"); powers move by <= 1e-6 relative below
rated, the wake solution (thrust table unchanged) does not move at all.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.floris_gch_numpy import farm_step_batch  # noqa: E402

L = json.load(open(os.path.join(ROOT, "wfcrl-env_amd", "environments", "layouts.json")))
rng = np.random.default_rng(20260101)
out = {}
cases = [("Turb3_Row1_", 6), ("Turb6_Row2_", 6), ("Ablaincourt_", 8), ("Turb16_Row5_", 4), ("Turb_TCRWP_", 3),
         ("HornsRev1_", 2), ("HornsRev2_", 2)]
for name, B in cases:
    l = L[name]
    N = l["num_turbines"]
    ws = np.clip(8 * rng.weibull(8, B), 3, 28)
    wd = rng.normal(270, 20, B) % 360
    wd[0] = 270.0  # exact-tie case for the grid layouts
    ws[0] = 8.0
    yaw = rng.uniform(-40, 40, (B, N))
    yaw[-1] = 0.0
    # inputs are rounded to what the C ABI carries (yaw float32) so every consumer sees identical inputs
    yaw = yaw.astype(np.float32).astype(np.float64)
    r = farm_step_batch(l["xcoords"], l["ycoords"], ws, wd, yaw)
    key = name.rstrip("_")
    out[f"{key}__ws"] = ws
    out[f"{key}__wd"] = wd
    out[f"{key}__yaw"] = yaw
    for k, v in r.items():
        out[f"{key}__{k}"] = v
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_goldens.npz"), **out)
print("wrote", len(out), "arrays")
