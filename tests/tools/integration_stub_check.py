"""GPU box: runs the ctypes stub of INTEGRATION.md section 2 exactly as printed there (extracted from the markdown) against the
reference's known-answer vector — the binding a maintainer would add must work as documented."""
import json, os, re, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
code = re.search(r"## 2\. The ctypes stub.*?```python\n(.*?)```", md, re.S).group(1)
code = code.replace('C.CDLL("libwfstep.so")', f'C.CDLL("{os.path.join(ROOT, "wfcrl-env_amd", "libwfstep.so")}")')
ns = {}
exec(code, ns)
kat = json.load(open(os.path.join(ROOT, "tests", "golden", "kat1_demo_notebook.json")))
farm = ns["WfFarm"](kat["xcoords"], kat["ycoords"])
farm.set_wind(kat["wind_speed_free"], kat["wind_direction_free"])
p, ws, wd, load = farm.step(np.zeros(7))
print("wind_speed rel err", np.abs(ws / np.array(kat["wind_speed"]) - 1).max(), " wind_direction abs err", np.abs(wd - np.array(kat["wind_direction"])).max())
assert np.abs(ws / np.array(kat["wind_speed"]) - 1).max() < 2e-6 and np.abs(wd - np.array(kat["wind_direction"])).max() < 1e-4
assert farm.risk_flags() == 0
# ... and as printed it honours the reference's float64 contract: on a farm whose overlap count float32 cannot decide
# (tests/golden/regime_cases.npz::overlap_flip — a deficit within 1e-7 of the threshold) the step is re-solved in float64
# behind the float32 kernel (the default mode of a new handle), its result is the oracle's and no flag is left
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity
rc = np.load(os.path.join(ROOT, "tests", "golden", "regime_cases.npz"), allow_pickle=True)
f2 = ns["WfFarm"](rc["overlap_flip_x"], rc["overlap_flip_y"])
f2.set_wind(float(rc["overlap_flip_ws"]), float(rc["overlap_flip_wd"]))
p, ws, wd, load = f2.step(rc["overlap_flip_yaw"].reshape(-1))
assert f2.risk_flags() == 0
ref = {k: rc["overlap_flip_ref_" + k].reshape((1,) + np.asarray(v).shape) for k, v in (("power", p), ("wind_speed", ws), ("wind_direction", wd), ("load", load))}
parity.check_strict({"power": p[None], "wind_speed": ws[None], "wind_direction": wd[None], "load": load[None]}, ref)
print("INTEGRATION.md stub: ok")
