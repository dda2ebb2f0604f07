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
print("INTEGRATION.md stub: ok")
