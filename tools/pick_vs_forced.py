import json, os, sys, torch
sys.path.insert(0, os.getcwd())
from tools.batch_sweep_fine import time_it, L
lay = L["HornsRev1_"]
for B in (12288, 8192):
    for rep in range(2):
        for label, c in (("pick", None), ("forced 16x1", dict(one_block="16")), ("forced slot", dict(one_block=False))):
            t, fam = time_it(lay, B, c)
            print(B, rep, label, fam, f"{t:.4f}", flush=True)
