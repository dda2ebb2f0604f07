"""GPU box: a wind per farm (reset distribution) — what the handle picks by itself against every forced family, over the
env batch.   python tools/fly_pick_sweep.py [layout] > gpurun_out/r04_fly_pick_sweep.txt"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from fly_family_ab import time_it, L

for name in (sys.argv[1:] or ["HornsRev1_", "Ormonde_"]):
    lay = L[name]
    print(f"# {name} N={lay['num_turbines']}: ms per step, a wind per farm")
    for B in (1024, 2048, 4096, 8192, 16384, 24576, 32768, 49152, 65536):
        row = {}
        for fam in (None, "2x2", "4x2", "8", "16", False):
            try:
                t, k = time_it(lay, B, fam)
            except Exception as e:  # a family the farm is too small for
                continue
            row["pick" if fam is None else ("slot" if fam is False else fam)] = (t, k)
        best = min(v[0] for k, v in row.items() if k != "pick")
        print(f"B={B:6d} " + "  ".join(f"{k}->{v[1]} {v[0]:.3f}" for k, v in row.items()) + f"   pick/best {best / row['pick'][0]:.3f}", flush=True)
