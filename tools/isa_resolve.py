"""Static instruction counts of the float64 re-solve's functions (wf_resolve.hip), after
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-fast-math -ffp-contract=off -fno-slp-vectorize -S --cuda-device-only -o build/asm/res.s wfcrl-env_amd/csrc/wf_resolve.hip"""
import re, sys
lines = open(sys.argv[1] if len(sys.argv) > 1 else "build/asm/res.s").read().split("\n")
cur, stats = None, {}
for l in lines:
    m = re.match(r"^(_Z\w+|wf_\w+):", l)
    if m:
        cur = m.group(1); stats[cur] = dict(n=0, scratch=0, valu=0, lds=0, calls=0)
    elif cur:
        mm = re.match(r"\s+([a-z][a-z0-9_]+)\s", l + " ")
        if mm and not mm.group(1).startswith("."):
            o = mm.group(1); st = stats[cur]; st["n"] += 1
            st["scratch"] += o.startswith("scratch_"); st["valu"] += o.startswith("v_"); st["lds"] += o.startswith("ds_"); st["calls"] += o.startswith("s_swappc")
for k, v in stats.items():
    if v["n"] > 20:
        print(f"{k[:58]:58s} {v}")
for l in lines:
    if re.search(r"\.name:|vgpr_count|vgpr_spill|private_segment_fixed", l) and "kernel" not in l or ".name:" in l:
        if any(x in l for x in ("wf_resolve", "vgpr", "private")): print(l.strip())
