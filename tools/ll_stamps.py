"""GPU box: where a wave of wf_step_ll_kernel spends its cycles.  Needs a stamp build:
    tools/build_alt.sh stamp -DWF_LL_STAMP
    python tools/ll_stamps.py build/alt/lib_stamp.so
Phases (s_memtime cycles summed over the waves of one launch, divided by the wave count): replay of logged sources,
this block's own sources, the chunk barrier, outputs; the rest is the prologue (yaw staging, tables)."""
import ctypes, json, os, sys
os.environ.setdefault("WF_RISK_RESOLVE", "0")  # float32 kernels on their own unless the script switches the re-solve on (a handle's default is on)
from pathlib import Path

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from wfcrl_env_amd import _lib

_lib.LIB_PATH = Path(os.path.abspath(sys.argv[1]))
os.environ["WFSTEP_NO_AUTOBUILD"] = "1"
from wfcrl_env_amd.backend import WfStep

L = json.load(open("wfcrl-env_amd/environments/layouts.json"))
name = os.environ.get("LAYOUT", "HornsRev1_")
B = int(os.environ.get("B", 65536))
l = L[name]
N = l["num_turbines"]
w = WfStep(l["xcoords"], l["ycoords"], env_batch=B)
if os.environ.get("PERFARM"):
    g0 = torch.Generator().manual_seed(2)
    w.set_wind((torch.rand(B, generator=g0) * 8 + 6).double().numpy(), (torch.rand(B, generator=g0) * 360).double().numpy())
else:
    w.set_wind(8.0, 270.0)
g = torch.Generator(device="cuda").manual_seed(1)
yaw = (torch.rand((B, N), device="cuda", generator=g) * 80 - 40).float()
out = w.step(yaw)
w.sync()
lib = ctypes.CDLL(str(_lib.LIB_PATH))
buf = (ctypes.c_ulonglong * 16)()
lib.wfk_ll_stamps(buf, 1)
w.timing_begin()
w.step(yaw, out)
ms = w.timing_end()
lib.wfk_ll_stamps(buf, 1)
nw = max(buf[5], 1)
names = ["replay", "own sources", "chunk barrier", "outputs", "total"]
tot = buf[4] / nw
print(f"{name} B={B} {w.kernel_info()}  {ms:.3f} ms with stamps; {nw} waves, {tot:.0f} cycles per wave")
for k in range(4):
    print(f"  {names[k]:14s} {buf[k] / nw:10.0f} cycles  {buf[k] / buf[4]:.3f}")
print(f"  {'  of replay: transverse pass':28s} {buf[11] / nw:10.0f} cycles  {buf[11] / buf[4]:.3f}   deficit pass + chunk test {(buf[0] - buf[11]) / nw:10.0f}  {(buf[0] - buf[11]) / buf[4]:.3f}")
print(f"  {'prologue/rest':14s} {(buf[4] - sum(buf[:4])) / nw:10.0f} cycles  {(buf[4] - sum(buf[:4])) / buf[4]:.3f}")
print("  own-source step, cycles per source (the stamps themselves add ~10 %):")
for k, nm in zip(range(6, 11), ["A state + broadcasts", "B cbrt, Ct lookup, induction", "C transverse pass in the block", "D steering, deflection/deficit constants, log store", "E deficit / TI pass in the block"]):
    print(f"    {nm:52s} {buf[k] / nw / N:8.0f}")
