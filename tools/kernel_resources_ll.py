"""Register / spill / occupancy table of the wf_step_ll_kernel instantiations (hipcc -Rpass-analysis=kernel-resource-usage,
the Makefile's flags for wf_kernels_ll.hip; no GPU needed).  usage: python tools/kernel_resources_ll.py [extra hipcc flags]"""
import re, subprocess, sys
from pathlib import Path
src = Path(__file__).resolve().parents[1] / "wfcrl-env_amd" / "csrc"
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-fast-math", "-ffp-contract=off", "-fno-slp-vectorize",
       "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-c", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null",
       str(src / "wf_kernels_ll.hip")] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        g = re.search(r"wf_step_ll_kernelILi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)ELi(\d+)ELb(\d)E", name)
        cur = {"kernel": (f"ll<{g.group(1)}x{g.group(2)},uws={g.group(3)},tab={g.group(4)},mc1={g.group(5)},veer={g.group(7)}>" if g else name[:30])}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
keys = ["kernel", "VGPRs", "SGPRs", "VGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]
print(" | ".join(keys))
for r in rows: print(" | ".join(str(r.get(k, "")) for k in keys))
