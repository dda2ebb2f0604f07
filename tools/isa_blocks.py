"""Per-basic-block instruction statistics of one step-kernel variant (after `make -C wfcrl-env_amd/csrc asm`).
usage: python tools/isa_blocks.py 16x5 [min_instrs]"""
import collections, re, sys
g, s_ = sys.argv[1].split("x")[:2]; TAB = 1 if sys.argv[1].endswith("t") else 0; s_ = s_.rstrip("t")
minn = int(sys.argv[2]) if len(sys.argv) > 2 else 120
lines = open("wfcrl-env_amd/csrc/wf_kernels.s").read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(f"_Z14wf_step_kernelILi{g}ELi{s_}ELb1ELb{TAB}E") and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(j for j in range(start, len(lines)) if "s_endpgm" in lines[j])
blocks, cur = [], ("entry", [])
for l in lines[start:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur); cur = (m.group(1), [])
    else:
        mm = re.match(r"\s+([a-z][a-z0-9_]+)\s", l + " ")
        if mm and not mm.group(1).startswith("."): cur[1].append(mm.group(1))
blocks.append(cur)
T = {"v_rcp_f32_e32", "v_exp_f32_e32", "v_log_f32_e32", "v_sqrt_f32_e32", "v_rsq_f32_e32"}
tot = 0
for name, ins in blocks:
    tot += len(ins)
    if len(ins) < minn: continue
    c = collections.Counter(ins)
    v = sum(n for k, n in c.items() if k.startswith("v_") and k not in T)
    t = sum(n for k, n in c.items() if k in T)
    mov = sum(n for k, n in c.items() if k.startswith(("v_mov", "v_accvgpr")))
    sc = sum(n for k, n in c.items() if k.startswith(("scratch_", "buffer_")))
    rl = sum(n for k, n in c.items() if k.startswith(("v_readlane", "v_writelane")))
    print(f"{name:12s} n={len(ins):4d} valu={v:4d} trans={t:3d} mov={mov:3d} scratch={sc:2d} lane={rl:2d} est_cyc={v*3.3+t*8.3:7.0f}  {c.most_common(5)}")
print("total static instructions", tot)
