"""Basic blocks of one wf_step_ll_kernel instantiation in an assembly listing (hipcc -S --cuda-device-only):
usage: python tools/isa_ll_blocks.py build/asm/ll.s 16 1 [UWS=1] [TAB=1] [VEER=0]  -> label, instruction counts, successors"""
import collections, re, sys
f, g, s = sys.argv[1], sys.argv[2], sys.argv[3]
uws = sys.argv[4] if len(sys.argv) > 4 else "1"
tab = sys.argv[5] if len(sys.argv) > 5 else "1"
veer = sys.argv[6] if len(sys.argv) > 6 else "0"
lines = open(f).read().split("\n")
pre = f"_Z17wf_step_ll_kernelILi{g}ELi{s}ELb{uws}ELb{tab}ELb1ELi4ELb{veer}EE"
start = next(i for i, l in enumerate(lines) if l.startswith(pre) and l.split(";")[0].strip().endswith(":"))
end = next(j for j in range(start, len(lines)) if lines[j].startswith(".Lfunc_end"))
T = ("v_rcp_f32", "v_exp_f32", "v_log_f32", "v_sqrt_f32", "v_rsq_f32", "v_sin_f32", "v_cos_f32")
blocks, cur = [], ["entry", [], start]
for n in range(start + 1, end):
    l = lines[n]
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur); cur = [m.group(1), [], n]
    else:
        mm = re.match(r"\s+([a-z][a-z0-9_]+)(\s|$)", l)
        if mm and not mm.group(1).startswith("."): cur[1].append((mm.group(1), l.strip()))
blocks.append(cur)
tot = 0
for name, ins, ln in blocks:
    ops = [o for o, _ in ins]
    tot += len(ops)
    v = sum(o.startswith("v_") and not o.startswith(T) for o in ops)
    t = sum(o.startswith(T) for o in ops)
    ds = sum(o.startswith("ds_") for o in ops)
    gm = sum(o.startswith(("global_", "buffer_", "scratch_", "flat_")) for o in ops)
    sa = sum(o.startswith("s_") for o in ops)
    br = [x.split()[-1] for o, x in ins if o.startswith(("s_cbranch", "s_branch"))]
    w = sum(o == "s_waitcnt" for o in ops)
    print(f"{ln - start:6d} {name:12s} n={len(ops):4d} valu={v:4d} trans={t:3d} ds={ds:3d} mem={gm:2d} salu={sa:3d} wait={w:2d} -> {' '.join(br)}")
print("total", tot)
