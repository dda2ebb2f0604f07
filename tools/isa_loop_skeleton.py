"""Skeleton of the loops of one kernel in an assembly listing: memory instructions, waits and branches with the number
of VALU instructions between them.  usage: isa_loop_skeleton.py file.s <mangled-name-prefix> [min_ds_reads]
(how the immediate `s_waitcnt vmcnt(0)` behind the log prefetch of wf_step_ll_kernel was found)"""
import re, sys

lines = open(sys.argv[1]).read().split("\n")
i = next(n for n, l in enumerate(lines) if l.startswith(sys.argv[2]) and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
body = lines[i:end]
lab = {m.group(1): n for n, l in enumerate(body) if (m := re.match(r"(\.LBB\d+_\d+):", l))}
loops = []
for n, l in enumerate(body):
    m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in lab and lab[m.group(1)] < n:
        loops.append((lab[m.group(1)], n))
want = int(sys.argv[3]) if len(sys.argv) > 3 else 18
for a, b in loops:
    seg = body[a:b + 1]
    nds = sum("ds_read_b128" in x for x in seg)
    if nds < want or any(a2 > a and b2 <= b and sum("ds_read_b128" in x for x in body[a2:b2 + 1]) >= want for a2, b2 in loops):
        continue  # innermost loop with that many LDS reads
    n_valu = sum(bool(re.match(r"\s+v_", x)) for x in seg)
    print(f"== loop at +{a}..+{b}: {n_valu} VALU")
    nv = 0
    for n, l in enumerate(seg):
        t = l.strip()
        if re.match(r"v_", t):
            nv += 1
        elif re.match(r"(ds_|s_waitcnt|global_|s_cbranch|s_branch|\.LBB|s_barrier|buffer_|scratch_|s_nop)", t):
            print(f"{n:5d} [{nv:3d} valu] {t[:100]}")
            nv = 0
