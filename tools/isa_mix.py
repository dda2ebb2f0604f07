"""Static instruction mix per kernel variant from wfcrl-env_amd/csrc/wf_kernels.s (make -C csrc asm)."""
import collections, re, sys
want = set(sys.argv[1:]) or {"16x5"}
lines = open("wfcrl-env_amd/csrc/wf_kernels.s").read().split("\n")
starts = [(i, re.match(r"_Z14wf_step_kernelILi(\d+)ELi(\d+)E", l)) for i, l in enumerate(lines)]
starts = [(i, m) for i, m in starts if m and l_is_label(lines[i])] if False else [(i, m) for i, m in starts if m and lines[i].rstrip().split(";")[0].strip().endswith(":")]
for n, (i, m) in enumerate(starts):
    key = f"{m.group(1)}x{m.group(2)}"
    if key not in want: continue
    end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
    body = lines[i:end]
    c = collections.Counter()
    for l in body:
        mm = re.match(r"\s+([a-z][a-z0-9_]+)\s", l)
        if mm and not mm.group(1).startswith("."): c[mm.group(1)] += 1
    g = collections.Counter()
    for k, v in c.items():
        if k.startswith(("v_readlane", "v_writelane")): g["v_read/writelane"] += v
        elif k in ("v_rcp_f32", "v_exp_f32", "v_log_f32", "v_sqrt_f32", "v_rsq_f32", "v_sin_f32", "v_cos_f32"): g["transcendental"] += v
        elif k.startswith("v_pk_"): g["v_pk"] += v
        elif k.startswith(("v_cndmask", "v_cmp")): g["cmp/cndmask"] += v
        elif k.startswith(("v_mov", "v_accvgpr")): g["v_mov/accvgpr"] += v
        elif k.startswith("v_"): g["valu other"] += v
        elif k.startswith("s_"): g["salu/smem/branch"] += v
        elif k.startswith("ds_"): g["lds"] += v
        elif k.startswith(("scratch_", "buffer_")): g["scratch"] += v
        else: g["vmem/other"] += v
    print(f"== step<{key}> static instructions: {sum(c.values())}")
    print(dict(g))
    print(c.most_common(22))
