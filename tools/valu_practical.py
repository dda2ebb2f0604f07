"""The PRACTICAL fp32 VALU ceiling bench.py prints beside the nominal one -> profiles/valu_practical.json.

Nominal peak: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 7.86e13 lane-ops/s = one wave-instruction per SIMD every 2 cycles.  That is
not reachable with two waves per SIMD (what a 248-VGPR kernel gets): tools/ubench/valu_rates.hip measured on this chip
(profiles/archive/r01_valu_issue_rates_ubench.txt) 3.30 cycles per plain fp32 wave-instruction at two waves per SIMD and
8.3-8.6 per transcendental (v_exp / v_log / v_rcp / v_rsq / v_sqrt).  The ceiling for a kernel is the mix of the two by the
kernel's OWN share of transcendentals among its VALU instructions — counted statically here from the shipped code object
(llvm-objdump of libwfstep.so; the replay loops that dominate the run hold the same mix within a point).
usage: python tools/valu_practical.py   (build container or GPU box: needs only the built .so)"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "wfcrl-env_amd", "libwfstep.so")
UB = os.path.join(ROOT, "profiles", "archive", "r01_valu_issue_rates_ubench.txt")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32")


def rates():
    r = {}
    for line in open(UB):
        m = re.match(r"(.+?)\s+waves/SIMD=(\d)\s+[\d.]+ ms\s+([\d.]+) cyc", line)
        if m:
            r[(m.group(1).strip(), int(m.group(2)))] = float(m.group(3))
    return r


def kernels():
    # the device code objects are bundled in the .so: llvm-objdump --offloading writes them next to its input — a copy in a
    # scratch directory — then each gfx950 object is disassembled and split by symbol
    import glob, shutil, tempfile
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(SO, so)
        subprocess.run([OBJDUMP, "--offloading", so], capture_output=True, text=True, cwd=tmp)
        for co in sorted(glob.glob(so + ".*gfx950")):
            out = subprocess.run([OBJDUMP, "-d", co], capture_output=True, text=True).stdout
            cur = None
            for line in out.splitlines():
                m = re.match(r"[0-9a-f]+ <(.+)>:", line)
                if m:
                    cur = m.group(1); res[cur] = {"valu": 0, "trans": 0, "total": 0}
                    continue
                m = re.match(r"\s+([a-z][a-z0-9_]+)\s", line)
                if cur and m:
                    op = m.group(1)
                    res[cur]["total"] += 1
                    if op.startswith("v_") and not op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
                        res[cur]["valu"] += 1
                        if op.startswith(TRANS):
                            res[cur]["trans"] += 1
    return res


def main():
    r = rates()
    out = {"_comment": __doc__.split("usage")[0].strip(), "source_rates": os.path.relpath(UB, ROOT), "nominal_cycles_per_wave_instr": 2.0, "kernels": {}}
    for waves in (2, 3, 4):
        w = waves if (("v_fma_f32 x8 indep", waves) in r) else (4 if waves == 3 else waves)
        plain = r[("v_fma_f32 x8 indep", w)]
        trans = sum(r[(k, w)] for k in ("v_exp_f32", "v_rcp_f32", "v_sqrt_f32")) / 3.0
        out[f"cycles_per_wave_instr_at_{waves}_waves_per_simd"] = {"plain_fp32": plain, "transcendental": trans,
                                                                   "measured_at_waves_per_simd": w}
    ks = kernels()
    pat = re.compile(r"_Z17wf_step_ll_kernelILi(\d+)ELi(\d+)ELb([01])ELb([01])ELb([01])ELi4ELb([01])ELb([01])EE|_Z14wf_step_kernelILi(\d+)ELi(\d+)E")
    for name, c in ks.items():
        m = pat.match(name)
        if not m or c["valu"] < 500:
            continue
        if m.group(1):
            key = f"ll_{m.group(1)}x{m.group(2)}_shared{m.group(3)}_tab{m.group(4)}_mc{m.group(5)}_veer{m.group(6)}_occ2{m.group(7)}"
        else:
            key = f"slot_{m.group(8)}x{m.group(9)}_{name[-24:]}"
        out["kernels"][key] = {"symbol": name, "valu_static": c["valu"], "transcendental_static": c["trans"],
                               "transcendental_share": c["trans"] / c["valu"]}
    json.dump(out, open(os.path.join(ROOT, "profiles", "valu_practical.json"), "w"), indent=1)
    hk = out["kernels"].get("ll_2x2_shared1_tab1_mc1_veer0_occ20")
    print("headline kernel:", hk)
    print(len(out["kernels"]), "kernels")


if __name__ == "__main__":
    main()
