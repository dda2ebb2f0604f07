#!/bin/bash
# GPU box, LAST thing of a round (two rounds running the final kernel commits went out unfuzzed):
# a short fuzz campaign on exactly the sources of HEAD — the legs of tools/fuzz_campaign.sh (one seed each), with one API-fuzz seed
# with batches beyond the latency regime, so that the handle's own kernel calibration (table path, on the fly, inside fused
# env steps) fires in the middle of the sessions — and a record of WHAT was fuzzed: gpurun_out/fuzz_head.json =
# {sha256 over csrc/*.hip, csrc/*.h, include/wfstep.h, totals per leg}.  Copy it to profiles/fuzz_head.json and commit:
# tests/test_abi.py::test_last_fuzz_campaign_ran_on_these_kernels recomputes the hash and fails when a kernel source changed
# after the recorded campaign.      gpurun --timeout 2400 -- 'bash tools/round_close.sh'
cd $GRAFT_REPO_ROOT
O=gpurun_out/fuzz_head.txt
: > $O
leg() { echo "## $*" >> $O; timeout ${LEG_TIMEOUT:-2400} env "$@" 2>&1 | grep -v amdgpu.ids | grep -E "^BAD|^fuzz|^api|^env|^family" | cut -c1-1500 >> $O; }
S=${FUZZ_SCALE:-5}  # (round 6: the default is the scale that found round 5's only real defect; FUZZ_SCALE=1 for an interim record)
leg python tests/tools/fuzz_parity.py $((500 * S)) 5011
leg WF_FUZZ_RESOLVE=1 python tests/tools/fuzz_parity.py $((400 * S)) 5021
leg WF_FUZZ_VEER=1 python tests/tools/fuzz_parity.py $((200 * S)) 5031
leg WF_FUZZ_SKIP=1 python tests/tools/fuzz_parity.py $((400 * S)) 5041
leg WF_FUZZ_SKIP=1 FUZZ_WS=2.5,26 WF_FUZZ_RESOLVE=1 python tests/tools/fuzz_parity.py $((300 * S)) 5043
leg python tests/tools/fuzz_api.py $((30 * S)) 50 5051
leg FUZZ_API_BIG=1 python tests/tools/fuzz_api.py $((4 * S)) 40 5053
leg python tests/tools/fuzz_env.py $((30 * S)) 5061
leg python tests/tools/fuzz_families.py $((150 * S)) 5071
python - <<'PY'
import glob, hashlib, json, re, subprocess
h = hashlib.sha256()
files = sorted(glob.glob("wfcrl-env_amd/csrc/*.hip") + glob.glob("wfcrl-env_amd/csrc/*.h")) + ["include/wfstep.h"]
for f in files:
    h.update(f.encode()); h.update(open(f, "rb").read())
legs, cur = [], None
for line in open("gpurun_out/fuzz_head.txt"):
    if line.startswith("## "):
        cur = {"leg": line[3:].strip(), "summary": None, "violations": None, "bad_lines": 0}; legs.append(cur)
    elif line.startswith("BAD"):
        cur["bad_lines"] += 1
    else:
        m = re.search(r"(\d+) violations", line) or re.search(r"violations: (\d+)", line)
        if m:
            cur["summary"], cur["violations"] = line.strip(), int(m.group(1))
import os
json.dump({"sources_sha256": h.hexdigest(), "files": files, "fuzz_scale": int(os.environ.get("FUZZ_SCALE", "5")), "legs": legs,
           "violations_total": sum((l["violations"] if l["violations"] is not None else 10 ** 6) for l in legs),
           "legs_without_a_summary": [l["leg"] for l in legs if l["summary"] is None]},
          open("gpurun_out/fuzz_head.json", "w"), indent=1)
PY
grep -E "^##|violations" $O | cut -c1-220
