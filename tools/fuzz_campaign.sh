#!/bin/bash
# GPU box: THE fuzz campaign (one script since round 6; rounds 3-5 had seven) — every leg of every earlier campaign, sized by SCALE:
#   tools/fuzz_campaign.sh [SCALE=1] [SEED_BASE=1000] [OUTFILE=fuzz_campaign.txt]     (output under gpurun_out/)
# legs: plain x2, re-solve on, veer, GCH internals, the far-pair-skip leg (dense layouts, randomised wake models, the table-path
# one-block families forced) x2, the same with wide wind speeds, the same with the re-solve, API sessions x2, env episodes, and the
# cross-family differential leg (no oracle).  SCALE=1 is rounds 3-5's "short" campaign (~4 min), 3 their full one, 5 what found
# round 5's negative-rotor-speed defect.  tools/round_close.sh is the recorded variant (hash of the sources + JSON summary).
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
S=${1:-1}; B=${2:-1000}; O=gpurun_out/${3:-fuzz_campaign.txt}
mkdir -p gpurun_out; : > $O
leg() { echo "## $*" >> $O; timeout ${LEG_TIMEOUT:-2400} env "$@" 2>&1 | grep -v amdgpu.ids | grep -E "^BAD|^fuzz|^api|^env|^family" | cut -c1-1200 >> $O; }
for k in 1 2; do leg python tests/tools/fuzz_parity.py $((500 * S)) $((B + 10 + k)); done
leg WF_FUZZ_RESOLVE=1 python tests/tools/fuzz_parity.py $((500 * S)) $((B + 21))
leg WF_FUZZ_VEER=1 python tests/tools/fuzz_parity.py $((270 * S)) $((B + 31))
leg WF_FUZZ_GCH=1 python tests/tools/fuzz_parity.py $((270 * S)) $((B + 36))
for k in 1 2; do leg WF_FUZZ_SKIP=1 python tests/tools/fuzz_parity.py $((500 * S)) $((B + 40 + k)); done
leg WF_FUZZ_SKIP=1 FUZZ_WS=2.5,26 python tests/tools/fuzz_parity.py $((330 * S)) $((B + 43))
leg WF_FUZZ_SKIP=1 WF_FUZZ_RESOLVE=1 python tests/tools/fuzz_parity.py $((330 * S)) $((B + 44))
for k in 1 2; do leg python tests/tools/fuzz_api.py $((20 * S)) 50 $((B + 50 + k)); done
leg FUZZ_API_BIG=1 python tests/tools/fuzz_api.py $((4 * S)) 40 $((B + 53))
leg python tests/tools/fuzz_env.py $((27 * S)) $((B + 61))
leg python tests/tools/fuzz_families.py $((150 * S)) $((B + 71))
grep -E "^##|violations|BAD" $O | cut -c1-300
