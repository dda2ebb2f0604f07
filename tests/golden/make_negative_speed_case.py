"""How tests/golden/negative_rotor_speed_case.npz was made (inputs only; expected values come from the oracle at test time).

On the GPU box, the failing case of the round-5 fuzz campaign (tools/round_close.sh at FUZZ_SCALE=5, leg WF_FUZZ_SKIP=1 seed
5041, case 1720) was replayed with its inputs dumped:
    WF_FUZZ_DUMP=gpurun_out/c1720 WF_FUZZ_SKIP=1 python tests/tools/fuzz_parity.py 1721 5041 1720
and the first wind mode's dump reduced to its inputs here."""
import sys

import numpy as np

d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/c1720/case_5041_1720_shared_13447.npz")
np.savez_compressed("tests/golden/negative_rotor_speed_case.npz", x=d["x"], y=d["y"], ws=d["ws"], wd=d["wd"], yaw=d["yaw"], model=d["model"])
