"""wfcrl-env_amd — MI355X-native batched wind-farm step behind the reference's FlorisInterface surface.

Hot path: HIP kernels in csrc/ behind the C ABI of include/wfstep.h, loaded with ctypes
(`backend.WfStep`).  `interface.HipFlorisInterface` mirrors reference wfcrl/interface.py:444-671.
"""
__version__ = "0.1.0"
