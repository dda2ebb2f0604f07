// wf_resolve_mt.hip — the float64 farm solve of wf_resolve.hip compiled for several turbine definitions per farm
// (include/wfstep.h: wf_set_turbine_types; FLORIS accepts a list of definitions in farm.turbine_type, reference
// wfcrl/simulators/floris/inputs/template/case.yaml:27-28): kernels wf_resolve_mt_kernel / wf_resolve4_mt_kernel, entry
// wfk_launch_resolve_mt.  See the RES_MT block at the top of wf_resolve.hip for what differs.
#define RES_MT 1
#include "wf_resolve.hip"
