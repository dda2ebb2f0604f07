// wf_resolve4_mt.hip — part 2 of wf_resolve.hip for several turbine definitions per farm (see wf_resolve_mt.hip, wf_resolve4.hip).
#define RES_MT 1
#define RES_PART 2
#include "wf_resolve.hip"
