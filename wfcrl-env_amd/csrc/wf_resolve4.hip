// wf_resolve4.hip — part 2 of wf_resolve.hip: the four-wave float64 kernel with its level stages, compiled without machine
// LICM (Makefile: SETRES; see RES_PART at the top of wf_resolve.hip).
#define RES_PART 2
#include "wf_resolve.hip"
