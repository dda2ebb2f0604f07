// wf_resolve.h — arguments of the float64 farm solve (wf_resolve.hip), shared with the host side (wf_abi.hip).
#pragma once
#include <hip/hip_runtime.h>

// Model constants in float64 (from wf_model_params, include/wfstep.h; nothing depends on the wind speed)
struct WfResolveConsts {
  int N, n_table;
  int sw_steer, sw_yar, sw_tv;  // solver switches of case.yaml:46-50
  int mirror_core;              // some exp(-zm^2 / eps^2) >= 1e-17: the ground mirrors' core factors differ from 1.0 in float64
  double D, HH, TSR, amb, eps2, num_eps, sqrt2;
  double inv_D, inv_TSR, inv_eps2;  // reciprocals of the three above (the kernels' per-stage chains hold no division by a constant)
  double off[3];       // rotor-grid offsets -D/4, 0, +D/4 (lateral and vertical) [A.1-3]
  double shearf[3];    // (z_k / HH)^shear [A.2]
  double uinf1;        // mean shearf: Uinf = ws * uinf1
  double nu1[3];       // eddy viscosity of the vortex decay per unit wind speed: lm_k^2 |dU/dz_k| / ws [A.3-4]
  double vel_top, vel_bot;
  // vertical offsets of the three vortices (and their ground mirrors) from the three grid rows: 7 + 7 distinct values
  // (wf_device.h: class m + 3, zc = m q + num_eps, zm = 2 HH + m q + num_eps, q = D / 4)
  double zc[7], zc2[7], ezc[7];    // ezc = exp(-zc^2 / eps^2)
  double zm7[7], zm2[7], ezm7[7];
  double k_top, k_bot, k_core;  // secondary steering: mean_9(z / (2 pi r) core) on the source's own grid [A.3-2]
  double alpha, beta, ka, kb, ad, bd, dm;
  double defl_alpha, defl_beta, defl_ka, defl_kb;
  double e0c1, e0c2;   // 3 e^(1/12), 3 e^(1/3)
  double near_c;       // near_wake_c * D
  double ch_constant, ch_ai, ch_amb_pow, ch_down;  // ch_amb_pow = ambient_ti ^ ch_initial
  double gch_gain, overlap_thr;
  double cos_veer, cos2_veer, sin2_veer, sin_2veer;  // wind_veer [gauss.py rCalt]
  double rho_ref, dens_cbrt, pP3;  // outputs [A.4]
};

struct WfResolveArgs {
  const double* tab64;  // [3][WF_TABLE_PAD]: wind speed, Ct, power (1/2 A Cp eta ws^3)
  int* list;            // [B] compacted farm indices
  int* count;           // [1]
  int* seen_host;       // pinned host int, or null: the four-wave kernel leaves the list's length there for the NEXT launch's choice of width
  int* seen_dev;        // its shadow in device memory (the host copy is written when the length changes only)
  int wide_hint;        // the caller expects a short list (from seen_host): a launch with helper waves
  int* flags;           // [B] WF_RISK_* of the float32 step; cleared for every farm solved here
  const double *gx, *gy;  // sorted geometry (float64)
  const int* gidx;
  size_t geom_stride;   // N for a geometry per farm, 0 for a shared one (ignored when farm_group is set)
  const int* farm_group;  // [B] direction group of each farm (grouped launches), or null
  int shift, mod;       // geometry index of group g = (g + shift) % mod
  const double *ws, *wd;
  int wind_stride;
  const int* n_real;       // [B] turbines the farm really has (padded layouts, wf_device.h: WfGroupArgs::n_real), or null
  const float* yaw_in;     // [B][N] commanded yaw (plain step)
  const float* yaw_state;  // [B][N] env state after the transition (fused env step), or null
  float *o_power, *o_ws, *o_wd, *o_load;  // caller's outputs, each may be null
  float* reward;           // [B] or null
  const double* ws_prev;   // [B] or null
  float load_coef;
  int power_mw;            // the power output in MW (float32 watts x 1e-6f, as the float32 kernels write it: WfEnvArgs::power_mw)
  // several turbine definitions per farm (wf_set_turbine_types; read by the kernels of wf_resolve_mt.hip only, where tab64 is
  // [n_types][3][WF_TABLE_PAD])
  int n_types;
  const int* type_of;         // [N] definition of each turbine, caller's order
  const double* type_consts;  // [n_types][1 + WF_TYPE_CONSTS]: table entries, then 1 / TSR, pP / 3, (air / ref density)^(1/3), ref density
};
#define WF_TYPE_CONSTS 4
#define WF_MAX_TYPES 4  // == WF_MAX_TURBINE_TYPES (include/wfstep.h)

extern "C" hipError_t wfk_launch_resolve_mt(const WfResolveConsts* c, const WfResolveArgs* a, int B, int all, int* raw_flags, int n_cu,
                                            hipStream_t s);
