// wf_abi.hip — host side of libwfstep.so: the C ABI declared in include/wfstep.h.
//
// Replaces the FLORIS object the reference holds in FlorisInterface (reference
// wfcrl/interface.py:479 `tools.FlorisInterface(simul_file)`) by a handle that owns device-resident
// geometry, model constants and staging buffers.  No CPU fallback: without a HIP device wf_create
// fails with WF_E_NODEVICE.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/wfstep.h"
#include "wf_device.h"
#include "wf_resolve.h"

extern "C" hipError_t wfk_launch_resolve(const WfResolveConsts* c, const WfResolveArgs* a, int B, int all, int* raw_flags,
                                         hipStream_t s);
extern "C" int wfk_num_variants();
extern "C" void wfk_variant(int i, int* G, int* S, const void** fn);
extern "C" int wfk_variant_has_table(int i);
extern "C" int wfk_tab_waves();
extern "C" const void* wfk_variant_fn(int i, int kind);
extern "C" hipError_t wfk_launch_geometry(int n_env, int N, const double* lx, const double* ly, double xc, double yc,
                                          const double* wd, double* gx, double* gy, int* gidx, int tie_block, int* farm_tie,
                                          int* any_tie, hipStream_t s);
extern "C" int wfk_ll_has_fly(int G, int S);
extern "C" hipError_t wfk_launch_step_ll_fly(int G, int S, const WfConsts* c, const WfTables* tab, const int* gidx, const double* gx,
                                             const double* gy, const double* ws, const double* wd, const float* yaw,
                                             float* power, float* o_ws, float* o_wd, float* load, int B, const WfEnvArgs* env,
                                             const int* farm_tie, float* src_log, size_t log_side_offset,
                                             const WfGroupArgs* grp, hipStream_t s);
extern "C" hipError_t wfk_launch_step(int variant, const WfConsts* c, const WfTables* tab, const double* gx,
                                      const double* gy, const int* gidx, int geom_stride, const double* ws,
                                      const double* wd, int wind_stride, const float* yaw, float* power, float* o_ws,
                                      float* o_wd, float* load, int B, const WfEnvArgs* env, const float* pair_tab,
                                      const int* pair_first, const WfGroupArgs* grp, hipStream_t s, int* grid_out);
extern "C" hipError_t wfk_launch_pair_table(const WfPairConsts* pc, int n_groups, const double* gx, const double* gy,
                                            float* tab, int* first_active, hipStream_t s);

extern "C" hipError_t wfk_launch_wind_sample(int B, unsigned long long seed, const double* dist, double* ws, double* wd,
                                             hipStream_t s);
extern "C" size_t wfk_ll_table_floats(int N, int G);
extern "C" int wfk_ll_farms_per_block(int G);
extern "C" hipError_t wfk_launch_pair_table_ll(const WfPairConsts* pc, int G, int n_groups, const double* gx, const double* gy,
                                               float* tab, int* cross_tie, hipStream_t s);
extern "C" hipError_t wfk_launch_step_ll(int G, int S, const WfConsts* c, const WfTables* tab, const int* gidx, const double* ws,
                                         const double* wd, int wind_stride, const float* yaw, float* power, float* o_ws,
                                         float* o_wd, float* load, int B, const WfEnvArgs* env, const float* ll_tab,
                                         const int* cross_tie, float* src_log, size_t log_side_offset,
                                         const WfGroupArgs* grp, hipStream_t s);
extern "C" hipError_t wfk_ll_func_attributes(int G, int S, int shared_speed, int table, hipFuncAttributes* a);
extern "C" hipError_t wfk_launch_fill(int n, double* a, hipStream_t s);  // a[1..n) = a[0]
extern "C" hipError_t wfk_launch_wind_sample_binned(int B, unsigned long long seed, const double* dist, double step, double* ws,
                                                    double* wd, int* bin, hipStream_t s);
extern "C" hipError_t wfk_launch_bin_centres(int K, double step, double* wd, hipStream_t s);
extern "C" hipError_t wfk_launch_series_start(int B, int T, unsigned long long seed, int* start, hipStream_t s);
extern "C" hipError_t wfk_launch_series_gather(int B, int T, int t, const int* start, const double* s_ws,
                                               const double* s_wd, double* ws, double* wd, hipStream_t s);

namespace {

// ---- nrel_5MW power/thrust table (DATA, replaceable via wf_set_model) ----
const double kCtFrom3[45] = {
    0.99,       0.99,       0.97373036, 0.92826162, 0.89210543, 0.86100905, 0.835423,   0.81237673, 0.79225789,
    0.77584769, 0.7629228,  0.76156073, 0.76261984, 0.76169723, 0.75232027, 0.74026851, 0.72987175, 0.70701647,
    0.54054532, 0.45509459, 0.39343381, 0.34250785, 0.30487242, 0.27164979, 0.24361964, 0.21973831, 0.19918151,
    0.18131868, 0.16537679, 0.15103727, 0.13998636, 0.1289037,  0.11970413, 0.11087113, 0.10339901, 0.09617888,
    0.09009926, 0.08395078, 0.0791188,  0.07448356, 0.07050731, 0.06684119, 0.06345518, 0.06032267, 0.05741999};
const double kCpFrom3SurveyA5[45] = {  // "nrel_5MW_survey_a5": the 8-decimal column (FLORIS v2 example input; 4.969 MW at 12 m/s, 5.116 MW at 25 m/s)
    0.1780851,  0.28907459, 0.34902166, 0.3847278,  0.40605878, 0.4202279,  0.42882274, 0.43387274, 0.43622267,
    0.43684468, 0.43657497, 0.43651053, 0.4365612,  0.43651728, 0.43590309, 0.43467276, 0.43322955, 0.43003137,
    0.37655587, 0.33328466, 0.29700574, 0.26420779, 0.23839379, 0.21459275, 0.19382354, 0.1756635,  0.15970926,
    0.14561785, 0.13287856, 0.12130194, 0.11219941, 0.10311631, 0.09545392, 0.08813781, 0.08186763, 0.07585005,
    0.07071926, 0.06557558, 0.06148104, 0.05755207, 0.05413366, 0.05097969, 0.04806545, 0.04536883, 0.04287006};
// "nrel_5MW_floris3", the default: the six-decimal column of FLORIS 3.x' turbine_library/nrel_5MW.yaml as recollected
// (not reference-held): below rated the values above rounded, from 11.5 m/s the rated-power plateau
// Cp = 5 MW / (1/2 rho A v^3)
const double kCpFrom3Floris3[45] = {
    0.178085, 0.289075, 0.349022, 0.384728, 0.406059, 0.420228, 0.428823, 0.433873, 0.436223,
    0.436845, 0.436575, 0.436511, 0.436561, 0.436517, 0.435903, 0.434673, 0.433230, 0.430466,
    0.378869, 0.335199, 0.297991, 0.266092, 0.238588, 0.214748, 0.193981, 0.175808, 0.159835,
    0.145741, 0.133256, 0.122157, 0.112257, 0.103399, 0.095449, 0.088294, 0.081836, 0.075993,
    0.070692, 0.065875, 0.061484, 0.057476, 0.053809, 0.050447, 0.047358, 0.044518, 0.041900,
};
double g_tab_ws[51], g_tab_ct[51], g_tab_cp[51], g_tab_cp_a5[51];
std::once_flag g_tab_once;
void fill_default_table() {
  int n = 0;
  const double head[3] = {0.0, 2.0, 2.5};
  for (int i = 0; i < 3; ++i) { g_tab_ws[n] = head[i]; g_tab_ct[n] = 0.0; g_tab_cp[n] = g_tab_cp_a5[n] = 0.0; ++n; }
  for (int i = 0; i < 45; ++i) { g_tab_ws[n] = 3.0 + 0.5 * i; g_tab_ct[n] = kCtFrom3[i]; g_tab_cp[n] = kCpFrom3Floris3[i]; g_tab_cp_a5[n] = kCpFrom3SurveyA5[i]; ++n; }
  const double tail[3] = {25.01, 25.02, 50.0};
  for (int i = 0; i < 3; ++i) { g_tab_ws[n] = tail[i]; g_tab_ct[n] = 0.0; g_tab_cp[n] = g_tab_cp_a5[n] = 0.0; ++n; }
}
void init_default_table() { std::call_once(g_tab_once, fill_default_table); }  // concurrent wf_create calls

thread_local std::string g_create_error;

}  // namespace

struct wf_handle {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;

  wf_model_params model{};
  std::vector<double> tws, tct, tcp;
  bool model_dirty = true;

  int N = 0;
  std::vector<double> lx, ly;
  double xc = 0, yc = 0;
  int B = 0;
  int wind_count = 0;  // 0 = not set
  int variant = -1;
  int grid = 0;

  WfConsts consts{};
  // device memory
  WfTables* d_tab = nullptr;
  double *d_lx = nullptr, *d_ly = nullptr;
  double *d_ws = nullptr, *d_wd = nullptr;  // [B]
  double* d_gx = nullptr;                   // [B*N] (or [N] when wind is shared)
  double* d_gy = nullptr;                   // sorted y' (float64: the lateral gate is decided on it)
  int* d_gidx = nullptr;
  int* d_flags = nullptr;                   // [B] WF_RISK_* bits of the last step
  double guard_rel = 5.0e-5;                // relative half-width of the overlap-threshold guard band (a deficit at the
                                            // threshold sits in the Gaussian tail: its float32 error reaches 1-3e-5)
  float *d_yaw = nullptr, *d_out = nullptr;  // staging for host callers: yaw [B*N], out [B*N*7]
  float *h_yaw = nullptr, *h_out = nullptr;  // pinned
  size_t cap_env = 0, cap_bn = 0;
  // fused env state
  wf_env_params env{-40.f, 40.f, 5.f, 0.3f, 60.f, 0.1f, 0.1f, 0};
  float *d_env_yaw = nullptr, *d_env_acc = nullptr, *d_env_act = nullptr, *d_env_out = nullptr;  // out: reward[B] + yaw[BN]
  int* d_env_moves = nullptr;
  float *h_env_act = nullptr, *h_env_out = nullptr;
  // wind series
  int series_T = 0, series_t = 0;
  double *d_series_ws = nullptr, *d_series_wd = nullptr;
  int* d_series_start = nullptr;
  double* d_ws_prev = nullptr;
  // shared-wind pair-coefficient table
  float* d_pair_tab = nullptr;
  int* d_pair_first = nullptr;  // per source: first sorted target index with dx >= 0
  bool pair_dirty = true;
  bool no_pair_table = false;  // WF_NO_PAIR_TABLE (A/B runs), read once at wf_create
  bool no_ll_fly = false;      // WF_LL_FLY=0 (A/B runs): a wind per farm stays on wf_step_kernel
  bool ws_prev_valid = false;  // d_ws_prev holds the free wind of the state before the coming env step (one use)
  bool shared_dir = false;  // one wind per farm, but the same direction for all: shared geometry + pair table
  // Direction groups: farms partitioned by a small set of K distinct wind directions (series rows, binned reset
  // directions); one sorted geometry + pair table per group, farms launched group by group (padded to whole blocks)
  int n_groups = 0;            // 0 = ungrouped
  int group_shift = 0;         // geometry / table of group g is (g + group_shift) % n_groups  (series: the tick)
  int n_blocks = 0;            // entries of d_blk_group (one per group_unit farms)
  int n_slots = 0;             // launch slots of the grouped launch (entries of d_perm)
  int *d_perm = nullptr, *d_blk_group = nullptr;
  size_t perm_cap = 0, blk_cap = 0;
  size_t pair_groups_cap = 0;  // groups the pair-table allocation holds
  double* d_group_wd = nullptr;  // [K] direction of each group (binned sampling; series mode uses d_series_wd)
  double grid_step = 0.0;      // binned sampling: direction grid the cached group geometry / tables were built for
  int* d_bins = nullptr;       // [B] bin of each farm (binned sampling)
  // One-block-at-a-time kernel (wf_kernels_ll.hip) for the pair-table path of farms with several lane-group blocks:
  // its own table layout, the per-farm source log, and the per-direction flag that hands a direction with x' ties
  // across a block boundary back to wf_step_kernel
  int ll_G = 0, ll_S = 1;      // lanes per farm and target slots per lane of that kernel; ll_G = 0: not used
  float* d_ll_tab = nullptr;   // [groups][wfk_ll_table_floats]
  int* d_ll_flag = nullptr;    // [groups] 1 = cross-block tie
  float* d_src_log = nullptr;  // [launch slots][N][WF_LOG_FLOATS], then [launch slots][N][WF_LOG_SIDE_FLOATS]
  size_t ll_groups_cap = 0, log_slots_cap = 0;
  int* d_farm_tie = nullptr;   // [B] + 1: per-farm cross-block-tie flag of the per-farm geometry, then the "any" flag
  int farm_ties = 2;           // a wind per farm: 0 no farm has such a tie, 1 some have, 2 not read back
  bool wind_sync = true;       // the wind was set by a call that synchronises anyway (host arrays, series, binned sampling)
  int ll_ties = 2;             // cross-block ties of the current directions: 0 none, 1 all of them, 2 some / not read back
  // float64 re-solve of the farms the float32 kernels flag (wf_resolve.hip)
  int resolve_mode = 0;        // 0 off, 1 flagged farms, 2 every farm (forced when the model has wind_veer != 0)
  WfResolveConsts rconsts{};
  double* d_tab64 = nullptr;   // [3][WF_TABLE_PAD] wind speed, Ct, power in float64
  int *d_res_list = nullptr, *d_res_count = nullptr, *d_flags_raw = nullptr;  // [B], [1], [B]
};

namespace {

// Every entry point runs on the handle's device and leaves the caller's current device as it found it (a torch
// process would otherwise see torch.cuda.current_device() change under it).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != device) {
      err = hipSetDevice(device);
      switched = err == hipSuccess;
    }
  }
  ~DeviceGuard() {
    if (switched) hipSetDevice(prev);
  }
};
#define WF_ON_DEVICE(h) \
  DeviceGuard guard_((h)->device); \
  if (guard_.err != hipSuccess) return fail(h, WF_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard_.err))

int fail(wf_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return code;
}
#define WF_HIP(h, call)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) return fail(h, WF_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

void free_batch(wf_handle* h) {
  hipFree(h->d_ws); hipFree(h->d_wd); hipFree(h->d_gx); hipFree(h->d_gy); hipFree(h->d_gidx); hipFree(h->d_flags);
  hipFree(h->d_farm_tie);
  hipFree(h->d_res_list); hipFree(h->d_res_count); hipFree(h->d_flags_raw);
  h->d_res_list = h->d_res_count = h->d_flags_raw = nullptr;
  h->d_flags = h->d_farm_tie = nullptr;
  hipFree(h->d_yaw); hipFree(h->d_out);
  hipFree(h->d_env_yaw); hipFree(h->d_env_acc); hipFree(h->d_env_act); hipFree(h->d_env_out); hipFree(h->d_env_moves);
  if (h->h_env_act) hipHostFree(h->h_env_act);
  if (h->h_env_out) hipHostFree(h->h_env_out);
  h->d_env_yaw = h->d_env_acc = h->d_env_act = h->d_env_out = h->h_env_act = h->h_env_out = nullptr;
  h->d_env_moves = nullptr;
  hipFree(h->d_series_ws); hipFree(h->d_series_wd); hipFree(h->d_series_start); hipFree(h->d_ws_prev);
  hipFree(h->d_pair_tab); hipFree(h->d_pair_first); h->d_pair_tab = nullptr; h->d_pair_first = nullptr; h->pair_dirty = true;
  hipFree(h->d_ll_tab); hipFree(h->d_ll_flag); hipFree(h->d_src_log);
  h->d_ll_tab = h->d_src_log = nullptr; h->d_ll_flag = nullptr; h->ll_groups_cap = h->log_slots_cap = 0;
  hipFree(h->d_perm); hipFree(h->d_blk_group); hipFree(h->d_group_wd); hipFree(h->d_bins);
  h->d_perm = h->d_blk_group = h->d_bins = nullptr; h->d_group_wd = nullptr;
  h->perm_cap = h->blk_cap = h->pair_groups_cap = 0; h->n_groups = 0; h->grid_step = 0.0;
  h->d_series_ws = h->d_series_wd = h->d_ws_prev = nullptr; h->d_series_start = nullptr; h->series_T = 0;
  if (h->h_yaw) hipHostFree(h->h_yaw);
  if (h->h_out) hipHostFree(h->h_out);
  h->d_ws = h->d_wd = h->d_gx = h->d_gy = nullptr; h->d_gidx = nullptr;
  h->d_yaw = h->d_out = h->h_yaw = h->h_out = nullptr;
  h->cap_env = h->cap_bn = 0;
}

// Kernel variant for N turbines and B farms: G lanes per farm, S target slots per lane, G*S >= N.
// Throughput regime (the grid fills the chip): smaller G wastes fewer lanes on the triangular
// (upstream->downstream) structure and amortises the per-source work over more farms per wave; S is bounded by
// the 256-VGPR budget that keeps two waves per SIMD resident.
// Latency regime (few farms, e.g. the reference's single-farm env): the chip is not full anyway, so G is widened
// as long as all waves still fit in one residency round — fewer slot passes per source step.
int find_variant(int G, int S) {
  for (int i = 0; i < wfk_num_variants(); ++i) {
    int g, s; const void* fn;
    wfk_variant(i, &g, &s, &fn);
    if (g == G && s == S) return i;
  }
  return -1;
}

int pick_variant(int N, int B) {
  const char* ov = getenv("WF_KERNEL_GS");  // tuning override, e.g. "16x5"
  int og = 0, os = 0;
  if (ov && sscanf(ov, "%dx%d", &og, &os) == 2 && og * os >= N) {
    const int v = find_variant(og, os);
    if (v >= 0) return v;
  }
  static const int pref[][3] = {  // {max N, G, S}
      {4, 4, 1}, {8, 4, 2}, {12, 4, 3}, {16, 4, 4}, {24, 8, 3}, {32, 8, 4}, {48, 16, 3}, {64, 16, 4},
      {80, 16, 5}, {96, 16, 6}, {128, 32, 4}, {192, 64, 3}, {256, 64, 4}};
  int G = 0, S = 0;
  for (auto& r : pref)
    if (N <= r[0]) { G = r[1]; S = r[2]; break; }
  if (!G) return -1;
  if (B > 0) {
    const long resident = 256L * 4 * 2;  // waves the chip holds at two per SIMD
    while (G < 64) {
      const int g2 = G * 2, s2 = (N + g2 - 1) / g2;
      if ((long)B * g2 / 64 > resident / 2 || find_variant(g2, s2) < 0) break;
      G = g2; S = s2;
    }
  }
  return find_variant(G, S);
}

struct LlFamily { int code, farms_per_block, per_cu; double t[3]; };  // code = (G << 4) | S, 0 = wf_step_kernel<16,5>
const LlFamily kLlFamilies[] = {{0, 16, 2, {0.235, 0.298, 0.0}},
                                {(8 << 4) | 1, 32, 3, {0.33, 0.42, 0.55}},
                                {(4 << 4) | 2, 64, 2, {0.49, 0.644, 0.0}},
                                {(4 << 4) | 1, 64, 3, {0.53, 0.67, 0.89}},
                                {(2 << 4) | 2, 128, 2, {0.855, 1.178, 0.0}}};
// rounds model of pick_ll below: ms (at N = 80) for `farms` farm slots
double ll_estimate(const LlFamily& f, long farms) {
  const long blocks = (farms + f.farms_per_block - 1) / f.farms_per_block, per_round = 256L * f.per_cu;
  const long full = blocks / per_round, rem = blocks % per_round;
  double t = full * f.t[f.per_cu - 1];
  if (rem) t += (full ? 0.8 : 1.0) * f.t[(rem + 255) / 256 - 1];
  return t;
}

// Lane-group width of the one-block-at-a-time kernel for N turbines and B farms, 0 = keep wf_step_kernel.  It pays once
// the farm spans several blocks (the register-slot kernel is then pinned at two waves per SIMD by its 27 S state
// registers) and the batch fills the chip; WF_LL=0 disables it, WF_LL_G=<4|8|16> forces a width (A/B runs).
int pick_ll(int N, int B) {  // returns (G << 4) | S, 0 = keep wf_step_kernel
  const char* off = getenv("WF_LL");
  if (off && off[0] == '0') return 0;
  if (N > WF_PAIR_MAX_N) return 0;
  const char* force = getenv("WF_LL_G");  // "8" or "4x2"
  if (force) {
    int g = 0, sl = 1;
    if (sscanf(force, "%dx%d", &g, &sl) < 1) return 0;
    const bool ok = ((g == 4 || g == 8 || g == 16) && sl == 1) || ((g == 4 || g == 2) && sl == 2);
    return (ok && N > g * sl) ? ((g << 4) | sl) : 0;
  }
  // A wave solves its 64 / G farms start to finish, so a launch runs in ROUNDS of (blocks the chip holds) x (farms per
  // block), and within a round the time depends on how many blocks share a CU (one wave per SIMD each).  Measured at
  // N = 80 (profiles/r02_v24_batch_sweep_variants.txt; ms for 1, 2, 3 blocks per CU; the ratios hold at N = 91):
  //   wf_step_kernel<16,5>  16 farms per block, 2 per CU: 0.235 0.298
  //   G = 8                 32 farms per block, 3 per CU: 0.33  0.42  0.55
  //   G = 4, two slots      64 farms per block, 2 per CU: 0.49  0.644
  //   G = 4                 64 farms per block, 3 per CU: 0.53  0.67  0.89
  //   G = 2, two slots     128 farms per block, 2 per CU: 0.855 1.178   (thirty-two farms per wave share the per-source
  //                         phase; twice the log re-reads of G = 4 x 2: pays only on full rounds of 65536 farms)
  // A partial round behind full ones overlaps with their tail (factor 0.8).  The estimates are within 6 % of the sweep
  // (4096 ... 131072 farms); the cheapest wins: the register-slot kernel up to ~8192 farms, G = 8 up to ~24576, then
  // the two G = 4 kernels depending on how the batch divides into rounds of 32768 / 49152.
  if (N <= 16) return 0;
  int best = 0;
  double t_best = 1e300;
  for (const LlFamily& f : kLlFamilies) {
    if (f.code && N <= (f.code >> 4) * (f.code & 15)) continue;  // needs more than one block
    if (f.code == ((8 << 4) | 1) && N <= 32) continue;           // (not instantiated to pay below that)
    if (f.code == ((2 << 4) | 2) && N < 48) continue;            // (measured at N = 80 and 91 only)
    const double t = ll_estimate(f, B);
    if (t < t_best) { t_best = t; best = f.code; }
  }
  return best;
}

// A grouped launch (series rows / binned directions) pads every group to whole blocks: more farm slots than farms.  The
// two G = 4 kernels have the same block size, so the choice between them can follow the padded count without touching
// the group lists (HornsRev1 x 65536 in 104 groups = 1072 blocks: three rounds of the two-slot kernel, 1.65 ms, against
// two of the one-slot kernel).
int repick_ll_slots(int N, int ll_G, int ll_S, long farm_slots) {
  if (ll_G != 4 || getenv("WF_LL_G") || N <= 8) return ll_S;
  return ll_estimate(kLlFamilies[3], farm_slots) < ll_estimate(kLlFamilies[2], farm_slots) ? 1 : 2;
}

// The one-block kernel's (G, S) of a handle: its pair table and source log are laid out for (N, G, S).  The caller has
// made sure no launch is in flight.
void set_ll_shape(wf_handle* h, int G, int S) {
  if (G == h->ll_G && S == h->ll_S) return;
  hipFree(h->d_ll_tab); hipFree(h->d_ll_flag); hipFree(h->d_src_log);
  h->d_ll_tab = h->d_src_log = nullptr; h->d_ll_flag = nullptr; h->ll_groups_cap = h->log_slots_cap = 0;
  h->ll_G = G; h->ll_S = S; h->pair_dirty = true;
}

// leaving a grouped launch: back to the choice for the plain batch
void ungroup(wf_handle* h) {
  if (h->n_groups > 0 && h->ll_G) {
    const int llg = pick_ll(h->N, h->B);
    if (llg && ((llg >> 4) != h->ll_G || (llg & 15) != h->ll_S)) {
      hipStreamSynchronize(h->stream);
      set_ll_shape(h, llg >> 4, llg & 15);
    }
  }
  h->n_groups = 0;
}

int build_consts(wf_handle* h) {
  const wf_model_params& m = h->model;
  const int n = (int)h->tws.size();
  if (n < 2 || n > WF_MAX_TABLE - 1) return fail(h, WF_E_INVALID, "power_thrust_table needs 2..63 entries");
  for (int i = 1; i < n; ++i)
    if (!(h->tws[i] > h->tws[i - 1])) return fail(h, WF_E_INVALID, "table wind speeds must be strictly ascending");
  if (!(m.rotor_diameter > 0) || !(m.hub_height > m.rotor_diameter / 2))
    return fail(h, WF_E_INVALID, "need rotor_diameter > 0 and hub_height > rotor radius");

  WfConsts& c = h->consts;
  const double D = m.rotor_diameter, HH = m.hub_height, R = D / 2, eps = m.eps_gain * D, eps2 = eps * eps;
  c.N = h->N;
  c.D = (float)D; c.invD = (float)(1.0 / D);
  const double off[3] = {-D / 4, 0.0, D / 4};
  double shearf[3], uinf = 0;
  for (int k = 0; k < 3; ++k) {
    c.off[k] = (float)off[k];
    c.yoff[k] = (float)(off[k] + m.num_eps);
    shearf[k] = std::pow((HH + off[k]) / HH, m.shear);
    c.shearf[k] = (float)shearf[k];
    uinf += shearf[k] / 3.0;
  }
  c.uinf_f = (float)uinf;
  for (int k = 0; k < 3; ++k) {
    const double z = HH + off[k];
    const double dudz = m.shear * std::pow(1.0 / HH, m.shear) * std::pow(z, m.shear - 1.0);  // per unit ws
    const double lm = m.kappa * z / (1.0 + m.kappa * z / (D / 8.0));
    const double nu = lm * lm * std::fabs(dudz);
    c.decay_a[k] = (float)(4.0 * nu / uinf / eps2);
  }
  c.exp_c = (float)(1.4426950408889634 / eps2);
  const double m_eps = m.num_eps;
  const double q = D / 4.0;
  for (int m = -3; m <= 3; ++m) {
    const double zc = m * q + m_eps, zm = 2.0 * HH + m * q + m_eps;
    c.zc[m + 3] = (float)zc; c.zc2[m + 3] = (float)(zc * zc); c.ez[m + 3] = (float)std::exp(-zc * zc / eps2);
    c.zm[m + 3] = (float)zm; c.zm2[m + 3] = (float)(zm * zm); c.ezm[m + 3] = (float)std::exp(-zm * zm / eps2);
  }
  // 1 - Ey*ezm with Ey <= 1 rounds to exactly 1.0f once ezm < 2^-25: those classes skip the core factor
  c.mirror_core_n = 0;
  for (int m = 0; m < 7; ++m)
    if (c.ezm[m] >= 2.9e-8f) c.mirror_core_n = m + 1;
  const double hs[3] = {HH + R, HH - R, HH};
  double ks[3] = {0, 0, 0};
  for (int v = 0; v < 3; ++v)
    for (int k = 0; k < 3; ++k) {
      const double zc = HH + off[k] - hs[v] + m_eps;
      for (int j = 0; j < 3; ++j) {  // secondary-steering means on the source's own grid [A.3-2]
        const double yL = off[j] + m_eps;
        const double r = yL * yL + zc * zc;
        ks[v] += zc / r * (1.0 - std::exp(-r / eps2)) / 9.0;
      }
    }
  c.ks_top = (float)ks[0]; c.ks_bot = (float)ks[1]; c.ks_core = (float)ks[2];
  const double vel_top = std::pow((HH + R) / HH, m.shear), vel_bot = std::pow((HH - R) / HH, m.shear);
  const double inv2pi = 1.0 / (2.0 * M_PI);
  c.gam_top = (float)(inv2pi * (M_PI / 8.0) * D * vel_top * uinf);
  c.gam_bot = (float)(inv2pi * (M_PI / 8.0) * D * vel_bot * uinf);
  c.gam_wr = (float)(inv2pi * 0.25 * 2.0 * M_PI * D / m.tsr);
  c.alpha4 = (float)(4.0 * m.alpha); c.beta2 = (float)(2.0 * m.beta);
  c.ka = (float)m.ka; c.kb = (float)m.kb; c.ad = (float)m.ad; c.bd = (float)m.bd; c.dm03 = (float)(0.3 * m.dm);
  c.alpha4_d = (float)(4.0 * m.defl_alpha); c.beta2_d = (float)(2.0 * m.defl_beta);
  c.ka_d = (float)m.defl_ka; c.kb_d = (float)m.defl_kb;
  c.sw_steer = m.enable_secondary_steering ? 2.0f : 0.0f;
  c.sw_tv = m.enable_transverse_velocities ? 1.0f : 0.0f;
  c.e0c1 = (float)(3.0 * std::exp(1.0 / 12.0)); c.e0c2 = (float)(3.0 * std::exp(1.0 / 3.0));
  c.sz0v = (float)(D / (2.0 * std::sqrt(2.0)));
  c.near_c = (float)(m.near_wake_c * D);
  c.kdef = (float)(D * D / 8.0);
  c.ch_c = (float)(m.ch_constant * std::pow(m.ambient_ti, m.ch_initial));
  c.ch_ai = (float)m.ch_ai; c.ch_down = (float)m.ch_downstream;
  c.amb = (float)m.ambient_ti; c.amb2 = (float)(m.ambient_ti * m.ambient_ti);
  c.gch_gain = m.enable_yaw_added_recovery ? (float)m.gch_gain : 0.0f; c.overlap_thr = (float)m.overlap_thresh;
  c.twoD = (float)(2.0 * D); c.fifteenD_d = 15.0 * D;
  c.q_d = D / 4.0;
  c.guard_inv = h->guard_rel > 0.0 ? (float)(1.0 / h->guard_rel) : 1125899906842624.0f;
  c.inv_overlap_thr = (float)(1.0 / m.overlap_thresh);
  c.yc_d = h->yc;
  c.ct_kappa = 5.0f;     // nrel_5MW: 5.9 on the cut-in ramp (2.5-3 m/s), 143 on the cut-out drop, <= 4.0 everywhere else
  c.knee_kappa = 30.0f;  // 30 x (wind-speed error ~3e-6) ~ 1e-4 of max(P, 1 kW)
  c.rho = (float)m.ref_density; c.pw = (float)(m.pP / 3.0);
  c.dens_f = (float)std::cbrt(m.air_density / m.ref_density);

  // table + bucket index
  WfTables t;
  const double area = M_PI * R * R;
  std::vector<double> pwv(n);
  for (int i = 0; i < n; ++i) pwv[i] = 0.5 * area * h->tcp[i] * m.gen_eff * h->tws[i] * h->tws[i] * h->tws[i];
  for (int i = 0; i < WF_TABLE_PAD; ++i) {
    const bool in = i < n;
    t.knot[i] = in ? (float)h->tws[i] : 3.0e38f;
    t.ct[i] = in ? (float)h->tct[i] : 0.f;
    t.pw[i] = in ? (float)pwv[i] : 0.f;
    const bool seg = i + 1 < n;
    t.ct_slope[i] = seg ? (float)((h->tct[i + 1] - h->tct[i]) / (h->tws[i + 1] - h->tws[i])) : 0.f;
    t.pw_slope[i] = seg ? (float)((pwv[i + 1] - pwv[i]) / (h->tws[i + 1] - h->tws[i])) : 0.f;
  }
  const double x0 = h->tws[0], x1 = h->tws[n - 1];
  const double bh = (x1 - x0) / WF_BUCKETS;
  c.n_table = n; c.bucket_x0 = (float)x0; c.bucket_h_inv = (float)(1.0 / bh);
  // bucket[b] = last knot <= start of bucket b-1; the kernel probes forward from there.  One bucket of
  // slack on either side absorbs float rounding of the bucket index computed on the device.
  int max_probe = 1;
  for (int b = 0; b < WF_BUCKETS; ++b) {
    const double lo = x0 + (b - 1) * bh, hi = x0 + (b + 2) * bh;
    int j = 0;
    while (j + 1 < n && h->tws[j + 1] <= lo) ++j;
    t.bucket[b] = (unsigned char)j;
    int k = j;
    while (k + 1 < n && h->tws[k + 1] <= hi) ++k;
    if (k - j > max_probe) max_probe = k - j;
  }
  c.max_probe = max_probe;
  // the float64 solve (wf_resolve.hip): the same model in double
  {
    WfResolveConsts& r = h->rconsts;
    r.N = h->N; r.n_table = n;
    r.sw_steer = m.enable_secondary_steering ? 1 : 0; r.sw_yar = m.enable_yaw_added_recovery ? 1 : 0;
    r.sw_tv = m.enable_transverse_velocities ? 1 : 0;
    r.D = D; r.HH = HH; r.TSR = m.tsr; r.amb = m.ambient_ti; r.eps2 = eps2; r.num_eps = m.num_eps; r.sqrt2 = std::sqrt(2.0);
    r.uinf1 = 0.0;
    for (int k = 0; k < 3; ++k) {
      r.off[k] = off[k];
      r.shearf[k] = shearf[k];
      r.uinf1 += shearf[k];
      const double z = HH + off[k];
      const double dudz1 = m.shear * std::pow(1.0 / HH, m.shear) * std::pow(z, m.shear - 1.0);
      const double lm = m.kappa * z / (1.0 + m.kappa * z / (D / 8.0));
      r.nu1[k] = lm * lm * std::fabs(dudz1);
    }
    r.uinf1 /= 3.0;
    r.vel_top = vel_top; r.vel_bot = vel_bot;
    double kk[3] = {0, 0, 0};
    for (int v = 0; v < 3; ++v)
      for (int k = 0; k < 3; ++k) {
        const double zc = HH + off[k] - hs[v] + m_eps;
        for (int j = 0; j < 3; ++j) {
          const double yL = off[j] + m_eps;
          const double rr = yL * yL + zc * zc;
          kk[v] += zc / (2.0 * M_PI * rr) * (1.0 - std::exp(-rr / eps2)) / 9.0;
        }
      }
    r.mirror_core = 0;
    for (int mm = -3; mm <= 3; ++mm) {
      const double zc = mm * q + m_eps, zm = 2.0 * HH + mm * q + m_eps;
      r.zc[mm + 3] = zc; r.zc2[mm + 3] = zc * zc; r.ezc[mm + 3] = std::exp(-zc * zc / eps2);
      r.zm7[mm + 3] = zm; r.zm2[mm + 3] = zm * zm; r.ezm7[mm + 3] = std::exp(-zm * zm / eps2);
      if (r.ezm7[mm + 3] >= 1.0e-17) r.mirror_core = 1;
    }
    r.k_top = kk[0]; r.k_bot = kk[1]; r.k_core = kk[2];
    r.alpha = m.alpha; r.beta = m.beta; r.ka = m.ka; r.kb = m.kb; r.ad = m.ad; r.bd = m.bd; r.dm = m.dm;
    r.defl_alpha = m.defl_alpha; r.defl_beta = m.defl_beta; r.defl_ka = m.defl_ka; r.defl_kb = m.defl_kb;
    r.e0c1 = 3.0 * std::exp(1.0 / 12.0); r.e0c2 = 3.0 * std::exp(1.0 / 3.0);
    r.near_c = m.near_wake_c * D;
    r.ch_constant = m.ch_constant; r.ch_ai = m.ch_ai; r.ch_amb_pow = std::pow(m.ambient_ti, m.ch_initial); r.ch_down = m.ch_downstream;
    r.gch_gain = m.gch_gain; r.overlap_thr = m.overlap_thresh;
    const double vr = m.veer * M_PI / 180.0;
    r.cos_veer = std::cos(vr); r.cos2_veer = std::cos(vr) * std::cos(vr); r.sin2_veer = std::sin(vr) * std::sin(vr);
    r.sin_2veer = std::sin(2.0 * vr);
    r.rho_ref = m.ref_density; r.dens_cbrt = std::pow(m.air_density / m.ref_density, 1.0 / 3.0); r.pP3 = m.pP / 3.0;
    std::vector<double> t64(3 * WF_TABLE_PAD, 0.0);
    for (int i = 0; i < n; ++i) { t64[i] = h->tws[i]; t64[WF_TABLE_PAD + i] = h->tct[i]; t64[2 * WF_TABLE_PAD + i] = pwv[i]; }
    if (!h->d_tab64) {
      hipError_t e64 = hipMalloc(&h->d_tab64, sizeof(double) * 3 * WF_TABLE_PAD);
      if (e64 != hipSuccess) return fail(h, WF_E_HIP, std::string("table64 alloc: ") + hipGetErrorString(e64));
    }
    hipError_t e64 = hipMemcpy(h->d_tab64, t64.data(), sizeof(double) * t64.size(), hipMemcpyHostToDevice);
    if (e64 != hipSuccess) return fail(h, WF_E_HIP, std::string("table64 upload: ") + hipGetErrorString(e64));
  }
  hipError_t e = hipMemcpyAsync(h->d_tab, &t, sizeof(WfTables), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return fail(h, WF_E_HIP, std::string("table upload: ") + hipGetErrorString(e));
  h->model_dirty = false;
  return WF_OK;
}

// Shared wind: (re)build the geometry-only pair table after the geometry kernel (same stream).  Returns the table
// pointer to hand to the step kernel, or nullptr when the on-the-fly path applies (per-farm wind, N too large,
// or WF_NO_PAIR_TABLE set for A/B runs).
int pair_table(wf_handle* h, const float** out) {
  *out = nullptr;
  if ((h->wind_count != 1 && !h->shared_dir && h->n_groups == 0) || h->N > WF_PAIR_MAX_N || !wfk_variant_has_table(h->variant) || h->no_pair_table)
    return WF_OK;
  int vG, vS; const void* vfn;
  wfk_variant(h->variant, &vG, &vS, &vfn);
  const int NP = vG * vS;
  const size_t ng = h->n_groups > 0 ? (size_t)h->n_groups : 1;
  if (!h->d_pair_tab || h->pair_groups_cap < ng) {
    hipFree(h->d_pair_tab); hipFree(h->d_pair_first);
    h->d_pair_tab = nullptr; h->d_pair_first = nullptr; h->pair_groups_cap = 0;
    WF_HIP(h, hipMalloc(&h->d_pair_tab, sizeof(float) * ng * h->N * WF_PAIR_ROW_FLOATS(NP)));
    WF_HIP(h, hipMalloc(&h->d_pair_first, sizeof(int) * ng * h->N));
    h->pair_groups_cap = ng;
    h->pair_dirty = true;
  }
  if (h->pair_dirty) {
    const wf_model_params& m = h->model;
    WfPairConsts pc{};
    const double D = m.rotor_diameter, HH = m.hub_height, eps = m.eps_gain * D;
    pc.N = h->N; pc.NP = NP; pc.D = D; pc.HH = HH; pc.eps2 = eps * eps; pc.num_eps = m.num_eps; pc.ch_down = m.ch_downstream;
    const double off[3] = {-D / 4, 0.0, D / 4};
    double uinf = 0;
    for (int k = 0; k < 3; ++k) uinf += std::pow((HH + off[k]) / HH, m.shear) / 3.0;
    pc.fifteenD = 15.0 * D;
    pc.twoD = 2.0 * D;
    pc.gam_top = (1.0 / 16.0) * D * std::pow((HH + D / 2) / HH, m.shear) * uinf;  // (1/2pi)(pi/8) D vel_top uinf
    pc.gam_bot = (1.0 / 16.0) * D * std::pow((HH - D / 2) / HH, m.shear) * uinf;
    for (int k = 0; k < 3; ++k) {
      pc.off[k] = off[k];
      const double z = HH + off[k];
      const double dudz = m.shear * std::pow(1.0 / HH, m.shear) * std::pow(z, m.shear - 1.0);
      const double lm = m.kappa * z / (1.0 + m.kappa * z / (D / 8.0));
      pc.decay_a[k] = 4.0 * lm * lm * std::fabs(dudz) / uinf / pc.eps2;
    }
    WF_HIP(h, wfk_launch_pair_table(&pc, (int)ng, h->d_gx, h->d_gy, h->d_pair_tab, h->d_pair_first, h->stream));
    if (h->ll_G) {  // the same records in target-block order, and the per-direction cross-block-tie flag
      if (!h->d_ll_tab || h->ll_groups_cap < ng) {
        hipFree(h->d_ll_tab); hipFree(h->d_ll_flag);
        h->d_ll_tab = nullptr; h->d_ll_flag = nullptr; h->ll_groups_cap = 0;
        WF_HIP(h, hipMalloc(&h->d_ll_tab, sizeof(float) * ng * wfk_ll_table_floats(h->N, h->ll_G * h->ll_S)));
        WF_HIP(h, hipMalloc(&h->d_ll_flag, sizeof(int) * ng));
        h->ll_groups_cap = ng;
      }
      WF_HIP(h, wfk_launch_pair_table_ll(&pc, h->ll_G * h->ll_S, (int)ng, h->d_gx, h->d_gy, h->d_ll_tab, h->d_ll_flag, h->stream));
      // which kernel serves which direction is decided on the device (no host round trip on the asynchronous path);
      // where the wind came through a synchronising call anyway, the flags are read back once so that a launch nobody
      // needs is not enqueued at all
      h->ll_ties = 2;
      if (h->wind_sync) {
        std::vector<int> f(ng);
        WF_HIP(h, hipMemcpyAsync(f.data(), h->d_ll_flag, sizeof(int) * ng, hipMemcpyDeviceToHost, h->stream));
        WF_HIP(h, hipStreamSynchronize(h->stream));
        size_t tied = 0;
        for (int v : f) tied += v != 0;
        h->ll_ties = tied == 0 ? 0 : (tied == ng ? 1 : 2);
      }
    }
    h->pair_dirty = false;
  }
  *out = h->d_pair_tab;
  return WF_OK;
}

// Rotation + sort of `n_env` wind conditions on the handle's stream.  A geometry per farm (n_env == B) also yields the
// per-farm cross-block-tie flags for the on-the-fly one-block kernel; sync_ok: the caller synchronises anyway, so the
// "any farm tied" flag is read back and a launch nobody needs is never enqueued.
int ll_fly_S(const wf_handle* h);
int ll_fly_G(const wf_handle* h);
int run_geometry(wf_handle* h, int n_env, const double* d_wd, bool sync_ok) {
  const bool per_farm = n_env == h->B && h->B > 1 && h->ll_G && wfk_ll_has_fly(ll_fly_G(h), ll_fly_S(h));
  WF_HIP(h, wfk_launch_geometry(n_env, h->N, h->d_lx, h->d_ly, h->xc, h->yc, d_wd, h->d_gx, h->d_gy, h->d_gidx,
                                per_farm ? ll_fly_G(h) * ll_fly_S(h) : 0, h->d_farm_tie, h->d_farm_tie ? h->d_farm_tie + h->B : nullptr,
                                h->stream));
  h->farm_ties = 2;
  if (per_farm && sync_ok) {
    int any = 0;
    WF_HIP(h, hipMemcpyAsync(&any, h->d_farm_tie + h->B, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    WF_HIP(h, hipStreamSynchronize(h->stream));
    h->farm_ties = any ? 1 : 0;
  }
  return WF_OK;
}

// Target slots per lane of the one-block kernel ON THE FLY (a wind per farm): two at G = 4 whatever the table path
// uses — there the second slot halves the per-source geometry work as well (HornsRev1 x 65536: 3.48 ms against 4.14).
int ll_fly_S(const wf_handle* h) { return h->ll_G <= 4 ? 2 : h->ll_S; }
// ... and its lane-group width: the table path's, except that G = 2 has no on-the-fly instantiation (G = 4 x 2 serves)
int ll_fly_G(const wf_handle* h) { return h->ll_G == 2 ? 4 : h->ll_G; }

// turbines per farm in the source log of the one-block kernel: whole lane-group blocks (of the larger of the two
// block sizes: the table path and the on-the-fly path share the buffer)
size_t ll_npad(const wf_handle* h) {
  const int a = h->ll_G * h->ll_S, b = ll_fly_G(h) * ll_fly_S(h), gs = a > b ? a : b;  // (powers of two)
  return (size_t)((h->N + gs - 1) / gs) * gs;
}

int ll_log_fpb(const wf_handle* h) {
  const int a = wfk_ll_farms_per_block(h->ll_G), b = wfk_ll_farms_per_block(ll_fly_G(h));
  return a > b ? a : b;
}

// Farms per block of the table-path launch of the handle's kernel variant (wf_step_kernel), and of the
// one-block-at-a-time kernel when it is in use.  A grouped launch pads every group to a multiple of the larger of the
// two (both are powers of two), and its block -> group list has one entry per `group_unit` farms (the smaller).
int farms_per_block(const wf_handle* h) {
  int vG, vS; const void* vfn;
  wfk_variant(h->variant, &vG, &vS, &vfn);
  return wfk_tab_waves() * (64 / vG);
}
int group_pad(const wf_handle* h) {
  // (a grouped launch never runs the G = 2 kernel: build_groups)
  const int a = farms_per_block(h), b = h->ll_G ? wfk_ll_farms_per_block(h->ll_G == 2 ? 4 : h->ll_G) : 0;
  return a > b ? a : b;
}
int group_unit(const wf_handle* h) {
  const int a = farms_per_block(h), b = h->ll_G ? wfk_ll_farms_per_block(h->ll_G == 2 ? 4 : h->ll_G) : a;
  return a < b ? a : b;
}

// Would a grouped launch over K direction groups pay off?  Every group is padded to whole blocks (half a block wasted
// per group on average) against the ~2x cost of the on-the-fly path.
bool groups_pay_off(const wf_handle* h, int K) {
  if (h->N > WF_PAIR_MAX_N || !wfk_variant_has_table(h->variant) || h->no_pair_table || K < 1) return false;
  if ((size_t)K * h->N > h->cap_bn) return false;  // group geometry lives in the per-farm geometry buffers
  const double waste = 0.5 * group_pad(h) * K / (double)h->B;
  int vG, vS; const void* vfn;
  wfk_variant(h->variant, &vG, &vS, &vfn);
  const size_t bytes = (size_t)K * h->N * WF_PAIR_ROW_FLOATS(vG * vS) * sizeof(float);
  return waste < 0.5 && bytes <= ((size_t)8 << 30);
}

// Partition the farms by `group_of_farm` (host, B entries in [0, K)): farm list sorted by group and padded per group to
// whole blocks (d_perm, -1 = padding), group of each block (d_blk_group).  Then the sorted geometry of the K
// directions `d_wd_groups` (device) is built into the geometry buffers; the pair tables follow lazily (pair_table()).
int build_groups(wf_handle* h, const int* group_of_farm, int K, const double* d_wd_groups, bool rebuild_geometry) {
  // the 128-farm blocks of the G = 2 kernel would double the padding of every group: grouped launches use G = 4
  // (the choice between its two kernels follows the padded count, below)
  if (h->ll_G == 2) {  // (also when WF_LL_G forces it for the plain batch: the group lists are laid out in 64-farm blocks)
    WF_HIP(h, hipStreamSynchronize(h->stream));
    set_ll_shape(h, 4, 2);
  }
  const int epb = group_pad(h), unit = group_unit(h);
  std::vector<int> count(K, 0);
  for (int b = 0; b < h->B; ++b) {
    if (group_of_farm[b] < 0 || group_of_farm[b] >= K) return fail(h, WF_E_INVALID, "direction group out of range");
    ++count[group_of_farm[b]];
  }
  std::vector<int> first_slot(K, 0), blk_group;
  int slots = 0;
  for (int g = 0; g < K; ++g) {
    first_slot[g] = slots;
    const int nb = (count[g] + epb - 1) / epb;
    for (int q = 0; q < nb * (epb / unit); ++q) blk_group.push_back(g);
    slots += nb * epb;
  }
  std::vector<int> perm(slots > 0 ? slots : 1, -1), cursor(first_slot);
  for (int b = 0; b < h->B; ++b) perm[cursor[group_of_farm[b]]++] = b;
  if (perm.size() > h->perm_cap) {
    hipFree(h->d_perm); h->d_perm = nullptr; h->perm_cap = 0;
    WF_HIP(h, hipMalloc(&h->d_perm, sizeof(int) * perm.size()));
    h->perm_cap = perm.size();
  }
  if (blk_group.size() > h->blk_cap) {
    hipFree(h->d_blk_group); h->d_blk_group = nullptr; h->blk_cap = 0;
    WF_HIP(h, hipMalloc(&h->d_blk_group, sizeof(int) * blk_group.size()));
    h->blk_cap = blk_group.size();
  }
  WF_HIP(h, hipStreamSynchronize(h->stream));  // a launch in flight may still read the previous lists
  WF_HIP(h, hipMemcpy(h->d_perm, perm.data(), sizeof(int) * perm.size(), hipMemcpyHostToDevice));
  WF_HIP(h, hipMemcpy(h->d_blk_group, blk_group.data(), sizeof(int) * blk_group.size(), hipMemcpyHostToDevice));
  h->n_blocks = (int)blk_group.size();
  h->n_slots = slots;
  h->n_groups = K;
  h->group_shift = 0;
  {
    const int s_new = repick_ll_slots(h->N, h->ll_G, h->ll_S, (long)slots);
    if (s_new != h->ll_S) set_ll_shape(h, h->ll_G, s_new);
  }
  if (rebuild_geometry) {
    WF_HIP(h, wfk_launch_geometry(K, h->N, h->d_lx, h->d_ly, h->xc, h->yc, d_wd_groups, h->d_gx, h->d_gy, h->d_gidx, 0, nullptr, nullptr, h->stream));
    h->pair_dirty = true;
  }
  return WF_OK;
}

// One launch of the step kernel on the handle's stream with the handle's current geometry / wind / table state.
int launch_step_f32(wf_handle* h, const float* yaw, float* power, float* wspd, float* wdir, float* load, const WfEnvArgs* ea) {
  const int gstride = (h->wind_count == 1 || h->shared_dir) ? 0 : h->N;
  const int wstride = (h->wind_count == 1) ? 0 : 1;
  const float* ptab = nullptr;
  int rc = pair_table(h, &ptab);
  if (rc != WF_OK) return rc;
  WfGroupArgs ga{};
  ga.mod = 1;
  ga.risk_flags = h->d_flags;
  ga.blk_unit = 1;
  if (h->n_groups > 0) {
    ga.perm = h->d_perm; ga.blk_group = h->d_blk_group; ga.n_blocks = h->n_blocks;
    ga.shift = h->group_shift; ga.mod = h->n_groups;
    ga.blk_unit = group_unit(h); ga.n_slots = h->n_slots;
  }
  if (ptab && h->ll_G) {
    // the one-block-at-a-time kernel serves every direction without a cross-block tie; wf_step_kernel, enqueued right
    // behind it, serves the others (device-side predicate, no host round trip)
    const int fpb = ll_log_fpb(h);  // (farm slots of the log: whole blocks of the wider of the two paths' blocks)
    const size_t slots = h->n_groups > 0 ? (size_t)h->n_slots : (size_t)((h->B + fpb - 1) / fpb) * fpb;
    if (slots > h->log_slots_cap) {
      WF_HIP(h, hipStreamSynchronize(h->stream));
      hipFree(h->d_src_log); h->d_src_log = nullptr; h->log_slots_cap = 0;
      WF_HIP(h, hipMalloc(&h->d_src_log, sizeof(float) * slots * ll_npad(h) * (WF_LOG_FLOATS + WF_LOG_SIDE_FLOATS)));
      h->log_slots_cap = slots;
    }
    if (h->ll_ties != 1)
      WF_HIP(h, wfk_launch_step_ll(h->ll_G, h->ll_S, &h->consts, h->d_tab, h->d_gidx, h->d_ws, h->d_wd, wstride, yaw, power, wspd, wdir,
                                   load, h->B, ea, h->d_ll_tab, h->d_ll_flag, h->d_src_log,
                                   h->log_slots_cap * ll_npad(h) * WF_LOG_FLOATS, &ga, h->stream));
    if (h->ll_ties == 0) return WF_OK;
    ga.pred = h->d_ll_flag;
  }
  if (!ptab && gstride != 0 && h->B > 1 && h->ll_G && wfk_ll_has_fly(ll_fly_G(h), ll_fly_S(h)) && !h->no_ll_fly) {
    // a wind per farm: the one-block kernel on the fly; wf_step_kernel behind it for the farms whose own geometry has
    // an x' tie across a block boundary (per-farm device flags from the geometry kernel)
    const int fpb = ll_log_fpb(h);
    const size_t slots = (size_t)((h->B + fpb - 1) / fpb) * fpb;
    if (slots > h->log_slots_cap) {
      WF_HIP(h, hipStreamSynchronize(h->stream));
      hipFree(h->d_src_log); h->d_src_log = nullptr; h->log_slots_cap = 0;
      WF_HIP(h, hipMalloc(&h->d_src_log, sizeof(float) * slots * ll_npad(h) * (WF_LOG_FLOATS + WF_LOG_SIDE_FLOATS)));
      h->log_slots_cap = slots;
    }
    WF_HIP(h, wfk_launch_step_ll_fly(ll_fly_G(h), ll_fly_S(h), &h->consts, h->d_tab, h->d_gidx, h->d_gx, h->d_gy, h->d_ws, h->d_wd, yaw,
                                     power, wspd, wdir, load, h->B, ea, h->d_farm_tie, h->d_src_log,
                                     h->log_slots_cap * ll_npad(h) * WF_LOG_FLOATS, &ga, h->stream));
    if (h->farm_ties == 0) return WF_OK;
    ga.farm_pred = h->d_farm_tie;
  }
  WF_HIP(h, wfk_launch_step(h->variant, &h->consts, h->d_tab, h->d_gx, h->d_gy, h->d_gidx, gstride, h->d_ws, h->d_wd,
                            wstride, yaw, power, wspd, wdir, load, h->B, ea, ptab, h->d_pair_first, &ga, h->stream, &h->grid));
  return WF_OK;
}

// The step as the ABI sees it: the float32 kernels, then — when asked for (wf_set_risk_resolve) or when the model needs
// it (wind_veer != 0) — the float64 solve of the flagged (or all) farms on the same stream, overwriting their outputs.
int launch_step(wf_handle* h, const float* yaw, float* power, float* wspd, float* wdir, float* load, const WfEnvArgs* ea) {
  int rc = launch_step_f32(h, yaw, power, wspd, wdir, load, ea);
  if (rc != WF_OK) return rc;
  const int mode = h->model.veer != 0.0 ? 2 : h->resolve_mode;
  if (mode == 0) return WF_OK;
  if (!h->d_res_list) {
    WF_HIP(h, hipMalloc(&h->d_res_list, sizeof(int) * h->cap_env));
    WF_HIP(h, hipMalloc(&h->d_res_count, sizeof(int)));
    WF_HIP(h, hipMalloc(&h->d_flags_raw, sizeof(int) * h->cap_env));
  }
  WfResolveArgs ra{};
  ra.tab64 = h->d_tab64; ra.list = h->d_res_list; ra.count = h->d_res_count; ra.flags = h->d_flags;
  ra.gx = h->d_gx; ra.gy = h->d_gy; ra.gidx = h->d_gidx;
  ra.geom_stride = (h->wind_count == 1 || h->shared_dir) ? 0 : (size_t)h->N;
  ra.mod = 1;
  if (h->n_groups > 0) {
    ra.farm_group = h->series_T > 0 ? h->d_series_start : h->d_bins;
    ra.shift = h->group_shift; ra.mod = h->n_groups;
  }
  ra.ws = h->d_ws; ra.wd = h->d_wd; ra.wind_stride = h->wind_count == 1 ? 0 : 1;
  ra.yaw_in = yaw;
  ra.o_power = power; ra.o_ws = wspd; ra.o_wd = wdir; ra.o_load = load;
  if (ea) {
    ra.yaw_state = ea->yaw_state; ra.reward = ea->reward; ra.ws_prev = ea->ws_prev; ra.load_coef = ea->load_coef;
  }
  h->rconsts.N = h->N;
  WF_HIP(h, wfk_launch_resolve(&h->rconsts, &ra, h->B, mode == 2 ? 1 : 0, h->d_flags_raw, h->stream));
  return WF_OK;
}

}  // namespace

extern "C" {

int wf_version(void) { return WF_ABI_VERSION; }

int wf_default_model(wf_model_params* p) {
  if (!p) return WF_E_INVALID;
  init_default_table();
  p->air_density = 1.225; p->ambient_ti = 0.06; p->shear = 0.12; p->veer = 0.0;
  p->rotor_diameter = 126.0; p->hub_height = 90.0; p->tsr = 8.0; p->pP = 1.88; p->pT = 1.88;
  p->gen_eff = 1.0; p->ref_density = 1.225;
  p->alpha = 0.58; p->beta = 0.077; p->ka = 0.38; p->kb = 0.004; p->ad = 0.0; p->bd = 0.0; p->dm = 1.0;
  p->ch_initial = 0.1; p->ch_constant = 0.5; p->ch_ai = 0.8; p->ch_downstream = -0.32;
  p->eps_gain = 0.2; p->num_eps = 0.001; p->kappa = 0.41; p->gch_gain = 2.0; p->overlap_thresh = 0.05;
  p->near_wake_c = 0.501;
  p->defl_alpha = p->alpha; p->defl_beta = p->beta; p->defl_ka = p->ka; p->defl_kb = p->kb;
  p->enable_secondary_steering = p->enable_yaw_added_recovery = p->enable_transverse_velocities = 1;
  p->n_table = 51; p->table_ws = g_tab_ws; p->table_ct = g_tab_ct; p->table_cp = g_tab_cp;
  return WF_OK;
}

int wf_turbine_table(const char* name, int* n, const double** ws, const double** ct, const double** cp) {
  if (!name || !n || !ws || !ct || !cp) return WF_E_INVALID;
  init_default_table();
  const bool f3 = std::strcmp(name, "nrel_5MW_floris3") == 0 || std::strcmp(name, "nrel_5MW") == 0;
  if (!f3 && std::strcmp(name, "nrel_5MW_survey_a5") != 0) return WF_E_INVALID;
  *n = 51; *ws = g_tab_ws; *ct = g_tab_ct; *cp = f3 ? g_tab_cp : g_tab_cp_a5;
  return WF_OK;
}

int wf_create(int device_id, wf_handle** out) {
  if (!out) return fail(nullptr, WF_E_INVALID, "out == NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, WF_E_NODEVICE, std::string("no HIP device visible (") + hipGetErrorString(e) +
                                            "); libwfstep has no CPU fallback");
  if (device_id < 0 || device_id >= ndev) return fail(nullptr, WF_E_NODEVICE, "device id out of range");
  wf_handle* h = new (std::nothrow) wf_handle();
  if (!h) return fail(nullptr, WF_E_NOMEM, "out of host memory");
  h->device = device_id;
  h->no_pair_table = getenv("WF_NO_PAIR_TABLE") != nullptr;
  { const char* f = getenv("WF_LL_FLY"); h->no_ll_fly = f && f[0] == '0'; }
  DeviceGuard guard(device_id);
  if ((e = guard.err) != hipSuccess || (e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreate(&h->ev0)) != hipSuccess || (e = hipEventCreate(&h->ev1)) != hipSuccess ||
      (e = hipMalloc(&h->d_tab, sizeof(WfTables))) != hipSuccess) {
    std::string msg = std::string("wf_create: ") + hipGetErrorString(e);
    delete h;
    return fail(nullptr, WF_E_HIP, msg);
  }
  h->stream = h->own_stream;
  wf_model_params p;
  wf_default_model(&p);
  h->model = p;
  h->tws.assign(p.table_ws, p.table_ws + p.n_table);
  h->tct.assign(p.table_ct, p.table_ct + p.n_table);
  h->tcp.assign(p.table_cp, p.table_cp + p.n_table);
  *out = h;
  return WF_OK;
}

int wf_destroy(wf_handle* h) {
  if (!h) return WF_OK;
  DeviceGuard guard(h->device);
  hipStreamSynchronize(h->stream);
  free_batch(h);
  hipFree(h->d_tab); hipFree(h->d_tab64); hipFree(h->d_lx); hipFree(h->d_ly);
  hipEventDestroy(h->ev0); hipEventDestroy(h->ev1);
  hipStreamDestroy(h->own_stream);
  delete h;
  return WF_OK;
}

int wf_set_stream(wf_handle* h, void* s, int external) {
  if (!h) return WF_E_INVALID;
  WF_ON_DEVICE(h);
  hipStream_t next = external ? (hipStream_t)s : h->own_stream;
  if (next != h->stream) {
    WF_HIP(h, hipStreamSynchronize(h->stream));
    h->stream = next;
  }
  return WF_OK;
}
void* wf_get_stream(wf_handle* h) { return h ? (void*)h->stream : nullptr; }

int wf_set_model(wf_handle* h, const wf_model_params* p) {
  if (!h || !p) return WF_E_INVALID;
  if (p->n_table < 2 || p->n_table > WF_MAX_TABLE - 1 || !p->table_ws || !p->table_ct || !p->table_cp)
    return fail(h, WF_E_INVALID, "power_thrust_table needs 2..63 entries");
  if (!std::isfinite(p->veer)) return fail(h, WF_E_INVALID, "wind_veer must be finite");
  {
    const struct { double v; const char* name; } positive[] = {
        {p->air_density, "air_density"}, {p->ambient_ti, "turbulence_intensity"}, {p->rotor_diameter, "rotor_diameter"},
        {p->hub_height, "hub_height"}, {p->tsr, "TSR"}, {p->pP, "pP"}, {p->gen_eff, "generator_efficiency"},
        {p->ref_density, "ref_density_cp_ct"}, {p->ka * p->ambient_ti + p->kb, "ka*TI + kb"},
        {p->defl_ka * p->ambient_ti + p->defl_kb, "deflection ka*TI + kb"}, {p->defl_alpha, "deflection alpha"},
        {p->alpha, "alpha"}, {p->eps_gain, "eps_gain"}, {p->num_eps, "num_eps"}, {p->kappa, "kappa"},
        {p->ch_constant, "crespo_hernandez.constant"}, {p->overlap_thresh, "overlap_thresh"}};
    for (const auto& q : positive)
      if (!(q.v > 0.0) || !std::isfinite(q.v))
        return fail(h, WF_E_INVALID, std::string("model parameter must be finite and > 0: ") + q.name);
    const double finite[] = {p->defl_beta, p->shear, p->beta, p->ad, p->bd, p->dm, p->ch_initial, p->ch_ai, p->ch_downstream, p->gch_gain,
                             p->overlap_thresh, p->near_wake_c, p->pT};
    for (double v : finite)
      if (!std::isfinite(v)) return fail(h, WF_E_INVALID, "model parameters must be finite");
    if (!(p->hub_height > 0.5 * p->rotor_diameter))
      return fail(h, WF_E_INVALID, "hub_height must exceed the rotor radius (blade tip above ground)");
    for (int i = 0; i < p->n_table; ++i) {
      if (!std::isfinite(p->table_ws[i]) || !std::isfinite(p->table_ct[i]) || !std::isfinite(p->table_cp[i]) ||
          p->table_ct[i] < 0.0 || p->table_cp[i] < 0.0)
        return fail(h, WF_E_INVALID, "power_thrust_table entries must be finite and non-negative");
      if (i && !(p->table_ws[i] > p->table_ws[i - 1]))
        return fail(h, WF_E_INVALID, "table wind speeds must be strictly ascending");
    }
  }
  h->model = *p;
  h->tws.assign(p->table_ws, p->table_ws + p->n_table);
  h->tct.assign(p->table_ct, p->table_ct + p->n_table);
  h->tcp.assign(p->table_cp, p->table_cp + p->n_table);
  h->model.table_ws = h->model.table_ct = h->model.table_cp = nullptr;
  h->model_dirty = true;
  h->pair_dirty = true;
  return WF_OK;
}

int wf_set_layout(wf_handle* h, int n, const double* x, const double* y) {
  if (!h || !x || !y) return WF_E_INVALID;
  if (n < 1 || n > WF_MAX_TURBINES) return fail(h, WF_E_INVALID, "n_turbines must be in 1..256");
  WF_ON_DEVICE(h);
  const int v = pick_variant(n, h->B);
  if (v < 0) return fail(h, WF_E_UNSUPPORTED, "no kernel variant for this turbine count");
  const int llg = pick_ll(n, h->B);
  if ((llg >> 4) != h->ll_G || (llg ? (llg & 15) : 1) != h->ll_S || n != h->N) {  // table and source log are laid out for (N, G, S)
    hipFree(h->d_ll_tab); hipFree(h->d_ll_flag); hipFree(h->d_src_log);
    h->d_ll_tab = h->d_src_log = nullptr; h->d_ll_flag = nullptr; h->ll_groups_cap = h->log_slots_cap = 0;
    h->ll_G = llg >> 4; h->ll_S = llg ? (llg & 15) : 1;
  }
  h->lx.assign(x, x + n); h->ly.assign(y, y + n);
  double xmin = x[0], xmax = x[0], ymin = y[0], ymax = y[0];
  for (int i = 1; i < n; ++i) {
    xmin = std::fmin(xmin, x[i]); xmax = std::fmax(xmax, x[i]);
    ymin = std::fmin(ymin, y[i]); ymax = std::fmax(ymax, y[i]);
  }
  h->xc = (xmin + xmax) / 2.0; h->yc = (ymin + ymax) / 2.0;  // centre of rotation [A.1-1]
  hipFree(h->d_lx); hipFree(h->d_ly); h->d_lx = h->d_ly = nullptr;
  WF_HIP(h, hipMalloc(&h->d_lx, sizeof(double) * n));
  WF_HIP(h, hipMalloc(&h->d_ly, sizeof(double) * n));
  WF_HIP(h, hipMemcpy(h->d_lx, x, sizeof(double) * n, hipMemcpyHostToDevice));
  WF_HIP(h, hipMemcpy(h->d_ly, y, sizeof(double) * n, hipMemcpyHostToDevice));
  if (n != h->N) { free_batch(h); h->B = 0; }
  h->N = n; h->variant = v; h->wind_count = 0; h->shared_dir = false; h->model_dirty = true;
  h->n_groups = 0; h->grid_step = 0.0;
  return WF_OK;
}

int wf_set_batch(wf_handle* h, int B) {
  if (!h) return WF_E_INVALID;
  if (h->N <= 0) return fail(h, WF_E_INVALID, "wf_set_layout must be called before wf_set_batch");
  if (B < 1) return fail(h, WF_E_INVALID, "env_batch must be >= 1");
  WF_ON_DEVICE(h);
  WF_HIP(h, hipStreamSynchronize(h->stream));
  {
    const int v = pick_variant(h->N, B);
    if (v < 0) return fail(h, WF_E_UNSUPPORTED, "no kernel variant for this turbine count");
    const int llg = pick_ll(h->N, B);
    if ((llg >> 4) != h->ll_G || (llg ? (llg & 15) : 1) != h->ll_S) {
      hipFree(h->d_ll_tab); hipFree(h->d_ll_flag); hipFree(h->d_src_log);
      h->d_ll_tab = h->d_src_log = nullptr; h->d_ll_flag = nullptr; h->ll_groups_cap = h->log_slots_cap = 0;
      h->ll_G = llg >> 4; h->ll_S = llg ? (llg & 15) : 1; h->pair_dirty = true;
    }
    if (v != h->variant) {  // the pair table is laid out for the variant's capacity
      hipFree(h->d_pair_tab); hipFree(h->d_pair_first);
      h->d_pair_tab = nullptr; h->d_pair_first = nullptr; h->pair_dirty = true; h->pair_groups_cap = 0;
      h->variant = v;
    }
  }
  if ((size_t)B != h->cap_env) {
    free_batch(h);
    // nothing is committed until every allocation has succeeded: a failure leaves the handle without a batch
    // (wf_set_batch must be called again) instead of with half-allocated buffers under the old sizes
    h->B = 0; h->wind_count = 0; h->shared_dir = false; h->ws_prev_valid = false;
    const size_t bn = (size_t)B * h->N;
    WF_HIP(h, hipMalloc(&h->d_ws, sizeof(double) * B));
    WF_HIP(h, hipMalloc(&h->d_wd, sizeof(double) * B));
    WF_HIP(h, hipMalloc(&h->d_gx, sizeof(double) * bn));
    WF_HIP(h, hipMalloc(&h->d_gy, sizeof(double) * bn));
    WF_HIP(h, hipMalloc(&h->d_flags, sizeof(int) * B));
    WF_HIP(h, hipMalloc(&h->d_farm_tie, sizeof(int) * ((size_t)B + 1)));
    WF_HIP(h, hipMalloc(&h->d_gidx, sizeof(int) * bn));
    h->cap_env = B; h->cap_bn = bn;
  }
  h->B = B; h->wind_count = 0; h->shared_dir = false; h->ws_prev_valid = false;
  h->n_groups = 0; h->grid_step = 0.0;
  return WF_OK;
}

int wf_set_wind_counts(wf_handle* h, const double* ws, int n_ws, const double* wd, int n_wd, int on_device) {
  if (!h || !ws || !wd) return WF_E_INVALID;
  if (h->B <= 0) return fail(h, WF_E_INVALID, "wf_set_batch must be called before wf_set_wind");
  if ((n_ws != 1 && n_ws != h->B) || (n_wd != 1 && n_wd != h->B)) return fail(h, WF_E_INVALID, "wind count must be 1 or env_batch");
  if (n_ws == 1 && n_wd != 1) return fail(h, WF_E_INVALID, "a direction per farm needs a speed per farm");
  WF_ON_DEVICE(h);
  if (!on_device) {
    for (int i = 0; i < n_ws; ++i)
      if (!(ws[i] > 0.0)) return fail(h, WF_E_INVALID, "wind speed must be > 0 and direction finite");
    for (int i = 0; i < n_wd; ++i)
      if (!std::isfinite(wd[i])) return fail(h, WF_E_INVALID, "wind speed must be > 0 and direction finite");
  }
  // One direction for every farm (explicitly: n_wd == 1; or host arrays whose directions are all equal, e.g. sampled
  // speeds under a fixed direction): the rotation, the sort and the pair table depend on the direction only, so this
  // is the shared-wind path with a speed per farm.
  bool same_dir = n_ws > 1 && (n_wd == 1 || !on_device);
  for (int i = 1; same_dir && i < n_wd; ++i) same_dir = wd[i] == wd[0];
  const int count = n_ws;
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  WF_HIP(h, hipMemcpyAsync(h->d_ws, ws, sizeof(double) * n_ws, kind, h->stream));
  WF_HIP(h, hipMemcpyAsync(h->d_wd, wd, sizeof(double) * n_wd, kind, h->stream));
  if (n_wd == 1 && n_ws > 1)  // the step kernel reads a direction per farm next to the speed per farm
    WF_HIP(h, wfk_launch_fill(h->B, h->d_wd, h->stream));
  {
    int rc = run_geometry(h, same_dir ? 1 : count, h->d_wd, !on_device);
    if (rc != WF_OK) return rc;
  }
  if (!on_device) WF_HIP(h, hipStreamSynchronize(h->stream));  // caller's host arrays may go away
  h->shared_dir = same_dir;
  h->wind_sync = !on_device;
  h->wind_count = count;
  h->series_T = 0;
  ungroup(h);
  h->ws_prev_valid = false;
  h->pair_dirty = true;
  return WF_OK;
}

int wf_set_wind(wf_handle* h, const double* ws, const double* wd, int count, int on_device) {
  return wf_set_wind_counts(h, ws, count, wd, count, on_device);
}

int wf_step(wf_handle* h, const float* yaw, float* power, float* wspd, float* wdir, float* load, int on_device) {
  if (!h || !yaw) return WF_E_INVALID;
  if (h->wind_count == 0) return fail(h, WF_E_INVALID, "wf_set_wind must be called before wf_step");
  WF_ON_DEVICE(h);
  if (h->model_dirty) {
    int rc = build_consts(h);
    if (rc != WF_OK) return rc;
  }
  const size_t bn = (size_t)h->B * h->N;
  if (on_device) return launch_step(h, yaw, power, wspd, wdir, load, nullptr);
  if (!h->d_yaw) {
    WF_HIP(h, hipMalloc(&h->d_yaw, sizeof(float) * bn));
    WF_HIP(h, hipHostMalloc(&h->h_yaw, sizeof(float) * bn, hipHostMallocDefault));
  }
  if (!h->d_out) WF_HIP(h, hipMalloc(&h->d_out, sizeof(float) * bn * 7));
  if (!h->h_out) WF_HIP(h, hipHostMalloc(&h->h_out, sizeof(float) * bn * 7, hipHostMallocDefault));
  std::memcpy(h->h_yaw, yaw, sizeof(float) * bn);
  WF_HIP(h, hipMemcpyAsync(h->d_yaw, h->h_yaw, sizeof(float) * bn, hipMemcpyHostToDevice, h->stream));
  {
    int rc = launch_step(h, h->d_yaw, h->d_out, h->d_out + bn, h->d_out + 2 * bn, h->d_out + 3 * bn, nullptr);
    if (rc != WF_OK) return rc;
  }
  WF_HIP(h, hipMemcpyAsync(h->h_out, h->d_out, sizeof(float) * bn * 7, hipMemcpyDeviceToHost, h->stream));
  WF_HIP(h, hipStreamSynchronize(h->stream));
  if (power) std::memcpy(power, h->h_out, sizeof(float) * bn);
  if (wspd) std::memcpy(wspd, h->h_out + bn, sizeof(float) * bn);
  if (wdir) std::memcpy(wdir, h->h_out + 2 * bn, sizeof(float) * bn);
  if (load) std::memcpy(load, h->h_out + 3 * bn, sizeof(float) * bn * 4);
  return WF_OK;
}

int wf_wind_sample(wf_handle* h, unsigned long long seed, const wf_wind_dist* dist) {
  if (!h) return WF_E_INVALID;
  if (h->B <= 0) return fail(h, WF_E_INVALID, "wf_set_batch must be called before wf_wind_sample");
  WF_ON_DEVICE(h);
  const wf_wind_dist def{8.0, 8.0, 3.0, 28.0, 270.0, 20.0, 0.0, 360.0};
  const wf_wind_dist d = dist ? *dist : def;
  if (!(d.ws_scale > 0) || !(d.ws_shape > 0) || !(d.ws_lo > 0) || !(d.ws_lo <= d.ws_hi) || !(d.wd_std >= 0))
    return fail(h, WF_E_INVALID, "invalid wind distribution parameters");
  const double dv[8] = {d.ws_scale, d.ws_shape, d.ws_lo, d.ws_hi, d.wd_mean, d.wd_std, d.wd_lo, d.wd_hi};
  WF_HIP(h, wfk_launch_wind_sample(h->B, seed, dv, h->d_ws, h->d_wd, h->stream));
  {
    int rc = run_geometry(h, h->B, h->d_wd, false);
    if (rc != WF_OK) return rc;
  }
  h->wind_count = h->B;
  h->shared_dir = false;
  h->wind_sync = false;
  h->series_T = 0;
  ungroup(h);
  h->ws_prev_valid = false;
  h->pair_dirty = true;  // env_batch 1: "one wind per farm" is also "one wind for the batch" (table path)
  return WF_OK;
}

int wf_wind_sample_binned(wf_handle* h, unsigned long long seed, const wf_wind_dist* dist, double step_deg) {
  if (!h) return WF_E_INVALID;
  if (h->B <= 0) return fail(h, WF_E_INVALID, "wf_set_batch must be called before wf_wind_sample_binned");
  if (!(step_deg > 0.0) || !(step_deg <= 90.0)) return fail(h, WF_E_INVALID, "direction step must be in (0, 90] degrees");
  const int K = (int)std::llround(360.0 / step_deg);
  if (std::fabs(K * step_deg - 360.0) > 1e-9) return fail(h, WF_E_INVALID, "direction step must divide 360 degrees");
  WF_ON_DEVICE(h);
  const wf_wind_dist def{8.0, 8.0, 3.0, 28.0, 270.0, 20.0, 0.0, 360.0};
  const wf_wind_dist d = dist ? *dist : def;
  if (!(d.ws_scale > 0) || !(d.ws_shape > 0) || !(d.ws_lo > 0) || !(d.ws_lo <= d.ws_hi) || !(d.wd_std >= 0))
    return fail(h, WF_E_INVALID, "invalid wind distribution parameters");
  if (!groups_pay_off(h, K)) {  // too many bins for this batch (or no table path): sample un-binned directions
    int rc = wf_wind_sample(h, seed, dist);
    return rc;
  }
  const double dv[8] = {d.ws_scale, d.ws_shape, d.ws_lo, d.ws_hi, d.wd_mean, d.wd_std, d.wd_lo, d.wd_hi};
  if (!h->d_bins) WF_HIP(h, hipMalloc(&h->d_bins, sizeof(int) * h->B));
  WF_HIP(h, wfk_launch_wind_sample_binned(h->B, seed, dv, step_deg, h->d_ws, h->d_wd, h->d_bins, h->stream));
  std::vector<int> bins(h->B);
  WF_HIP(h, hipMemcpyAsync(bins.data(), h->d_bins, sizeof(int) * h->B, hipMemcpyDeviceToHost, h->stream));
  WF_HIP(h, hipStreamSynchronize(h->stream));
  // geometry and pair tables of the K grid directions depend on layout and model only: built once, kept across resets
  const bool cached = h->grid_step == step_deg && h->n_groups == K && h->series_T == 0;
  if (!cached) {
    hipFree(h->d_group_wd); h->d_group_wd = nullptr;
    WF_HIP(h, hipMalloc(&h->d_group_wd, sizeof(double) * K));
    WF_HIP(h, wfk_launch_bin_centres(K, step_deg, h->d_group_wd, h->stream));
  }
  int rc = build_groups(h, bins.data(), K, h->d_group_wd, !cached);
  if (rc != WF_OK) return rc;
  h->grid_step = step_deg;
  h->wind_sync = true;
  h->wind_count = h->B;
  h->shared_dir = false;
  h->series_T = 0;
  h->ws_prev_valid = false;
  return WF_OK;
}

int wf_wind_series(wf_handle* h, int T, const double* ws, const double* wd, const int* start, unsigned long long seed) {
  if (!h || !ws || !wd) return WF_E_INVALID;
  if (h->B <= 0) return fail(h, WF_E_INVALID, "wf_set_batch must be called before wf_wind_series");
  if (T < 1) return fail(h, WF_E_INVALID, "the wind series needs at least one row");
  for (int i = 0; i < T; ++i)
    if (!(ws[i] > 0.0) || !std::isfinite(wd[i])) return fail(h, WF_E_INVALID, "wind speed must be > 0 and direction finite");
  WF_ON_DEVICE(h);
  WF_HIP(h, hipStreamSynchronize(h->stream));
  hipFree(h->d_series_ws); hipFree(h->d_series_wd); hipFree(h->d_series_start);
  h->d_series_ws = h->d_series_wd = nullptr; h->d_series_start = nullptr;
  WF_HIP(h, hipMalloc(&h->d_series_ws, sizeof(double) * T));
  WF_HIP(h, hipMalloc(&h->d_series_wd, sizeof(double) * T));
  WF_HIP(h, hipMalloc(&h->d_series_start, sizeof(int) * h->B));
  WF_HIP(h, hipMemcpy(h->d_series_ws, ws, sizeof(double) * T, hipMemcpyHostToDevice));
  WF_HIP(h, hipMemcpy(h->d_series_wd, wd, sizeof(double) * T, hipMemcpyHostToDevice));
  if (start) {
    for (int b = 0; b < h->B; ++b)
      if (start[b] < 0 || start[b] >= T) return fail(h, WF_E_INVALID, "series start out of range");
    WF_HIP(h, hipMemcpy(h->d_series_start, start, sizeof(int) * h->B, hipMemcpyHostToDevice));
  } else {
    WF_HIP(h, wfk_launch_series_start(h->B, T, seed, h->d_series_start, h->stream));
  }
  h->series_T = T;
  h->series_t = -1;
  h->ws_prev_valid = false;
  ungroup(h);
  h->grid_step = 0.0;
  // A shared series has only T distinct winds: farms are grouped by their start row (farms with the same start see the
  // same row at every tick), one sorted geometry + pair table per ROW, and the table path serves the whole playback.
  if (groups_pay_off(h, T)) {
    std::vector<int> st(h->B);
    if (start) std::memcpy(st.data(), start, sizeof(int) * h->B);
    else {
      WF_HIP(h, hipMemcpyAsync(st.data(), h->d_series_start, sizeof(int) * h->B, hipMemcpyDeviceToHost, h->stream));
      WF_HIP(h, hipStreamSynchronize(h->stream));
    }
    int rc = build_groups(h, st.data(), T, h->d_series_wd, true);
    if (rc != WF_OK) return rc;
  }
  h->wind_sync = true;
  return wf_wind_series_step(h);
}

int wf_wind_series_step(wf_handle* h) {
  if (!h) return WF_E_INVALID;
  if (h->series_T <= 0) return fail(h, WF_E_INVALID, "wf_wind_series must be called first");
  if (h->series_t + 1 >= h->series_T) return fail(h, WF_E_INVALID, "wind series exhausted");
  WF_ON_DEVICE(h);
  h->series_t += 1;
  if (h->series_t >= 1) {  // keep the wind of the state before this tick for the reward normalisation
    if (!h->d_ws_prev) WF_HIP(h, hipMalloc(&h->d_ws_prev, sizeof(double) * h->B));
    WF_HIP(h, hipMemcpyAsync(h->d_ws_prev, h->d_ws, sizeof(double) * h->B, hipMemcpyDeviceToDevice, h->stream));
    h->ws_prev_valid = true;  // consumed by the next wf_env_step that computes a reward
  }
  WF_HIP(h, wfk_launch_series_gather(h->B, h->series_T, h->series_t, h->d_series_start, h->d_series_ws, h->d_series_wd,
                                     h->d_ws, h->d_wd, h->stream));
  h->wind_count = h->B;
  h->shared_dir = false;
  if (h->n_groups > 0) {
    h->group_shift = h->series_t;  // group g (= start row g) is on row (g + t) % T now: geometry and tables are per row
  } else {
    int rc = run_geometry(h, h->B, h->d_wd, false);
    if (rc != WF_OK) return rc;
    h->pair_dirty = true;  // see wf_wind_sample
  }
  return WF_OK;
}

int wf_get_wind(wf_handle* h, double* ws, double* wd, int on_device) {
  if (!h || !ws || !wd) return WF_E_INVALID;
  if (h->wind_count == 0) return fail(h, WF_E_INVALID, "no wind has been set");
  WF_ON_DEVICE(h);
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (h->wind_count == h->B) {
    WF_HIP(h, hipMemcpyAsync(ws, h->d_ws, sizeof(double) * h->B, kind, h->stream));
    WF_HIP(h, hipMemcpyAsync(wd, h->d_wd, sizeof(double) * h->B, kind, h->stream));
    if (!on_device) WF_HIP(h, hipStreamSynchronize(h->stream));
    return WF_OK;
  }
  double v[2];
  WF_HIP(h, hipMemcpyAsync(&v[0], h->d_ws, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  WF_HIP(h, hipMemcpyAsync(&v[1], h->d_wd, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  WF_HIP(h, hipStreamSynchronize(h->stream));
  std::vector<double> a(h->B, v[0]), b(h->B, v[1]);
  const hipMemcpyKind k2 = on_device ? hipMemcpyHostToDevice : hipMemcpyHostToHost;
  WF_HIP(h, hipMemcpy(ws, a.data(), sizeof(double) * h->B, k2));
  WF_HIP(h, hipMemcpy(wd, b.data(), sizeof(double) * h->B, k2));
  return WF_OK;
}

int wf_env_config(wf_handle* h, const wf_env_params* p) {
  if (!h || !p) return WF_E_INVALID;
  if (!(p->yaw_lo < p->yaw_hi) || !(p->yaw_step > 0) || !(p->actuator_rate > 0) || !(p->dt > 0))
    return fail(h, WF_E_INVALID, "need yaw_lo < yaw_hi, yaw_step > 0, actuator_rate > 0, dt > 0");
  h->env = *p;
  return WF_OK;
}

static int env_alloc(wf_handle* h) {
  if (h->B <= 0) return fail(h, WF_E_INVALID, "wf_set_batch must be called first");
  if (h->d_env_yaw) return WF_OK;
  const size_t bn = (size_t)h->B * h->N;
  WF_HIP(h, hipMalloc(&h->d_env_yaw, sizeof(float) * bn));
  WF_HIP(h, hipMalloc(&h->d_env_acc, sizeof(float) * bn));
  WF_HIP(h, hipMalloc(&h->d_env_moves, sizeof(int) * h->B));
  WF_HIP(h, hipMemsetAsync(h->d_env_yaw, 0, sizeof(float) * bn, h->stream));
  WF_HIP(h, hipMemsetAsync(h->d_env_acc, 0, sizeof(float) * bn, h->stream));
  WF_HIP(h, hipMemsetAsync(h->d_env_moves, 0, sizeof(int) * h->B, h->stream));
  return WF_OK;
}

int wf_env_reset(wf_handle* h) {
  if (!h) return WF_E_INVALID;
  WF_ON_DEVICE(h);
  const bool fresh = h->d_env_yaw == nullptr;
  int rc = env_alloc(h);
  if (rc != WF_OK) return rc;
  if (!fresh) {
    const size_t bn = (size_t)h->B * h->N;
    WF_HIP(h, hipMemsetAsync(h->d_env_yaw, 0, sizeof(float) * bn, h->stream));
    WF_HIP(h, hipMemsetAsync(h->d_env_acc, 0, sizeof(float) * bn, h->stream));
    WF_HIP(h, hipMemsetAsync(h->d_env_moves, 0, sizeof(int) * h->B, h->stream));
  }
  return WF_OK;
}

int wf_env_set_prev_wind(wf_handle* h, const double* ws, int on_device) {
  if (!h || !ws) return WF_E_INVALID;
  if (h->B <= 0) return fail(h, WF_E_INVALID, "wf_set_batch must be called first");
  WF_ON_DEVICE(h);
  if (!h->d_ws_prev) WF_HIP(h, hipMalloc(&h->d_ws_prev, sizeof(double) * h->B));
  WF_HIP(h, hipMemcpyAsync(h->d_ws_prev, ws, sizeof(double) * h->B, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                           h->stream));
  if (!on_device) WF_HIP(h, hipStreamSynchronize(h->stream));
  h->ws_prev_valid = true;
  return WF_OK;
}

int wf_env_state(wf_handle* h, float* yaw, float* acc, int* moves, int set, int on_device) {
  if (!h) return WF_E_INVALID;
  WF_ON_DEVICE(h);
  int rc = env_alloc(h);
  if (rc != WF_OK) return rc;
  const size_t bn = (size_t)h->B * h->N;
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : (set ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost);
  auto xfer = [&](void* user, void* dev, size_t bytes) -> hipError_t {
    if (!user) return hipSuccess;
    return set ? hipMemcpyAsync(dev, user, bytes, kind, h->stream) : hipMemcpyAsync(user, dev, bytes, kind, h->stream);
  };
  WF_HIP(h, xfer(yaw, h->d_env_yaw, sizeof(float) * bn));
  WF_HIP(h, xfer(acc, h->d_env_acc, sizeof(float) * bn));
  WF_HIP(h, xfer(moves, h->d_env_moves, sizeof(int) * h->B));
  if (!on_device) WF_HIP(h, hipStreamSynchronize(h->stream));
  return WF_OK;
}

int wf_env_step(wf_handle* h, const float* action, float* reward, float* yaw, float* power, float* wspd, float* wdir,
                float* load, int on_device) {
  if (!h) return WF_E_INVALID;
  if (h->wind_count == 0) return fail(h, WF_E_INVALID, "wf_set_wind must be called before wf_env_step");
  WF_ON_DEVICE(h);
  int rc = env_alloc(h);
  if (rc != WF_OK) return rc;
  if (h->model_dirty && (rc = build_consts(h)) != WF_OK) return rc;
  const size_t bn = (size_t)h->B * h->N, B = (size_t)h->B;
  WfEnvArgs ea{};
  ea.yaw_state = h->d_env_yaw; ea.acc = h->d_env_acc; ea.moves = h->d_env_moves;
  ea.yaw_step = h->env.yaw_step; ea.yaw_lo = h->env.yaw_lo; ea.yaw_hi = h->env.yaw_hi;
  ea.rate = h->env.actuator_rate; ea.dt = h->env.dt; ea.budget = h->env.budget;
  ea.load_coef = h->env.load_coef; ea.discrete = h->env.discrete;
  // free wind of the state BEFORE the step, when it differs from the current one (series tick, wf_env_set_prev_wind):
  // valid for one reward only — a second env step without a new tick normalises by the current wind again
  ea.ws_prev = (h->ws_prev_valid && h->d_ws_prev) ? h->d_ws_prev : nullptr;
  if (reward) h->ws_prev_valid = false;
  if (on_device) {
    ea.action = action; ea.reward = reward;
    if ((rc = launch_step(h, nullptr, power, wspd, wdir, load, &ea)) != WF_OK) return rc;
    if (yaw) WF_HIP(h, hipMemcpyAsync(yaw, h->d_env_yaw, sizeof(float) * bn, hipMemcpyDeviceToDevice, h->stream));
    return WF_OK;
  }
  if (!h->d_env_act) {
    WF_HIP(h, hipMalloc(&h->d_env_act, sizeof(float) * bn));
    WF_HIP(h, hipMalloc(&h->d_env_out, sizeof(float) * B));
    WF_HIP(h, hipHostMalloc(&h->h_env_act, sizeof(float) * bn, hipHostMallocDefault));
    WF_HIP(h, hipHostMalloc(&h->h_env_out, sizeof(float) * (B + bn), hipHostMallocDefault));
  }
  if (!h->d_out) WF_HIP(h, hipMalloc(&h->d_out, sizeof(float) * bn * 7));
  if (!h->h_out) WF_HIP(h, hipHostMalloc(&h->h_out, sizeof(float) * bn * 7, hipHostMallocDefault));
  if (action) {
    std::memcpy(h->h_env_act, action, sizeof(float) * bn);
    WF_HIP(h, hipMemcpyAsync(h->d_env_act, h->h_env_act, sizeof(float) * bn, hipMemcpyHostToDevice, h->stream));
    ea.action = h->d_env_act;
  }
  ea.reward = reward ? h->d_env_out : nullptr;
  if ((rc = launch_step(h, nullptr, power ? h->d_out : nullptr, wspd ? h->d_out + bn : nullptr,
                        wdir ? h->d_out + 2 * bn : nullptr, load ? h->d_out + 3 * bn : nullptr, &ea)) != WF_OK) return rc;
  if (reward) WF_HIP(h, hipMemcpyAsync(h->h_env_out, h->d_env_out, sizeof(float) * B, hipMemcpyDeviceToHost, h->stream));
  if (yaw) WF_HIP(h, hipMemcpyAsync(h->h_env_out + B, h->d_env_yaw, sizeof(float) * bn, hipMemcpyDeviceToHost, h->stream));
  if (power || wspd || wdir || load)
    WF_HIP(h, hipMemcpyAsync(h->h_out, h->d_out, sizeof(float) * bn * 7, hipMemcpyDeviceToHost, h->stream));
  WF_HIP(h, hipStreamSynchronize(h->stream));
  if (reward) std::memcpy(reward, h->h_env_out, sizeof(float) * B);
  if (yaw) std::memcpy(yaw, h->h_env_out + B, sizeof(float) * bn);
  if (power) std::memcpy(power, h->h_out, sizeof(float) * bn);
  if (wspd) std::memcpy(wspd, h->h_out + bn, sizeof(float) * bn);
  if (wdir) std::memcpy(wdir, h->h_out + 2 * bn, sizeof(float) * bn);
  if (load) std::memcpy(load, h->h_out + 3 * bn, sizeof(float) * bn * 4);
  return WF_OK;
}

int wf_set_risk_guard(wf_handle* h, double rel_band) {
  if (!h) return WF_E_INVALID;
  if (!(rel_band >= 0.0) || !(rel_band < 0.5)) return fail(h, WF_E_INVALID, "risk guard band must be in [0, 0.5)");
  h->guard_rel = rel_band;
  h->consts.guard_inv = rel_band > 0.0 ? (float)(1.0 / rel_band) : 1125899906842624.0f;
  return WF_OK;
}

int wf_set_risk_resolve(wf_handle* h, int mode) {
  if (!h) return WF_E_INVALID;
  if (mode < 0 || mode > 2) return fail(h, WF_E_INVALID, "risk resolve mode must be 0 (off), 1 (flagged farms) or 2 (every farm)");
  h->resolve_mode = mode;
  return WF_OK;
}

int wf_get_resolve_stats(wf_handle* h, int* n_resolved, int* raw_flags, int on_device) {
  if (!h) return WF_E_INVALID;
  if (h->B <= 0) return fail(h, WF_E_INVALID, "wf_set_batch must be called first");
  WF_ON_DEVICE(h);
  if (!h->d_res_list || (h->resolve_mode == 0 && h->model.veer == 0.0)) {  // nothing is being resolved on this batch
    if (n_resolved) *n_resolved = 0;
    if (raw_flags) return wf_get_risk_flags(h, raw_flags, on_device);
    return WF_OK;
  }
  if (raw_flags)
    WF_HIP(h, hipMemcpyAsync(raw_flags, h->d_flags_raw, sizeof(int) * h->B, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                             h->stream));
  if (n_resolved) WF_HIP(h, hipMemcpyAsync(n_resolved, h->d_res_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if (n_resolved || (raw_flags && !on_device)) WF_HIP(h, hipStreamSynchronize(h->stream));
  return WF_OK;
}

int wf_get_risk_flags(wf_handle* h, int* flags, int on_device) {
  if (!h || !flags) return WF_E_INVALID;
  if (h->B <= 0 || !h->d_flags) return fail(h, WF_E_INVALID, "wf_set_batch must be called first");
  WF_ON_DEVICE(h);
  WF_HIP(h, hipMemcpyAsync(flags, h->d_flags, sizeof(int) * h->B, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                           h->stream));
  if (!on_device) WF_HIP(h, hipStreamSynchronize(h->stream));
  return WF_OK;
}

int wf_sync(wf_handle* h) {
  if (!h) return WF_E_INVALID;
  WF_ON_DEVICE(h);
  WF_HIP(h, hipStreamSynchronize(h->stream));
  return WF_OK;
}

int wf_timing_begin(wf_handle* h) {
  if (!h) return WF_E_INVALID;
  WF_ON_DEVICE(h);
  WF_HIP(h, hipEventRecord(h->ev0, h->stream));
  return WF_OK;
}

int wf_timing_end(wf_handle* h, float* ms) {
  if (!h || !ms) return WF_E_INVALID;
  WF_ON_DEVICE(h);
  WF_HIP(h, hipEventRecord(h->ev1, h->stream));
  WF_HIP(h, hipEventSynchronize(h->ev1));
  WF_HIP(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
  return WF_OK;
}

int wf_get_kernel_info(wf_handle* h, wf_kernel_info* info) {
  if (!h || !info) return WF_E_INVALID;
  if (h->variant < 0) return fail(h, WF_E_INVALID, "wf_set_layout must be called first");
  int G, S; const void* fn;
  wfk_variant(h->variant, &G, &S, &fn);
  // the instantiation the next step would launch: pair table (shared wind), general mirror cores, or default
  if (h->model_dirty && h->N > 0) { int rc = build_consts(h); if (rc != WF_OK) return rc; }
  const bool tab = (h->wind_count == 1 || h->shared_dir || h->n_groups > 0) && h->N <= WF_PAIR_MAX_N && wfk_variant_has_table(h->variant) && !h->no_pair_table;
  fn = wfk_variant_fn(h->variant, tab ? (h->wind_count == 1 ? 2 : 3) : (h->consts.mirror_core_n <= 1 ? 0 : 1));
  info->pair_table = tab ? 1 : 0;
  info->direction_groups = h->n_groups;
  hipFuncAttributes a;
  WF_ON_DEVICE(h);
  WF_HIP(h, hipFuncGetAttributes(&a, fn));
  info->lanes_per_env = G; info->slots_per_lane = S;
  const int wpb = tab ? wfk_tab_waves() : 4;
  info->envs_per_block = wpb * (64 / G); info->threads_per_block = 64 * wpb;
  info->grid_blocks = h->n_groups > 0 ? (h->n_slots + info->envs_per_block - 1) / info->envs_per_block
                                      : (h->B > 0 ? (h->B + info->envs_per_block - 1) / info->envs_per_block : 0);
  const bool ll_fly = !tab && h->wind_count == h->B && h->B > 1 && h->n_groups == 0 && h->ll_G && wfk_ll_has_fly(ll_fly_G(h), ll_fly_S(h)) && !h->no_ll_fly;
  info->one_block_kernel = ((tab && h->ll_G) || ll_fly) ? 1 : 0;
  if (info->one_block_kernel) {
    // what serves every wind direction without an x' tie across a block boundary; wf_step_kernel (the variant the
    // fields above would describe) is enqueued behind it for the directions that have one
    const int ll_s = tab ? h->ll_S : ll_fly_S(h), ll_g = tab ? h->ll_G : ll_fly_G(h);
    WF_HIP(h, wfk_ll_func_attributes(ll_g, ll_s, h->wind_count == 1 ? 1 : 0, tab ? 1 : 0, &a));
    info->lanes_per_env = ll_g; info->slots_per_lane = ll_s;
    info->envs_per_block = wfk_ll_farms_per_block(ll_g); info->threads_per_block = 256;
    info->grid_blocks = (int)(((h->n_groups > 0 ? (size_t)h->n_slots : (size_t)h->B) + info->envs_per_block - 1) / info->envs_per_block);
  }
  info->vgprs = a.numRegs;
  info->lds_bytes = (int)a.sharedSizeBytes; info->scratch_bytes = (int)a.localSizeBytes;
  return WF_OK;
}

const char* wf_last_error(wf_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

}  // extern "C"
