// wf_handle.h — the handle behind the C ABI (include/wfstep.h) and what the host-side translation units share:
//   wf_abi.hip       handle life cycle, layout / batch, wf_step, flags, timing
//   wf_model.hip     turbine tables, model constants (wf_default_model, wf_turbine_table, wf_set_model, build_consts)
//   wf_dispatch.hip  which kernel serves a handle (per-handle choice, rounds model), pair tables, the launch, introspection
//   wf_groups.hip    direction groups (series rows, binned reset directions)
//   wf_wind_abi.hip  wf_set_wind*, on-device wind process
//   wf_env_abi.hip   fused env step
// Replaces the FLORIS object the reference holds in FlorisInterface (reference wfcrl/interface.py:479
// `tools.FlorisInterface(simul_file)`) by a handle that owns device-resident geometry, model constants and staging
// buffers.  No CPU fallback: without a HIP device wf_create fails with WF_E_NODEVICE.
#pragma once
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
                                         int n_cu, hipStream_t s);
extern "C" int wfk_num_variants();
extern "C" void wfk_variant(int i, int* G, int* S, const void** fn);
extern "C" int wfk_variant_has_table(int i);
extern "C" int wfk_tab_waves();
extern "C" const void* wfk_variant_fn(int i, int kind);
extern "C" hipError_t wfk_launch_geometry(int n_env, int N, const double* lx, const double* ly, const double* centre,
                                          int layout_mode, const int* layout_of, const int* layout_n, const double* wd, int wd_stride, double* gx,
                                          double* gy, int* gidx, int tie_block, int* farm_tie, int* any_tie, hipStream_t s);
extern "C" int wfk_ll_has_fly(int G, int S);
extern "C" int wfk_ll_has_veer(int G, int S, int table);
extern "C" hipError_t wfk_launch_step_ll_fly(int G, int S, const WfConsts* c, const WfTables* tab, const int* gidx, const double* gx,
                                             const double* gy, const double* ws, const double* wd, const float* yaw,
                                             float* power, float* o_ws, float* o_wd, float* load, int B, const WfEnvArgs* env,
                                             const int* farm_tie, float* src_log, size_t log_records,
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
                                         const int* cross_tie, float* src_log, size_t log_records,
                                         const WfGroupArgs* grp, hipStream_t s);
extern "C" hipError_t wfk_ll_func_attributes(int G, int S, int shared_speed, int table, int veer, int occ2, hipFuncAttributes* a);
extern "C" hipError_t wfk_launch_fill(int n, double* a, hipStream_t s);  // a[1..n) = a[0]
extern "C" hipError_t wfk_sort_tmp_bytes(int B, size_t* bytes);
extern "C" hipError_t wfk_sort_by_direction(int B, int n_slots, const double* wd, float* keys, int* vals, void* tmp, size_t tmp_bytes,
                                            int* perm, hipStream_t s);
extern "C" hipError_t wfk_launch_wind_sample_binned(int B, unsigned long long seed, const double* dist, double step, double* ws,
                                                    double* wd, int* bin, hipStream_t s);
extern "C" hipError_t wfk_launch_bin_centres(int K, double step, double* wd, hipStream_t s);
extern "C" hipError_t wfk_launch_series_start(int B, int T, unsigned long long seed, int* start, hipStream_t s);
extern "C" hipError_t wfk_launch_series_gather(int B, int T, int t, const int* start, const double* s_ws,
                                               const double* s_wd, double* ws, double* wd, hipStream_t s);

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
  std::vector<double> lx, ly;  // [n_layouts][N]
  double xc = 0, yc = 0;       // centre of rotation of layout 0
  // wf_set_layouts: n_layouts > 1 layouts in one batch, layout_of[b] the layout of farm b (empty: farm b has layout b)
  int n_layouts = 1;
  std::vector<int> layout_of;
  int* d_layout_of = nullptr;
  double* d_centre = nullptr;  // [n_layouts][2] centres of rotation
  // wf_set_layouts_counts: layouts of fewer than N turbines, padded with placeholders (wf_geometry_kernel puts them far
  // downstream of the real ones): turbines per layout, and per farm (null: every farm has all N)
  std::vector<int> layout_n;
  int* d_layout_n = nullptr;   // [n_layouts]
  int* d_nreal = nullptr;      // [B]
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
  double guard_rel = 1.0e-5;                // relative half-width of the overlap-threshold guard band.  Measured (tests/tools/
                                            // band_study.py, the float64 device kernel as the checker; round 4: 1.47 M farms, round 6:
                                            // 4.3 M — HornsRev1 / 2, TCRWP, Ablaincourt under the reset distribution, over all speeds
                                            // and directions, under one shared wind; random fuzzer layouts up to 128 turbines,
                                            // profiles/r06_band_study.txt): no unflagged farm leaves the tolerances down to a band
                                            // of 2e-6; the first does at 1e-6 (1 of 524 288 TCRWP farms).  Round 6: 1e-5 up to 128
                                            // turbines — 5 x the narrowest clean band, half the farms of rounds 4-5's 2e-5 to solve
                                            // again in float64 — and 2e-5 beyond (float32 rounding is amplified along the deep rows
                                            // of a 256-turbine grid, tests/parity.py: LARGE_FARM_FACTOR); wf_set_risk_guard overrides
  bool guard_user = false;                  // wf_set_risk_guard was called: wf_set_layout leaves the band alone
  float *d_yaw = nullptr, *d_out = nullptr;  // staging for host callers: yaw [B*N], out [B*N*7]
  float *h_yaw = nullptr, *h_out = nullptr;  // pinned
  size_t cap_env = 0, cap_bn = 0;
  // fused env state
  wf_env_params env{-40.f, 40.f, 5.f, 0.3f, 60.f, 0.1f, 0.1f, 0};
  int env_power_mw = 0;  // wf_env_set_power_unit: the power output of wf_env_step in MW
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
  float* d_src_log = nullptr;  // R = launch slots x padded N records: [R][2] hot, [R][12] cold, [R][4] side (wf_device.h)
  size_t ll_groups_cap = 0, log_records_cap = 0;
  int* d_farm_tie = nullptr;   // [B] + 1: per-farm cross-block-tie flag of the per-farm geometry, then the "any" flag
  // a wind per farm: launch slots of the on-the-fly one-block kernel in ascending wind direction (wf_sort.hip), so that the
  // farms of a wave nearly share their geometry and the kernel's wave-uniform skips take
  int* d_dir_perm = nullptr;   // [dir_slots] farm per launch slot, -1 = padding
  float* d_sort_keys = nullptr;  // [2 B]
  int* d_sort_vals = nullptr;    // [B]
  void* d_sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0, dir_perm_cap = 0;
  int dir_slots = 0;           // 0 = no direction order (identity)
  int geo_tie_block = 0;       // turbines per block the per-farm tie flags / launch order of the last geometry pass were laid out for
  int farm_ties = 2;           // a wind per farm: 0 no farm has such a tie, 1 some have, 2 not read back
  bool wind_sync = true;       // the wind was set by a call that synchronises anyway (host arrays, series, binned sampling)
  int ll_ties = 2;             // cross-block ties of the current directions: 0 none, 1 all of them, 2 some / not read back
  // which kernels may serve this handle (wf_set_kernel_choice; the WF_* environment variables only seed it at wf_create)
  wf_kernel_choice choice{0, 0, -1, 0, 0, -1, -1, -1, -1, -1};
  int n_cu = 256;              // compute units of the handle's device (hipDeviceProp_t::multiProcessorCount)
  // per-handle calibration of the kernel family (wf_dispatch.hip: calibrate_families): whether the families have been
  // timed (or taken from the process cache / wf_set_calibration) for the current configuration
  bool calib_done = false;
  int calib_code = -1;         // what the timing chose: (G << 4) | S, 0 = the register-slot kernel, -1 = never ran
  int mix_main = 0;            // mixed launch of the ungrouped shared-wind table path: farms [0, mix_main) on the one-block family, the rest
                               // on wf_step_kernel (0: one launch) — wf_dispatch.hip: mix_candidate
  bool tab_slot = false;       // the calibration kept the register-slot kernel for the UNGROUPED shared-wind table path; ll_G / ll_S
                               // keep the rounds model's shape, which grouped launches and the on-the-fly path go on using
  float calib_ms[8] = {};      // ms per launch of each family of the rounds model it timed (0 = not timed)
  // ... and of the on-the-fly path (a wind per farm; calibrate_fly): the one-block kernel the table path's family stands
  // for against the register-slot kernel
  int fly_calib = 0;           // 0 = not timed yet (the one-block kernel runs), 1 = the one-block kernel, 2 = wf_step_kernel
  float fly_calib_ms[2] = {};  // {one-block kernel + wf_step_kernel for its tied farms, wf_step_kernel alone}
  // float64 re-solve of the farms the float32 kernels flag (wf_resolve.hip)
  int resolve_mode = 1;        // 0 off, 1 flagged farms (the default: the reference computes every step in float64), 2 every farm
                               // (wf_set_risk_resolve; wind_veer models are served by the VEER float32 instantiations and take
                               // the same modes — nothing forces mode 2)
  WfResolveConsts rconsts{};
  double* d_tab64 = nullptr;   // [3][WF_TABLE_PAD] wind speed, Ct, power in float64
  int *d_res_list = nullptr, *d_res_count = nullptr, *d_flags_raw = nullptr;  // [B], [2] (used alternately: res_parity) + [1] (shadow of h_res_seen), [B]
  int* h_res_seen = nullptr;   // pinned host int: the length of the flagged list as the float64 kernel last found it
  int res_parity = 0;          // which of the two counters the last step with a re-solve used
  bool res_last = false;       // the last step had a re-solve behind it (its list, counter and raw flags are current)
  int res_mask = 0;            // nonzero only while launch_step enqueues the real launch: WF_RISK_* bits that put a farm on the list
  // several turbine definitions per farm (wf_set_turbine_types): every farm is solved by the float64 kernels of
  // wf_resolve_mt.hip, whatever resolve_mode says
  struct TurbineType {
    std::vector<double> ws, ct, cp;
    double tsr, pP, gen_eff, ref_density;
  };
  std::vector<TurbineType> types;  // empty: the one definition of `model`
  std::vector<int> type_of;        // [N] definition of each turbine, caller's order
  double* d_tab64_mt = nullptr;    // [n_types][3][WF_TABLE_PAD]
  double* d_type_consts = nullptr; // [n_types][1 + WF_TYPE_CONSTS]
  int* d_type_of = nullptr;        // [N]
};

namespace wfi {

extern thread_local std::string g_create_error;

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
inline int fail(wf_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return code;
}
#define WF_HIP(h, call)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) return fail(h, WF_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

// wf_abi.hip
void free_batch(wf_handle* h);
// wf_model.hip
void init_default_table();
int build_consts(wf_handle* h);
// wf_dispatch.hip
int pick_variant(const wf_handle* h, int N, int B);
int pick_ll(const wf_handle* h, int N, int B);  // (G << 4) | S, 0 = keep wf_step_kernel
int model_mix(const wf_handle* h, int code, int N, int B);
int repick_ll_slots(const wf_handle* h, int N, int ll_G, int ll_S, long farm_slots);
void set_ll_shape(wf_handle* h, int G, int S);
void apply_kernel_pick(wf_handle* h, int N, int B, bool* variant_changed);
int pair_table(wf_handle* h, const float** out);
int run_geometry(wf_handle* h, int n_env, const double* d_wd, bool sync_ok);
int ll_fly_S(const wf_handle* h);
int ll_fly_G(const wf_handle* h);
size_t ll_npad(const wf_handle* h);
int ll_log_fpb(const wf_handle* h);
int launch_step(wf_handle* h, const float* yaw, float* power, float* wspd, float* wdir, float* load, const WfEnvArgs* ea);
void reset_calibration(wf_handle* h);
int calibrate_now(wf_handle* h, const float* yaw, float* power, float* wspd, float* wdir, float* load, const WfEnvArgs* probe, bool use_cache);
int apply_saved_calibration(wf_handle* h, int code, int fly_choice);
// wf_groups.hip
void ungroup(wf_handle* h);
int farms_per_block(const wf_handle* h);
int group_pad(const wf_handle* h);
int group_unit(const wf_handle* h);
bool groups_pay_off(const wf_handle* h, int K);
bool groups_fit(const wf_handle* h, int K);
double ll_estimate(const wf_handle* h, int fi, int N, long farms);  // wf_dispatch.hip: ms for `farms` farm slots on family fi
int build_groups(wf_handle* h, const int* group_of_farm, int K, const double* d_wd_groups, bool rebuild_geometry);

}  // namespace wfi
