// wf_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4) of the batched wind-farm step.
//
// What they replace: the arithmetic behind
//   reference wfcrl/interface.py:564  fi.calculate_wake          (FLORIS 3.5 sequential GCH solve)
//   reference wfcrl/interface.py:623  fi.get_turbine_powers
//   reference wfcrl/interface.py:629-648  local_load_proxies / local_wind_measurements
//   reference wfcrl/interface.py:663-671  update_wind -> fi.reinitialize (rotation + sort)
// following SURVEY.md Appendix A ([A.x] tags below).
//
// Mapping: a VALU-bound pairwise recurrence with an N-stage sequential dependency, not a
// contraction — no MFMA.
//   * one farm instance is owned by a GROUP of G lanes of a 64-wide wavefront (64/G farms per wave); lane `sub`
//     of the group owns S target turbines; register slot p holds turbine  t = (blk + p)*G + sub  while the sources
//     of block blk are being processed;
//   * the per-turbine state (sum of squared deficits: 6, V and W on the 3x3 rotor grid: 9 + 9, TI per grid column: 3
//     = 27 floats) lives in VGPRs for the whole solve; HBM is touched once for yaw in and once for the 7 outputs;
//     yaw (with its sin/cos) and the sorted geometry are staged per wave in LDS;
//   * the upstream->downstream recurrence runs over sources i = 0..N-1: the owner lane's rotor means are broadcast
//     in the group with ds_bpermute (__shfl), every lane derives the source's constants and applies the source to
//     its own targets in two passes (transverse velocities; then deflection + deficit + SOSFS + wake-added TI);
//     when the G sources of a block are done nothing later touches the block: its outputs are written (after the
//     transverse velocities of later sources that tie with it in x') and the slots shift down by one (static register
//     indices, no scratch);
//   * decisions that are discontinuities of the model are taken on the float64 coordinates, as the reference takes
//     them: the order and sign of x'_t - x'_i, the 15 D reach of the wake-added TI (x_t <= x_i + 15 D);
//   * template parameters: <G, S> lane group / slots per lane (chosen by the host from N and the batch size);
//     MC1  compile-time skip of ground-mirror vortex cores that are exactly 1.0f in float32 (on the table path,
//          where no core is evaluated: shared wind speed (true) or a speed per farm (false));
//     TAB  shared-wind-direction path: the geometry-only coefficients of the transverse pass come from a float64
//          pair-coefficient table (wf_pair_table_kernel) whose rows are staged into a double-buffered LDS slab with
//          global_load_lds_dwordx4 (LDS-DMA) one source ahead, one __syncthreads() per source;
//     WPB  waves per block.
//   * optional fused env step (WfEnvArgs): actuation-budget gate, clipped yaw transition and reward in the launch.
// (The ablation / staging-experiment hooks of rounds 1-3 — WF_ABLATE, WF_DIAG_* — are gone from the source; their
// measurements are in DESIGN_HISTORY.md, the builds in the commits named there.)
#include <hip/hip_runtime.h>

#include <cstring>

#include "wf_device.h"

#define WF_TAB_WAVES 4  // waves per block on the pair-table path (two-wave blocks: +8 %, twice the row staging per farm; eight: slower)
// The library is built from this file twice (csrc/Makefile): WF_KSET=1 carries the step-kernel variants with one, two,
// four or five target slots per lane — they have registers to spare at their occupancy — compiled with LLVM's
// iterative-ilp scheduling strategy (-3 % on HornsRev1/65536, -2 % on the B = 1 latency), plus the small kernels
// and the variant dispatcher; WF_KSET=2 carries the register-tight variants (three slots at three waves per SIMD,
// six slots), which spill 40-140 VGPRs under that strategy, with the default scheduler.  WF_KSET=0 (default):
// everything in one translation unit.
#ifndef WF_KSET
#define WF_KSET 0
#endif

#include "wf_kernel_common.h"

#if WF_KSET != 2
// ---------------------------------------------------------------------------------------------
// Geometry: wd % 360, rotation about the layout's bounding-box centre, stable ascending sort [A.1]
// One block per wind condition; float64 throughout.
// ---------------------------------------------------------------------------------------------
// tie_block > 0: also flag the farms whose sorted order has an exact x' tie across a boundary of blocks of tie_block
// turbines (farm_tie[e] = 1, *any_tie = 1): the one-block-at-a-time kernel leaves those to wf_step_kernel.
// Layouts: lx, ly hold one layout of N turbines (layout_mode 0), one per wind condition (1: layout e), or a set indexed
// by layout_of[e] (2) — wf_set_layouts; centre[2 l], centre[2 l + 1] is the centre of rotation of layout l.
__global__ void wf_geometry_kernel(int N, const double* __restrict__ lx, const double* __restrict__ ly,
                                   const double* __restrict__ centre, int layout_mode, const int* __restrict__ layout_of,
                                   const int* __restrict__ layout_n,
                                   const double* __restrict__ wd, int wd_stride, double* __restrict__ gx,
                                   double* __restrict__ gy, int* __restrict__ gidx, int tie_block, int* __restrict__ farm_tie,
                                   int* __restrict__ any_tie) {
  __shared__ double sx[WF_TABLE_PAD * 4];
  __shared__ double sorted_x[WF_TABLE_PAD * 4];
  const int e = blockIdx.x;
  const int t = threadIdx.x;
  const int l = layout_mode == 0 ? 0 : (layout_mode == 1 ? e : layout_of[e]);
  lx += (size_t)l * N; ly += (size_t)l * N;
  const double xc = centre[2 * l], yc = centre[2 * l + 1];
  double w = fmod(wd[(size_t)e * wd_stride], 360.0);
  if (w < 0.0) w += 360.0;
  double dev = fmod(w - 270.0, 360.0);
  if (dev < 0.0) dev += 360.0;
  dev = fmod(dev + 360.0, 360.0);
  const double a = dev * (M_PI / 180.0);
  const double ca = cos(a), sa = sin(a);
  double xr = 0.0, yr = 0.0;
  const int n_l = layout_n ? layout_n[l] : N;  // turbines layout l really has (wf_set_layouts_counts), the rest are placeholders
  if (t < n_l) {
    const double xo = lx[t] - xc, yo = ly[t] - yc;
    xr = xo * ca - yo * sa + xc;
    yr = xo * sa + yo * ca + yc;
    sx[t] = xr;
  }
  __syncthreads();
  if (t >= n_l && t < N) {
    // A placeholder sits 10 000 km and more DOWNSTREAM of every real turbine, each further than the last, whatever the wind
    // direction: last in the sorted order, out of every reach and gate, no x' tie — it receives (negligible) wakes and
    // gives none to a real turbine, so the real turbines' results are those of the unpadded farm.
    double mx = sx[0];
    for (int u = 1; u < n_l; ++u) mx = fmax(mx, sx[u]);
    xr = mx + 1.0e7 * (double)(t - n_l + 1);
    yr = yc;
  }
  __syncthreads();
  if (t >= n_l && t < N) sx[t] = xr;
  __syncthreads();
  if (t < N) {
    int rank = 0;
    for (int u = 0; u < N; ++u) {
      const double xu = sx[u];
      rank += (xu < xr) || (xu == xr && u < t);
    }
    const size_t o = (size_t)e * N + rank;
    gx[o] = xr;
    gy[o] = yr;  // absolute y' in float64: the lateral gate |y_i - y| < 2 D is decided on it [A.3-8]
    gidx[o] = t;
    if (tie_block > 0) sorted_x[rank] = xr;
  }
  if (tie_block > 0) {  // uniform
    if (t == 0) farm_tie[e] = 0;
    __syncthreads();
    if (t + 1 < N && (t + 1) % tie_block == 0 && sorted_x[t] == sorted_x[t + 1]) {
      farm_tie[e] = 1;
      *any_tie = 1;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// On-device wind process
//   reset sampling  ws = clip(scale * Weibull(shape), lo, hi), wd = clip(Normal(mean, std) mod 360, lo, hi)
//                   (reference wfcrl/mdp.py:237-258; NumPy draws there, a counter-based Philox4x32-10 here:
//                   same distributions, different stream)
//   series playback per-farm position in a shared (ws, wd) series, rolled to a per-farm start
//                   (reference wfcrl/interface.py:512-524)
// ---------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0,
                                             unsigned k1) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
  const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
  const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ void philox4x32_10(unsigned long long seed, unsigned long long ctr, unsigned stream,
                                              unsigned out[4]) {
  unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = stream, c3 = 0;
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ double u01(unsigned hi, unsigned lo) {  // (0, 1), 53 bits
  const unsigned long long v = (((unsigned long long)hi << 32) | lo) >> 11;
  return ((double)v + 0.5) * (1.0 / 9007199254740992.0);
}
}  // namespace

__global__ void wf_wind_sample_kernel(int B, unsigned long long seed, double ws_scale, double ws_shape, double ws_lo,
                                      double ws_hi, double wd_mean, double wd_std, double wd_lo, double wd_hi,
                                      double* __restrict__ ws, double* __restrict__ wd) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  unsigned r0[4], r1[4];
  philox4x32_10(seed, (unsigned long long)b, 0u, r0);
  philox4x32_10(seed, (unsigned long long)b, 1u, r1);
  const double e = -log(u01(r0[0], r0[1]));                 // standard exponential
  double s = ws_scale * pow(e, 1.0 / ws_shape);               // Weibull(shape) = E^(1/shape)
  s = fmin(fmax(s, ws_lo), ws_hi);
  const double rad = sqrt(-2.0 * log(u01(r0[2], r0[3])));    // Box-Muller
  double d = wd_mean + wd_std * rad * cos(2.0 * M_PI * u01(r1[0], r1[1]));
  d = fmod(d, 360.0);
  if (d < 0.0) d += 360.0;
  d = fmin(fmax(d, wd_lo), wd_hi);
  ws[b] = s;
  wd[b] = d;
}

// Binned reset sampling: as wf_wind_sample_kernel, with the direction rounded to the nearest point of a grid of
// `step` degrees (bin index = round(wd / step) mod K): K = 360 / step distinct directions in the whole batch.
__global__ void wf_wind_sample_binned_kernel(int B, unsigned long long seed, double ws_scale, double ws_shape, double ws_lo,
                                             double ws_hi, double wd_mean, double wd_std, double wd_lo, double wd_hi,
                                             double step, int K, double* __restrict__ ws, double* __restrict__ wd,
                                             int* __restrict__ bin) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  unsigned r0[4], r1[4];
  philox4x32_10(seed, (unsigned long long)b, 0u, r0);
  philox4x32_10(seed, (unsigned long long)b, 1u, r1);
  const double e = -log(u01(r0[0], r0[1]));
  double s = ws_scale * pow(e, 1.0 / ws_shape);
  s = fmin(fmax(s, ws_lo), ws_hi);
  const double rad = sqrt(-2.0 * log(u01(r0[2], r0[3])));
  double d = wd_mean + wd_std * rad * cos(2.0 * M_PI * u01(r1[0], r1[1]));
  d = fmod(d, 360.0);
  if (d < 0.0) d += 360.0;
  d = fmin(fmax(d, wd_lo), wd_hi);
  int k = (int)llrint(d / step) % K;  // 360 -> bin 0
  ws[b] = s;
  wd[b] = k * step;
  bin[b] = k;
}
__global__ void wf_bin_centres_kernel(int K, double step, double* __restrict__ wd) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K) wd[k] = k * step;
}
__global__ void wf_fill_kernel(int n, double* __restrict__ a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 1 && i < n) a[i] = a[0];
}

__global__ void wf_series_start_kernel(int B, int T, unsigned long long seed, int* __restrict__ start) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  unsigned r[4];
  philox4x32_10(seed, (unsigned long long)b, 2u, r);
  start[b] = (int)(((unsigned long long)r[0] * (unsigned long long)T) >> 32);  // uniform in [0, T)
}

__global__ void wf_series_gather_kernel(int B, int T, int t, const int* __restrict__ start,
                                        const double* __restrict__ s_ws, const double* __restrict__ s_wd,
                                        double* __restrict__ ws, double* __restrict__ wd) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int k = (start[b] + t) % T;
  ws[b] = s_ws[k];
  wd[b] = s_wd[k];
}

extern "C" hipError_t wfk_launch_wind_sample(int B, unsigned long long seed, const double* dist, double* ws, double* wd,
                                             hipStream_t s) {
  hipLaunchKernelGGL(wf_wind_sample_kernel, dim3((B + 255) / 256), dim3(256), 0, s, B, seed, dist[0], dist[1], dist[2],
                     dist[3], dist[4], dist[5], dist[6], dist[7], ws, wd);
  return hipGetLastError();
}
extern "C" hipError_t wfk_launch_wind_sample_binned(int B, unsigned long long seed, const double* dist, double step, double* ws,
                                                    double* wd, int* bin, hipStream_t s) {
  const int K = (int)llround(360.0 / step);
  hipLaunchKernelGGL(wf_wind_sample_binned_kernel, dim3((B + 255) / 256), dim3(256), 0, s, B, seed, dist[0], dist[1], dist[2],
                     dist[3], dist[4], dist[5], dist[6], dist[7], step, K, ws, wd, bin);
  return hipGetLastError();
}
extern "C" hipError_t wfk_launch_bin_centres(int K, double step, double* wd, hipStream_t s) {
  hipLaunchKernelGGL(wf_bin_centres_kernel, dim3((K + 255) / 256), dim3(256), 0, s, K, step, wd);
  return hipGetLastError();
}
extern "C" hipError_t wfk_launch_fill(int n, double* a, hipStream_t s) {
  hipLaunchKernelGGL(wf_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, a);
  return hipGetLastError();
}
extern "C" hipError_t wfk_launch_series_start(int B, int T, unsigned long long seed, int* start, hipStream_t s) {
  hipLaunchKernelGGL(wf_series_start_kernel, dim3((B + 255) / 256), dim3(256), 0, s, B, T, seed, start);
  return hipGetLastError();
}
extern "C" hipError_t wfk_launch_series_gather(int B, int T, int t, const int* start, const double* s_ws,
                                               const double* s_wd, double* ws, double* wd, hipStream_t s) {
  hipLaunchKernelGGL(wf_series_gather_kernel, dim3((B + 255) / 256), dim3(256), 0, s, B, T, t, start, s_ws, s_wd, ws, wd);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Pair-coefficient table for a wind condition shared by the whole batch: everything in the transverse-velocity
// pass [A.3-4] that does not depend on the farm's state.  One thread per (source i, target t); float64.
// ---------------------------------------------------------------------------------------------
__global__ void wf_pair_table_kernel(const WfPairConsts pc, const double* __restrict__ gx, const double* __restrict__ gy,
                                     float* __restrict__ tab, int* __restrict__ first_active) {
  // blockIdx.y = direction group (one table per distinct wind direction; 1 group for a shared wind)
  const int i = blockIdx.x, t = threadIdx.x, grp = blockIdx.y;
  if (t >= pc.NP) return;
  gx += (size_t)grp * pc.N;
  gy += (size_t)grp * pc.N;
  first_active += (size_t)grp * pc.N;
  float* o = tab + ((size_t)grp * pc.N + i) * WF_PAIR_ROW_FLOATS(pc.NP) + (size_t)t * WF_PAIR_STRIDE;
  if (t < pc.N && gx[t] - gx[i] >= 0.0) atomicMin(&first_active[i], t);  // lowest sorted index the source reaches (ties included)
  wf_pair_record(pc, gx, gy, i, t, o);
}

extern "C" hipError_t wfk_launch_pair_table(const WfPairConsts* pc, int n_groups, const double* gx, const double* gy,
                                            float* tab, int* first_active, hipStream_t s) {
  const int threads = ((pc->NP + 63) / 64) * 64;
  hipError_t e = hipMemsetAsync(first_active, 0x7f, sizeof(int) * pc->N * (size_t)n_groups, s);  // "infinity" for atomicMin
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(wf_pair_table_kernel, dim3(pc->N, n_groups), dim3(threads), 0, s, *pc, gx, gy, tab, first_active);
  return hipGetLastError();
}

#endif  // WF_KSET != 2

// ---------------------------------------------------------------------------------------------
// The farm step
// ---------------------------------------------------------------------------------------------
// MC1: only the lowest mirror vortex offset needs its core factor (WfConsts::mirror_core_n <= 1, the case for
// every physical turbine: for the other mirror offsets 1 - Ey*ezm == 1.0f exactly in float32).
// TAB: the transverse-velocity pass reads the shared-wind pair-coefficient table instead of evaluating the
// vortex system per farm (MC1 is then irrelevant).
// WPB: waves per block (4; 8 for the table path so that one staged row serves twice as many farms).
// Blocks per CU the register allocator is asked to make room for (4-wave blocks: = waves per SIMD).  Instruction
// throughput grows ~27 % from two to three waves per SIMD (measured); the <= 168-VGPR budget that needs is within
// reach of the variants with at most three target slots (81 state registers), provided the LDS slab fits thrice.
template <int G, int S, bool TAB, int WPB>
constexpr int min_blocks_per_cu() {
  constexpr size_t lds = sizeof(TableLds) + WPB * sizeof(GeoLds<64 / G, G * S, !TAB>) +
                         (TAB ? 2 * 4 * WF_PAIR_ROW_FLOATS(G * S) + 4 * G * S : 16);
  return ((S <= 3 && 3 * (4 / WPB) * lds <= 160 * 1024) ? 3 : 2) * (4 / WPB);  // WPB is 4, or 2 in staging experiments
}

// VEER: wind_veer != 0 [FLORIS gauss.py rCalt] — on the fly (MC1 = false) and on the pair-table path (the table holds the
// transverse pass only, which veer does not touch): the rotated Gaussian is not even in z - HH, so the SOSFS state holds 9 instead of 6 sums per slot and a column costs one more exp + rcp.
template <int G, int S, bool MC1, bool TAB, int WPB, bool VEER = false>
__global__ __launch_bounds__(64 * WPB, (min_blocks_per_cu<G, S, TAB, WPB>())) void wf_step_kernel(
    const WfConsts c, const WfTables* __restrict__ tab, const double* __restrict__ gx, const double* __restrict__ gy,
    const int* __restrict__ gidx, int geom_stride, const double* __restrict__ ws_in, const double* __restrict__ wd_in,
    int wind_stride, const float* __restrict__ yaw_in, float* __restrict__ o_power, float* __restrict__ o_ws,
    float* __restrict__ o_wd, float* __restrict__ o_load, int B, const WfEnvArgs ea,
    const float* __restrict__ pair_tab, const int* __restrict__ pair_first, const WfGroupArgs ga) {
  constexpr int EPW = 64 / G;  // envs per wave
  constexpr int NP = G * S;    // turbine capacity of this variant
  __shared__ TableLds T;
  __shared__ GeoLds<EPW, NP, !TAB> geo[WPB];
  // shared-wind pair table: the current and the next source's row, filled by LDS-DMA (no VGPR staging)
  constexpr int ROWF = TAB ? WF_PAIR_ROW_FLOATS(NP) : 4;
  __shared__ __attribute__((aligned(16))) float prow[2][ROWF];
  __shared__ int pfirst[TAB ? NP : 1];  // per source: first sorted target index it reaches (dx >= 0, ties included)
  __shared__ unsigned risk_lds[WPB][EPW];  // per farm: WF_RISK_* bits raised during the solve
  // Direction groups (a pair table + sorted geometry per distinct wind direction): all farms of a block belong to one
  // group; blocks beyond the padded farm list carry group -1.
  if (ga.res_zero && blockIdx.x == 0 && threadIdx.x == 0) *ga.res_zero = 0;  // the re-solve counter of the NEXT step (before any early return)
  int grp = 0;
  if (ga.blk_group) {
    grp = ga.blk_group[(blockIdx.x * (WPB * EPW)) / ga.blk_unit];
    if (grp < 0) return;  // whole block, before any barrier
    grp = (grp + ga.shift) % ga.mod;
    if constexpr (TAB) {
      pair_tab += (size_t)grp * c.N * WF_PAIR_ROW_FLOATS(NP);
      pair_first += (size_t)grp * c.N;
    }
  }
  if (ga.pred && !ga.pred[grp]) return;  // this direction is served by wf_step_ll_kernel (whole block, before any barrier)
  if constexpr (TAB) {
    for (int k = threadIdx.x; k < NP; k += blockDim.x) pfirst[k] = (k < c.N) ? pair_first[k] : 0;
    __syncthreads();
  }
  constexpr int row_chunks = TAB ? (WF_PAIR_ROW_FLOATS(NP) / 256) : 0;  // 1-KiB pieces per row
  auto stage_row = [&](int src_i) {
    if constexpr (TAB) {
      const char* g0 = reinterpret_cast<const char*>(pair_tab + (size_t)src_i * WF_PAIR_ROW_FLOATS(NP));
      char* l0 = reinterpret_cast<char*>(&prow[src_i & 1][0]);
      // targets upstream of the source are never read: start at the 1-KiB piece holding its first active target
      const int ch0 = __builtin_amdgcn_readfirstlane(pfirst[src_i]) * (WF_PAIR_STRIDE * 4) / 1024;
      // (hand-issued since round 5 — wf_kernel_common.h: lds_dma16 —: through the builtin the compiler waited for the NEXT row
      // at this step's first LDS read, the thrust-table probe sixty instructions on)
      for (int ch = ch0 + (int)(threadIdx.x >> 6); ch < row_chunks; ch += WPB)
        lds_dma16(g0 + ch * 1024 + (threadIdx.x & 63) * 16, l0 + ch * 1024, false);
    }
  };
  stage_row(0);
  for (int k = threadIdx.x; k < WF_TABLE_PAD; k += blockDim.x) {
    T.knot[k] = tab->knot[k];
    T.ct[k] = tab->ct[k];
    T.cts[k] = tab->ct_slope[k];
    T.pw[k] = tab->pw[k];
    T.pws[k] = tab->pw_slope[k];
  }
  for (int k = threadIdx.x; k < WF_BUCKETS; k += blockDim.x) T.bucket[k] = tab->bucket[k];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane & (G - 1);
  const int gbase = lane & ~(G - 1);
  const int eiw = lane / G;  // env index inside the wave
  int env_raw = (blockIdx.x * (blockDim.x >> 6) + wave) * EPW + eiw + ga.env_base;  // (env_base: a mixed launch's farm range)
  if (ga.perm) env_raw = ga.perm[env_raw];  // padded farm list of the grouped launch: -1 = no farm
  bool env_ok = env_raw >= 0 && env_raw < (ga.env_end ? ga.env_end : B);
  const int env = env_ok ? env_raw : (B - 1);
  if (ga.farm_pred) {  // a wind per farm, behind wf_step_ll_kernel: only the farms it left (x' tie across its blocks)
    const int mine = env_ok ? ga.farm_pred[env] : 0;
    env_ok = env_ok && mine;
    if (!__syncthreads_or(mine)) return;  // nothing to do for this block (the usual case)
  }
  if (sub == 0) risk_lds[wave][eiw] = 0u;
  const int N = c.N;

  // one wind for the whole batch: everything derived from it is wave-uniform and lives in SGPRs (the table path also
  // serves one direction with a speed per farm: instantiation MC1 = false)
  constexpr bool UWS = TAB && MC1;  // table path with a shared wind speed (MC1 is otherwise unused there)
  auto uni = [](float v) { return UWS ? __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))) : v; };
  const float ws = uni((float)ws_in[(size_t)env * wind_stride]);
  const double wd_d = fmod(wd_in[(size_t)env * wind_stride], 360.0);
  const float wd = uni((float)(wd_d < 0.0 ? wd_d + 360.0 : wd_d));
  const float Ui[3] = {uni(ws * c.shearf[0]), uni(ws * c.shearf[1]), uni(ws * c.shearf[2])};
  const float offk = c.off[2] * kGs;
  const float U02c = uni(Ui[0] * Ui[0] * Ui[0] + Ui[2] * Ui[2] * Ui[2]), U1c = uni(Ui[1] * Ui[1] * Ui[1]);
  // sum over the rotor grid of u^3 from the SOSFS sums of a turbine (rows 0 and 2 share their sum without veer)
  auto cube_sum = [&](const float* e) {
    if constexpr (VEER) {
      float m = 0.0f;
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float u = Ui[k] * (1.0f - fsqrt(e[3 * j + k]));
          m = fmaf(u * u, u, m);
        }
      return m;
    } else {
      float fe = 0.0f, fc = 0.0f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float ue = 1.0f - fsqrt(e[2 * j]), uc = 1.0f - fsqrt(e[2 * j + 1]);
        fe = fmaf(ue * ue, ue, fe);
        fc = fmaf(uc * uc, uc, fc);
      }
      return fmaf(U02c, fe, U1c * fc);
    }
  };
  // overlap test "deficit * Uinit_k > threshold" [A.3-8] as a threshold on the deficit itself, with the guard band of
  // the risk flag folded in (see the wake-added-TI block): scale ovs[k] = 1 / (2 g thr_k), offset ovc = 1/2 - 1/(2 g)
  const float ovh = 0.5f * c.guard_inv;  // 1 / (2 g); g = 0 -> 2^49 (no band: f is 0 or 1 except at e == thr exactly)
  const float ovs[3] = {uni(ovh * Ui[0] * c.inv_overlap_thr), uni(ovh * Ui[1] * c.inv_overlap_thr), uni(ovh * Ui[2] * c.inv_overlap_thr)};
  const float ovc = 0.5f - ovh;

  const size_t gofs = ga.blk_group ? (size_t)grp * N : (size_t)env * geom_stride;
  const size_t yofs = (size_t)env * N;
  GeoLds<EPW, NP, !TAB>& L = geo[wave];
  // fused MDP transition: budget gate -> clip increment -> clip setpoint -> accumulate
  const bool env_mode = ea.yaw_state != nullptr;
  int moves_new = 0;
  if (env_mode && ea.action) {
    moves_new = ea.moves[env] + 1;
  }
#pragma unroll
  for (int p = 0; p < S; ++p) {
    const int t = p * G + sub;
    const bool ok = t < N;
    const int tt = ok ? t : 0;
    if constexpr (!TAB) {
      L.x[eiw][t] = ok ? gx[gofs + tt] : -1.0e300;  // padding is never downstream of anything
      const double ygd = gy[gofs + tt];
      L.yd[eiw][t] = ygd;
    }
    const size_t oi = yofs + gidx[gofs + tt];
    float yw;
    if (env_mode) {
      yw = ea.yaw_state[oi];
      if (ea.action) {
        float a = ea.action[oi];
        float acc = ea.acc[oi];
        // actuating_frac = acc / rate / num_moves / dt >= budget  -> action zeroed   (simple_env.py:64-72)
        const float frac = __fdiv_rn(__fdiv_rn(__fdiv_rn(acc, ea.rate), (float)moves_new), ea.dt);
        // the gate zeroes the RAW action (simple_env.py:72) — in the discrete encoding 0 means "down",
        // a quirk of the reference that is kept for parity
        if (frac >= ea.budget) a = 0.0f;
        if (ea.discrete) a = (a - 1.0f) * ea.yaw_step;       // 0/1/2 = down/hold/up   (mdp.py:305-309)
        if (!ea.discrete) a = fminf(fmaxf(a, -ea.yaw_step), ea.yaw_step);
        yw = fminf(fmaxf(yw + a, ea.yaw_lo), ea.yaw_hi);
        acc += fabsf(a);
        if (ok && env_ok) {
          ea.yaw_state[oi] = yw;
          ea.acc[oi] = acc;
          if (ea.yaw_out) ea.yaw_out[oi] = yw;
        }
      }
    } else {
      yw = yaw_in[oi];
    }
    L.yaw[eiw][t] = yw;
    float sy_, cy_;
    sincosf(yw * kDeg2Rad, &sy_, &cy_);
    L.cg[eiw][t] = cy_;
    L.sg[eiw][t] = sy_;
  }
  if (env_mode && ea.action && sub == 0 && env_ok) ea.moves[env] = moves_new;
  if constexpr (TAB) wf_dma_wait();  // (row 0 of the pair table)
  __syncthreads();

  constexpr int NE = VEER ? 9 : 6;
  Slots<S, NE> st;
  // ambient TI exactly as the wake-added-TI candidate evaluates to with no added turbulence (sqrt(0 + amb^2) in
  // device arithmetic): a no-op update then leaves the three grid columns bit-identical (uniform fast path)
  const float amb0 = fsqrt(c.amb2);
#pragma unroll
  for (int p = 0; p < S; ++p) {
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      st.V[p][q] = 0.0f;
      st.W[p][q] = 0.0f;
    }
#pragma unroll
    for (int q = 0; q < NE; ++q) st.esq[p][q] = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) st.TI[p][j] = amb0;
  }

  // ---- transverse velocities of one source on the targets of register slot p [A.3-4] -------------------
  // table path: geometry-only coefficients of the (source, target) pair, one float4 {aV, bV, aW, bW} per grid
  // point, a grid column (three 16-byte reads) at a time
  // Ratio form (round 4, as wf_step_ll_kernel's replay): with rho = Gy / Gwr,  max(Gy aW + Gwr bW, 0) = Gwr max(rho aW + bW, 0)
  // for Gwr > 0 — the wake-rotation circulation gam_wr (a - a^2) ubar is positive — and Gwr min(rho aW + bW, 0) for
  // Gwr < 0 (a rotor wind speed driven negative by an unphysically tight layout: the reference keeps computing): both are
  // Gwr med3(rho aW + bW, 0, copysign(inf, Gwr)) — five instead of six instructions per grid point.  Gwr == 0 (transverse
  // velocities switched off: Gy is 0 too): rho = 0, nothing is added.
  auto apply_tab = [&](int p, const float4* pr, float rho, float Gwr) {
    const float lim = copysignf(__builtin_inff(), Gwr);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float4 cf[3] = {pr[3 * j], pr[3 * j + 1], pr[3 * j + 2]};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int q = j * 3 + k;
        st.V[p][q] = fmaf(Gwr, fmaf(rho, cf[k].x, cf[k].y), st.V[p][q]);
        st.W[p][q] = fmaf(Gwr, __builtin_amdgcn_fmed3f(fmaf(rho, cf[k].z, cf[k].w), 0.0f, lim), st.W[p][q]);  // W[W<0] = 0, quirk (5)
      }
    }
  };
  // on the fly (a wind condition per farm)
  auto apply_fly = [&](int p, float dx, float dy, float Gt, float Gb, float Gwr) {
    float dec[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) dec[k] = frcp(fmaf(c.decay_a[k], dx, 1.0f));
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float yL = dy + c.yoff[j];
      const float yL2 = yL * yL;
      const float Ey = fexp2(-yL2 * c.exp_c);
      // core/r of the 7 distinct real and 7 distinct mirror vortex offsets, accumulated on the fly into
      // the row sums  A_k = sum Gamma z core/r  (-> V)  and  B_k = sum Gamma core/r  (-> W):
      //   class index mi = m + 3;  real:   top k = mi (mi<=2), bottom k = mi-4 (mi>=4), rotation k = mi-2
      //                            mirror: top k = mi-4 (mi>=4), bottom k = mi (mi<=2), rotation k = mi-2
      float A[3] = {0.0f, 0.0f, 0.0f}, Bw[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int mi = 0; mi < 7; ++mi) {
        float tr = fmaf(-Ey, c.ez[mi], 1.0f) * frcp(yL2 + c.zc2[mi]);
        if (mi == 3) {
          // class 0 (zc = num_eps): a target column within ~2 m of the vortex line has r^2 / eps^2 so small that
          // 1 - Ey ez cancels in float32 — exactly behind the source it returns 0 for 5e-9, and on an aligned grid every
          // upstream source leaves 3e-5 of V standing (wd 3e-4 deg, power 3e-4 on 112 turbines: tests/tools/
          // wd_error_probe.py).  There: (1 - exp(-s)) / r^2 = (1 - s / 2) / eps^2 (s < 0.005: the next term is 4e-6 of a
          // coefficient that is itself 1e-4 of its neighbours).  Branch-free: a wave-uniform branch here cost 8 %.
          const float ts = fmaf(yL2 + c.zc2[3], c.m_half_inv_eps4, c.inv_eps2);
          tr = (yL2 < c.yl2_small) ? ts : tr;
        }
        const float pr = c.zc[mi] * tr;
        float tm = frcp(yL2 + c.zm2[mi]);
        if (mi == 0 || !MC1) tm *= fmaf(-Ey, c.ezm[mi], 1.0f);  // compile-time: see mirror_core_n
        const float pm = c.zm[mi] * tm;
        if (mi <= 2) {
          A[mi] = fmaf(Gt, pr, A[mi]);   Bw[mi] = fmaf(Gt, tr, Bw[mi]);      // real top
          A[mi] = fmaf(-Gb, pm, A[mi]);  Bw[mi] = fmaf(-Gb, tm, Bw[mi]);     // mirror bottom
        }
        if (mi >= 4) {
          A[mi - 4] = fmaf(Gb, pr, A[mi - 4]);   Bw[mi - 4] = fmaf(Gb, tr, Bw[mi - 4]);   // real bottom
          A[mi - 4] = fmaf(-Gt, pm, A[mi - 4]);  Bw[mi - 4] = fmaf(-Gt, tm, Bw[mi - 4]);  // mirror top
        }
        if (mi >= 2 && mi <= 4) {
          A[mi - 2] = fmaf(Gwr, pr - pm, A[mi - 2]);  // rotation, real - mirror
          Bw[mi - 2] = fmaf(Gwr, tr - tm, Bw[mi - 2]);
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        st.V[p][j * 3 + k] = fmaf(A[k], dec[k], st.V[p][j * 3 + k]);
        st.W[p][j * 3 + k] += fmaxf(-yL * Bw[k] * dec[k], 0.0f);  // W[W<0] = 0, quirk (5)
      }
    }
  };
  // circulations (over 2 pi) of source i from the sum over its rotor grid of u^3 [A.3-1, A.3-4]
  // enable_transverse_velocities off (c.sw_tv = 0, else 1): the circulations the transverse pass applies (Gt, Gb, Gy,
  // Gwt) are zeroed, so V and W stay exactly zero; the wake-rotation circulation Gwr itself still enters the secondary
  // steering [A.3-2]
  auto circulations = [&](float m3, int i, float& Gt, float& Gb, float& Gwr, float& Gwt, float& Gy, float& ubar, float& ct,
                          float& a, float& gt, float& gb) {
    ubar = fcbrt_pos(m3 * (1.0f / 9.0f));
    const float cg = L.cg[eiw][i], sg = L.sg[eiw][i];
    unsigned trisk;
    ct = table_ct(c, T, ubar, trisk) * cg;
    if (trisk) atomicOr(&risk_lds[wave][eiw], trisk);
    const float sq1 = fsqrt(1.0f - ct * cg);
    a = 0.5f * ct * frcp(1.0f + sq1);  // == 0.5/cg*(1 - sqrt(1 - ct*cg))
    Gwr = c.gam_wr * (a - a * a) * ubar;
    gt = c.gam_top * ws * ct;
    gb = c.gam_bot * ws * ct;
    const float scg = sg * cg * c.sw_tv;  // commanded yaw
    Gt = scg * gt;
    Gb = -scg * gb;
    Gy = scg * ct * ws;  // table path: Gt = gam_top*Gy, Gb = -gam_bot*Gy folded into the coefficients
    Gwt = Gwr * c.sw_tv;
  };

  const int nblk = (N + G - 1) / G;
  float psum = 0.0f, lsum = 0.0f;  // per-lane partial sums for the fused reward
  const int n_real = ga.n_real ? ga.n_real[env] : N;  // turbines the farm really has (padded layouts: WfGroupArgs)
  // Register slot p holds turbine block blk + p: once a block's own sources are done nothing downstream in the
  // recurrence touches it again, so its outputs are written and the slots shift down by one (static indices).
  for (int blk = 0; blk < nblk; ++blk) {
    const int nsrc = min(G, N - blk * G);
    const int live = S - blk;  // slots p >= live hold nothing any more
    for (int li = 0; li < nsrc; ++li) {
      const int src = gbase + li;
      const int i = blk * G + li;
      if (TAB && i + 1 < N) stage_row(i + 1);  // lands in the other buffer while this source is processed
      // ---- A. the source's state (slot 0 of lane `li` of the group) ------------------------
      __builtin_amdgcn_s_setprio(3);  // (the per-source scalar phases are latency chains: ahead of the other wave's passes, wf_kernels_ll.hip)
      float vsum = 0.0f;
#pragma unroll
      for (int q = 0; q < 9; ++q) vsum += st.V[0][q];
      const float m3 = __shfl(cube_sum(st.esq[0]), src);  // sum over the grid of u^3
      const float Vmean = __shfl(vsum, src) * (1.0f / 9.0f);
      float TIs[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) TIs[j] = __shfl(st.TI[0][j], src);
      double x_i = 0.0;
      double yd_i = 0.0;  // lateral offsets from the float64 coordinates, as the pair table has them (the difference of two
                          // float32 y' - yc leaves 6e-7 of the offset standing where symmetric neighbours cancel)
      int first_i = 0;
      if constexpr (TAB) first_i = __builtin_amdgcn_readfirstlane(pfirst[i]);
      if constexpr (!TAB) {
        x_i = L.x[eiw][i];
        yd_i = L.yd[eiw][i];
      }
      const float yaw_i = L.yaw[eiw][i];

      // ---- B. source constants, part 1 [A.3-1 .. A.3-4] ------------------------------------
      float ubar, ct, a, Gwr, Gwt, gt, gb, Gt, Gb, Gy;
      circulations(m3, i, Gt, Gb, Gwr, Gwt, Gy, ubar, ct, a, gt, gb);
      const float cg = L.cg[eiw][i], sg = L.sg[eiw][i];

      // ---- C. pass 1: transverse velocities on every target at or downstream of the source --
      __builtin_amdgcn_s_setprio(0);
      const float rho_tab = (TAB && Gwt != 0.0f) ? Gy * frcp(Gwt) : 0.0f;  // (apply_tab's ratio form)
      float vbar = 0.0f, wbar = 0.0f;  // mean (V,W) of the SOURCE after its own contribution
#pragma unroll
      for (int p = 0; p < S; ++p) {
        if (p > 0 && p >= live) break;
        const int t = (blk + p) * G + sub;
        // Slot 0 holds the source's own block: its lanes sort out upstream / tied / downstream by the sign of dx.
        // Slots p >= 1 hold later blocks of the ascending sort: every real turbine there is at or downstream of
        // the source (dx >= 0), so only the padding beyond N is masked — no dependent read of dx in front of the
        // branch.
        float dx = 0.0f;
        bool act;
        if (p == 0) {
          if constexpr (TAB) dx = (t >= first_i) ? prow[i & 1][t * WF_PAIR_STRIDE + WF_PAIR_DX] : -1.0f;  // un-staged pieces hold stale rows
          else dx = (float)(L.x[eiw][t] - x_i);
          act = dx >= 0.0f;
        } else {
          if constexpr (!TAB) dx = (float)(L.x[eiw][t] - x_i);
          act = t < N;
        }
        if (act) {
         if constexpr (TAB) {
          apply_tab(p, reinterpret_cast<const float4*>(&prow[i & 1][t * WF_PAIR_STRIDE]), rho_tab, Gwt);
         } else {
          apply_fly(p, dx, (float)(L.yd[eiw][t] - yd_i), Gt, Gb, Gwt);
         }
        }
        if (p == 0) {
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            vbar += st.V[0][q];
            wbar += st.W[0][q];
          }
        }
      }
      vbar = __shfl(vbar, src) * (1.0f / 9.0f);
      wbar = __shfl(wbar, src) * (1.0f / 9.0f);

      // ---- B2. steering + deflection constants [A.3-2, A.3-3] (kept out of pass 1's live range) ----
      __builtin_amdgcn_s_setprio(3);
      // secondary steering
      // (c.sw_steer is 1, or 0 with enable_secondary_steering off: the deflection model then sees the commanded yaw)
      float val = c.sw_steer * (Vmean - Gwr * c.ks_core) * frcp(gt * c.ks_top - gb * c.ks_bot);
      val = fminf(fmaxf(val, -1.0f), 1.0f);
      const float asv = __any(fabsf(val) > 0.3f) ? asinf(val) : asin_small(val);
      const float gd = -(yaw_i * kDeg2Rad + 0.5f * asv);  // radians, deflection sign convention
      // cos(gd) = cos(yaw + h), h = asin(val)/2: half-angle identities instead of a second libm call
      const float c2h = fsqrt(fmaxf(fmaf(-val, val, 1.0f), 0.0f));  // cos(2h) >= 0
      const float ch = fsqrt(0.5f * (1.0f + c2h));                   // cos h >= 0.707
      const float sh = 0.5f * val * frcp(ch);                        // sin h = sin(2h) / (2 cos h)
      const float cgd = fmaf(cg, ch, -sg * sh);
      const float s_cc = fsqrt(1.0f - ct * cgd), s_c = fsqrt(1.0f - ct);
      const float om_scc = ct * cgd * frcp(1.0f + s_cc);  // 1 - s_cc
      const float om_sc = ct * frcp(1.0f + s_c);          // 1 - s_c  (== C0)
      SrcConsts sc;
      sc.sM = fsqrt(ct);  // sqrt(M0), M0 = C0(2-C0) = ct
      const float E0 = fmaf(om_sc, om_sc, fmaf(-c.e0c1, om_sc, c.e0c2));
      sc.sz0d = 0.5f * c.D * fsqrt((1.0f + s_cc) * frcp(2.0f * (1.0f + s_c)));
      sc.sy0d = sc.sz0d * cgd;
      if constexpr (VEER) sc.sy0d *= c.cos_veer;
      const float th0 = c.dm03 * gd * frcp(cgd) * om_scc;
      {
        // |th0| < 0.3 for every admissible yaw: odd polynomial (rel. err < 1e-8 there).  Beyond 0.35 rad (not
        // reachable with |yaw| <= 45 deg) the hardware sin/cos ratio takes over: no libm range reduction inline.
        const float t2 = th0 * th0;
        const float poly = th0 * fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, 0.0088632355f, 0.0218694885f), 0.0539682540f),
                                                          0.1333333333f), 0.3333333333f), 1.0f);
        sc.tan_th0 = poly;
        if (__any(fabsf(th0) > 0.35f)) {  // wave-uniform: never taken for admissible yaw
          const float rev = th0 * 0.15915494309189535f;  // v_sin/v_cos take revolutions
          const float hw = __builtin_amdgcn_sinf(rev) * frcp(__builtin_amdgcn_cosf(rev));
          sc.tan_th0 = (fabsf(th0) > 0.35f) ? hw : poly;
        }
      }
      sc.inv_s0d = frcp(sc.sy0d * sc.sz0d);
      const float pfac = th0 * E0 * (1.0f / 5.2f) * fsqrt(sc.sy0d * sc.sz0d * frcp(ct)) * kLn2;  // * log2(arg)/ky
      sc.lnA = 1.6f + sc.sM;
      sc.lnB = 1.6f - sc.sM;
      const float x0num_d = c.D * cgd * (1.0f + s_cc) * (1.0f / 1.41421356237f);
      // The three grid columns of a source almost always carry the same TI (they differ only when the
      // |dy| < 2D gate of A.3-8 split a rotor): then sigma, C and the far-wake deflection are evaluated
      // once per target instead of once per column, from column 0's constants.  Wave-uniform choice,
      // identical results.
      const bool uni = __all((TIs[0] == TIs[1]) && (TIs[1] == TIs[2]));
      // ---- D. yaw-added recovery [A.3-5] and deficit constants [A.3-6] -----------------------
      const float I0 = TIs[0];
      const float uI = ubar * I0;
      const float mix2 = (vbar * vbar + wbar * wbar) * (1.0f / 3.0f);
      const float inv_ubar = frcp(ubar);
      const float Itot = fsqrt(fmaf(uI, uI, mix2)) * inv_ubar;
      const float Imix = mix2 * inv_ubar * inv_ubar * frcp(Itot + I0);  // == Itot - I0, no cancellation
      const float dTI = c.gch_gain * Imix;  // gch_gain is 0 with enable_yaw_added_recovery off
      if (lane == src) {
#pragma unroll
        // (FLORIS ends every source step with TI = maximum(sqrt(ti_added^2 + ambient^2), TI) over ALL turbines — for the source's
        // own turbine that is max(ambient, TI + dTI): a no-op unless its rotor-mean speed is NEGATIVE (an unphysically tight farm
        // behind a thrust table clipped at 0.9999), where I_tot / ubar and with it dTI turn negative; the deficit pass below
        // still sees TI + dTI, as the reference's does.  Round-5 fuzz, case 5041/1720: TI -0.565 against 0.04.)
        for (int j = 0; j < 3; ++j) st.TI[0][j] = fmaxf(st.TI[0][j] + dTI, amb0);
      }
      sc.sy0v = c.sz0v * cg;
      if constexpr (VEER) sc.sy0v *= c.cos_veer;
      sc.snw = c.near_c * fsqrt(0.5f * ct);
      sc.kdef = ct * cg * c.kdef;
      const float ch_pref = c.ch_c * fexp2(c.ch_ai * flog2(a));
      const float x0num_v = c.D * cg * (1.0f + s_c) * (1.0f / 1.41421356237f);
      // Only column 0's constants are kept across the target loop; in the rare split-TI case the other two columns'
      // are re-derived per target from (TIs[j], dTI) — 14 registers less at the pressure peak of the hot path.
      const float b2om = c.beta2 * om_sc, b2om_d = c.beta2_d * om_sc;
      auto col_consts = [&](float ti_pre, float ti_post) {
        ColConsts k;
        k.x0d = x0num_d * frcp(fmaf(c.alpha4_d, ti_pre, b2om_d));
        k.kyd = fmaf(c.ka_d, ti_pre, c.kb_d);
        k.d0 = sc.tan_th0 * k.x0d;
        k.pj = pfac * frcp(k.kyd);
        k.x0v = x0num_v * frcp(fmaf(c.alpha4, ti_post, b2om));
        k.ix0v = frcp(k.x0v);
        k.kyv = fmaf(c.ka, ti_post, c.kb);
        return k;
      };
      const ColConsts k0 = col_consts(TIs[0], TIs[0] + dTI);

      // ---- E. pass 2: deflection, deficit, SOSFS, wake-added turbulence ----------------------
      __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int p = 0; p < S; ++p) {
        if (p > 0 && p >= live) break;
        const int t = (blk + p) * G + sub;
        float dx;
        float4 ex = {0.0f, 0.0f, 0.0f, 0.0f};  // {dx, dy, tipow, decision bits}
        // Discontinuities of the pair are decided on the float64 coordinates, in FLORIS' own form (table path: by
        // wf_pair_table_kernel, stored in the record): within reach of the wake-added TI  x_t <= x_i + 15 D  (tipow > 0),
        // velocity deficit on  x_t > x_i + 0.1  (bit 3), lateral gate of grid column j  |y_i - (y_t + off_j)| < 2 D
        // (bits 0-2; on the fly they are evaluated inside the wake-added-TI block, where they are needed).
        bool in15;
        int bits;
        if constexpr (TAB) {
          ex = *reinterpret_cast<const float4*>(&prow[i & 1][t * WF_PAIR_STRIDE + WF_PAIR_DX]);
          dx = (p > 0 || t >= first_i) ? ex.x : -1.0f;  // un-staged pieces hold stale rows (slot 0 only)
          in15 = ex.z > 0.0f;
          bits = __float_as_int(ex.w);
        } else {
          const double xt = L.x[eiw][t];
          dx = (float)(xt - x_i);
          in15 = xt <= x_i + c.fifteenD_d;
          bits = (xt > x_i + 0.1) ? 8 : 0;
        }
        // slots p >= 1: all real turbines have dx >= 0 (see pass 1), and at 0 <= dx <= 0.1 (ties) everything below is
        // an exact no-op: amp_on = 0 zeroes the deficits and the TI candidate is the ambient value
        const bool act = (p == 0) ? (dx > 0.0f) : (t < N);
        if (act) {
          float dy;
          if constexpr (TAB) dy = ex.y;
          else dy = (float)(L.yd[eiw][t] - yd_i);
          const float lin = fmaf(c.bd, dx, c.ad);
          const float amp_on = (bits & 8) ? 1.0f : 0.0f;
          float e1[3], e0[3];
          float e2[VEER ? 3 : 1];  // with veer: the deficit of row k = 2 (e0 is row 0)
          if constexpr (VEER) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
              column_deficit_veer(c, sc, (j == 0 || uni) ? k0 : col_consts(TIs[j], TIs[j] + dTI), dx, dy + c.off[j], lin, amp_on,
                                  e0[j], e1[j], e2[j]);
          } else if (uni) {
            // column-independent part once
            const float xs = fmaxf(dx - k0.x0d, 0.0f);
            const float syd = fmaf(k0.kyd, xs, sc.sy0d), szd = fmaf(k0.kyd, xs, sc.sz0d);
            const float s = fsqrt(syd * szd * sc.inv_s0d);
            const float arg = sc.lnA * fmaf(1.6f, s, -sc.sM) * frcp(sc.lnB * fmaf(1.6f, s, sc.sM));
            const float d_far = fmaf(k0.pj, flog2(arg), k0.d0);
            const float delta = ((dx > k0.x0d) ? d_far : dx * sc.tan_th0) + lin;
            const bool far = dx >= k0.x0v;
            const float up = dx * k0.ix0v;
            const float xf = dx - k0.x0v;
            const float sy = far ? fmaf(k0.kyv, xf, sc.sy0v) : fmaf(up, sc.sy0v - sc.snw, sc.snw);
            const float sz = far ? fmaf(k0.kyv, xf, c.sz0v) : fmaf(up, c.sz0v - sc.snw, sc.snw);
            const float isy = frcp(sy), isz = frcp(sz);
            const float xarg = sc.kdef * isy * isz;
            const float C = (xarg >= 1.0f) ? 1.0f : xarg * frcp(1.0f + fsqrt(fmaxf(1.0f - xarg, 0.0f)));
            // exp(-t^2/2) = exp2(-(t*kGs)^2); the grid offsets are (-D/4, 0, +D/4) in y and in z
            const float zz = offk * isz;
            const float ez = fexp2(-(zz * zz));
            const float amp = amp_on * C;
            const float y1 = (dy - delta) * isy * kGs, oy = offk * isy;
            const float yy[3] = {y1 - oy, y1, y1 + oy};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              e1[j] = amp * fexp2(-(yy[j] * yy[j]));
              e0[j] = e1[j] * ez;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 3; ++j)
              column_deficit(c, sc, j == 0 ? k0 : col_consts(TIs[j], TIs[j] + dTI), dx, dy + c.off[j], lin, amp_on, e1[j], e0[j]);
          }
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            if constexpr (VEER) {
              st.esq[p][3 * j] = fmaf(e0[j], e0[j], st.esq[p][3 * j]);
              st.esq[p][3 * j + 1] = fmaf(e1[j], e1[j], st.esq[p][3 * j + 1]);
              st.esq[p][3 * j + 2] = fmaf(e2[j], e2[j], st.esq[p][3 * j + 2]);
            } else {
              st.esq[p][2 * j] = fmaf(e0[j], e0[j], st.esq[p][2 * j]);
              st.esq[p][2 * j + 1] = fmaf(e1[j], e1[j], st.esq[p][2 * j + 1]);
            }
          }
          // Wake-added TI reaches a target only within 15 D downstream and 2 D laterally [A.3-8]; elsewhere the
          // candidate is the ambient value, which never exceeds the running maximum: skipped when no lane needs it.
          if constexpr (TAB) {
            if (!__any(in15 && (bits & 7))) continue;
          } else {
            if (!__any(in15 && (fabsf(dy) < c.twoD + c.off[2] + 1.0f))) continue;  // float32 prefilter with a margin
            // grid offsets -D/4, 0, +D/4 and 2 D = 8 (D/4): exact in float64 from the one constant D/4
            const double yt = L.yd[eiw][t], yi_d = L.yd[eiw][i], twoD_d = 8.0 * c.q_d;
            bits |= (fabs(yi_d - (yt - c.q_d)) < twoD_d) ? 1 : 0;
            bits |= (fabs(yi_d - yt) < twoD_d) ? 2 : 0;
            bits |= (fabs(yi_d - (yt + c.q_d)) < twoD_d) ? 4 : 0;
          }
          // overlap count: grid points with deficit * Uinit_k > threshold; `near` tracks how close any of them comes to
          // the threshold (this is the one state-dependent discontinuity of the model: float32 cannot reproduce the
          // float64 decision inside a band of rounding width around it — such farms are flagged, WF_RISK_OVERLAP)
          // One fma with the clamp output modifier per grid point:  f = clamp(0.5 + (e - thr_k) / (2 g thr_k))  is exactly 0
          // below the guard band |e / thr_k - 1| < g, exactly 1 above it and fractional inside (no VCC round trips).
          // A point in the band shows as mantissa / low exponent bits in the OR of the nine bit patterns (0 and 1.0f =
          // 0x3f800000 contribute none): the farm is flagged (WF_RISK_OVERLAP).  The count is the sum rounded to the
          // nearest integer: exact whenever no point is in the band, and with one point in it that point counts iff
          // f > 1/2, i.e. e > thr_k.
          float cnt = 0.0f;
          unsigned fbits = 0u;
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const float f0 = __builtin_amdgcn_fmed3f(fmaf(e0[j], ovs[0], ovc), 0.0f, 1.0f);
            const float f1 = __builtin_amdgcn_fmed3f(fmaf(e1[j], ovs[1], ovc), 0.0f, 1.0f);
            const float f2 = __builtin_amdgcn_fmed3f(fmaf((VEER ? e2[VEER ? j : 0] : e0[j]), ovs[2], ovc), 0.0f, 1.0f);
            cnt += (f0 + f1) + f2;
            fbits |= __float_as_uint(f0) | __float_as_uint(f1) | __float_as_uint(f2);
          }
          cnt = rintf(cnt);
          if ((fbits & 0xc07fffffu) && in15 && (bits & 7)) atomicOr(&risk_lds[wave][eiw], (unsigned)WF_RISK_OVERLAP);
          // Crespo-Hernandez with overlap gating [A.3-8]
          float tipow;
          if constexpr (TAB) {
            tipow = ex.z;
          } else {
            const float dxp = (dx > 0.1f) ? dx : dx + 1.0f;
            tipow = fexp2(c.ch_down * flog2(dxp * c.invD));
          }
          const float ti = ch_pref * tipow;
          const float tia = in15 ? ti * (cnt * (1.0f / 9.0f)) : 0.0f;
          const float cand = fsqrt(fmaf(tia, tia, c.amb2));
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            // TI and cand are non-negative: the maximum is taken on the bit patterns (no NaN canonicalisation)
            const float cm = (bits & (1 << j)) ? cand : 0.0f;
            st.TI[p][j] = __uint_as_float(max(__float_as_uint(st.TI[p][j]), __float_as_uint(cm)));
          }
        }
      }
      if constexpr (TAB) {
        wf_dma_wait();    // this wave's pieces of the next row have landed ...
        __syncthreads();  // ... and so have everyone else's; everyone is done with the current one
      }
    }  // li

    // ---- x' ties across the block boundary ------------------------------------------------------
    // dx = 0 counts as downstream in the transverse pass [A.3-4], in both directions of a tie.  Sources of later
    // blocks that tie with turbines of this block therefore still owe them their transverse velocities (they change
    // nothing but these turbines' wind-direction / std v / std w outputs: the turbines have acted already).  Such a
    // source's circulations depend on its wake sum only, which is final (everything between the tied turbines is
    // tied as well, and the deficit pass is a no-op at dx = 0), so they are evaluated ahead of the source's turn.
    // Exact ties are what axis-aligned grid layouts have at wd = 270.
    if (blk + 1 < nblk) {
      const int t0 = blk * G + sub;
      for (int k = 0; (blk + 1) * G + k < N; ++k) {
        const int i2 = (blk + 1) * G + k;
        const int p2 = 1 + k / G, l2 = k & (G - 1);  // register slot and lane (in the group) of that source
        float dx0;
        bool tied;
        if constexpr (TAB) {
          tied = __builtin_amdgcn_readfirstlane(pfirst[i2]) < (blk + 1) * G;
          dx0 = 0.0f;
        } else {
          dx0 = (float)(L.x[eiw][t0] - L.x[eiw][i2]);  // <= 0 in the ascending sort; padding is -inf
          tied = __any(dx0 >= 0.0f);
        }
        if (!tied) break;  // ties are contiguous in the sort
        float ee[NE];  // the SOSFS sums of that source's slot
#pragma unroll
        for (int q = 0; q < NE; ++q) {
          ee[q] = st.esq[S > 1 ? 1 : 0][q];
#pragma unroll
          for (int p = 2; p < S; ++p) ee[q] = (p2 == p) ? st.esq[p][q] : ee[q];
        }
        const float m3 = __shfl(cube_sum(ee), gbase + l2);
        float ubar, ct, a, Gwr, Gwt, gt, gb, Gt, Gb, Gy;
        circulations(m3, i2, Gt, Gb, Gwr, Gwt, Gy, ubar, ct, a, gt, gb);
        if constexpr (TAB) {
          // the source's table row is not staged yet: its (source, target) record comes straight from L2
          const float* rec = pair_tab + (size_t)i2 * WF_PAIR_ROW_FLOATS(NP) + (size_t)t0 * WF_PAIR_STRIDE;
          if (rec[WF_PAIR_DX] >= 0.0f) apply_tab(0, reinterpret_cast<const float4*>(rec), Gwt != 0.0f ? Gy * frcp(Gwt) : 0.0f, Gwt);
        } else {
          if (dx0 >= 0.0f) apply_fly(0, dx0, (float)(L.yd[eiw][t0] - L.yd[eiw][i2]), Gt, Gb, Gwt);
        }
      }
    }

    // ---- outputs [A.4] of block blk (slot 0) ---------------------------------------------------
    {
      const int t = blk * G + sub;
      if (t < N) {
        const int o = gidx[gofs + t];
        float U[9], m3 = 0.0f, mu = 0.0f, mv = 0.0f, mw = 0.0f, adir = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if constexpr (VEER) {
#pragma unroll
            for (int k = 0; k < 3; ++k) U[3 * j + k] = Ui[k] * (1.0f - fsqrt(st.esq[0][3 * j + k]));
          } else {
            const float ue = 1.0f - fsqrt(st.esq[0][2 * j]), uc = 1.0f - fsqrt(st.esq[0][2 * j + 1]);
            U[3 * j] = Ui[0] * ue; U[3 * j + 1] = Ui[1] * uc; U[3 * j + 2] = Ui[2] * ue;
          }
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          m3 = fmaf(U[q] * U[q], U[q], m3);
          mu += U[q]; mv += st.V[0][q]; mw += st.W[0][q];
        }
        if (o_wd) {
          // mean of atan2(V, U) over the rotor grid: U > 0 and |V| << U except in unphysical layouts
          bool small = true;
          float rr[9];
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            rr[q] = st.V[0][q] * frcp(U[q]);
            small = small && (U[q] > 0.0f) && (fabsf(rr[q]) <= 0.25f);
          }
          if (__all(small)) {
#pragma unroll
            for (int q = 0; q < 9; ++q) adir += atan_small(rr[q]);
          } else {
#pragma unroll
            for (int q = 0; q < 9; ++q) adir += atan2f(st.V[0][q], U[q]);
          }
        }
        mu *= (1.0f / 9.0f); mv *= (1.0f / 9.0f); mw *= (1.0f / 9.0f);
        float su = 0.0f, sv = 0.0f, sw = 0.0f;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          const float du = U[q] - mu, dv = st.V[0][q] - mv, dw = st.W[0][q] - mw;
          su = fmaf(du, du, su); sv = fmaf(dv, dv, sv); sw = fmaf(dw, dw, sw);
        }
        const float wsp = fcbrt_pos(m3 * (1.0f / 9.0f));
        {  // a rotor-grid speed that is not positive (an unphysically tight farm: summed deficits beyond 1; the reference keeps
           // computing): the cube mean of mixed-sign speeds cancels, float32 keeps 1e-5 of it at best
          float umin = U[0];
#pragma unroll
          for (int q = 1; q < 9; ++q) umin = fminf(umin, U[q]);
          if (!(umin > 0.0f)) atomicOr(&risk_lds[wave][eiw], (unsigned)WF_RISK_NEGATIVE_SPEED);
        }
        const float cy = L.cg[eiw][t];
        const float veff = c.dens_f * wsp * fexp2(c.pw * flog2(cy));
        float pslope;
        const float pwr = c.rho * table_pw(c, T, veff, pslope);
        // knees of the power curve (just above cut-in, the cut-out drop): where the relative condition number
        // v |P'| / max(P, 1 kW) exceeds knee_kappa, a wind-speed error of float32 size (~3e-6 relative after the
        // recurrence) is amplified past the power tolerance
        if (c.rho * fabsf(pslope) * veff > c.knee_kappa * fmaxf(pwr, 1.0e3f))
          atomicOr(&risk_lds[wave][eiw], (unsigned)WF_RISK_POWER_KNEE);
        float4 l;
        l.x = (st.TI[0][0] + st.TI[0][1] + st.TI[0][2]) * (1.0f / 3.0f);
        l.y = fsqrt(su * (1.0f / 9.0f));
        l.z = fsqrt(sv * (1.0f / 9.0f));
        l.w = fsqrt(sw * (1.0f / 9.0f));
        const bool real = o < n_real;  // (a placeholder of a padded layout: zeros out, nothing into the reward)
        psum += real ? pwr : 0.0f;
        lsum += real ? (l.x + l.y) + (l.z + l.w) : 0.0f;  // loads are non-negative: |.| is the identity
        if (env_ok) {
          const size_t oo = yofs + o;
          if (o_power) o_power[oo] = real ? (ea.power_mw ? pwr * 1.0e-6f : pwr) : 0.0f;
          if (o_ws) o_ws[oo] = real ? wsp : 0.0f;
          if (o_wd) o_wd[oo] = real ? wd - adir * (kRad2Deg / 9.0f) : 0.0f;
          if (o_load) reinterpret_cast<float4*>(o_load)[oo] = real ? l : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
      }
    }
    // shift the register slots down: the next block of sources sits in slot 0
#pragma unroll
    for (int p = 0; p + 1 < S; ++p) {
#pragma unroll
      for (int q = 0; q < 9; ++q) { st.V[p][q] = st.V[p + 1][q]; st.W[p][q] = st.W[p + 1][q]; }
#pragma unroll
      for (int q = 0; q < NE; ++q) st.esq[p][q] = st.esq[p + 1][q];
#pragma unroll
      for (int j = 0; j < 3; ++j) st.TI[p][j] = st.TI[p + 1][j];
    }
  }  // blk

  if (ga.risk_flags && sub == 0 && env_ok) {
    const int rf = (int)risk_lds[wave][eiw];
    ga.risk_flags[env] = rf;
    if (ga.res_list) {  // the float64 re-solve's work list (wf_device.h: WfGroupArgs)
      ga.flags_raw[env] = rf;
      if (rf & ga.res_mask) ga.res_list[atomicAdd(ga.res_count, 1)] = env;
    }
  }
  if (ea.reward) {
    // r = mean_j(P_j[MW] * 1e3 / ws^3) - load_coef * mean|loads|      (simple_env.py:78-84)
#pragma unroll
    for (int w = G / 2; w >= 1; w >>= 1) {
      psum += __shfl_xor(psum, w);
      lsum += __shfl_xor(lsum, w);
    }
    if (sub == 0 && env_ok) {
      const float invN = __fdiv_rn(1.0f, (float)n_real);
      const float wr = ea.ws_prev ? (float)ea.ws_prev[env] : ws;  // normalised by the PREVIOUS state's free wind
      const float r = psum * invN * 1.0e-3f * frcp(wr * wr * wr) - ea.load_coef * lsum * invN * 0.25f;
      ea.reward[env] = r;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Launch table
// ---------------------------------------------------------------------------------------------
struct WfVariant {
  int G, S;
  const void* fn;      // MC1 = true
  const void* fn_all;  // MC1 = false (general mirror cores)
  const void* fn_tab;     // shared-wind pair table
  const void* fn_tab_ws;  // pair table of a shared wind direction, a wind speed per farm
  const void* fn_veer;    // wind_veer != 0: on the fly, general mirror cores
  const void* fn_veer_tab, *fn_veer_tab_ws;  // ... on the pair-table path (the table holds the transverse pass only: no veer in it)
};

// The table path is instantiated only where two blocks per CU still fit in the 160 KiB LDS and N <= WF_PAIR_MAX_N.
constexpr int kTabWaves = WF_TAB_WAVES;  // waves per block on the table path
template <int G, int S>
constexpr bool tab_fits() {
  return G * S <= WF_PAIR_MAX_N &&
         (8 / kTabWaves) * (sizeof(TableLds) + kTabWaves * sizeof(GeoLds<64 / G, G * S, false>) +
                            2 * 4 * WF_PAIR_ROW_FLOATS(G * S) + 4 * G * S) <= 160 * 1024;
}
// On the table path the MC1 parameter (meaningless there: no vortex core is evaluated) selects whether the wind
// speed is shared as well (true: its derived constants live in SGPRs) or given per farm (false).
template <int G, int S, bool SHARED_SPEED, bool VEER = false>
const void* tab_kernel() {
  if constexpr (tab_fits<G, S>()) return (const void*)&wf_step_kernel<G, S, SHARED_SPEED, true, kTabWaves, VEER>;
  else return nullptr;
}
#define WF_VARIANT(G_, S_)                                                                              \
  {G_, S_, (const void*)&wf_step_kernel<G_, S_, true, false, 4>, (const void*)&wf_step_kernel<G_, S_, false, false, 4>, \
   tab_kernel<G_, S_, true>(), tab_kernel<G_, S_, false>(), (const void*)&wf_step_kernel<G_, S_, false, false, 4, true>,     \
   tab_kernel<G_, S_, true, true>(), tab_kernel<G_, S_, false, true>()}
static const WfVariant kVariants[] = {
#if WF_KSET != 2
    WF_VARIANT(4, 4),  WF_VARIANT(8, 4),  WF_VARIANT(16, 4), WF_VARIANT(16, 5), WF_VARIANT(32, 4), WF_VARIANT(64, 4),
    WF_VARIANT(4, 1),  WF_VARIANT(4, 2),  WF_VARIANT(8, 1),  WF_VARIANT(8, 2),  WF_VARIANT(16, 1), WF_VARIANT(16, 2),
    WF_VARIANT(32, 1), WF_VARIANT(32, 2), WF_VARIANT(64, 1), WF_VARIANT(64, 2),
#endif
#if WF_KSET != 1
    WF_VARIANT(4, 3),  WF_VARIANT(8, 3),  WF_VARIANT(16, 3), WF_VARIANT(16, 6), WF_VARIANT(32, 3), WF_VARIANT(64, 3),
#endif
};
constexpr int kNumLocal = (int)(sizeof(kVariants) / sizeof(kVariants[0]));

// variant i of THIS translation unit
static void local_variant(int i, int* G, int* S, const void** fn) {
  *G = kVariants[i].G;
  *S = kVariants[i].S;
  *fn = kVariants[i].fn;
}
// kind 0: MC1 on-the-fly, 1: general mirror cores, 2: shared-wind pair table, 3: pair table with a speed per farm;
// 4, 5, 6: wind veer on the fly / with kinds 2 / 3
static const void* local_variant_fn(int i, int kind) {
  if (kind == 4) return kVariants[i].fn_veer;
  if (kind == 5) return kVariants[i].fn_veer_tab;
  if (kind == 6) return kVariants[i].fn_veer_tab_ws;
  return kind == 3 ? kVariants[i].fn_tab_ws : (kind == 2 ? kVariants[i].fn_tab : (kind == 1 ? kVariants[i].fn_all : kVariants[i].fn));
}
static hipError_t local_launch_step(int variant, const WfConsts* c, const WfTables* tab, const double* gx, const double* gy,
                                    const int* gidx, int geom_stride, const double* ws, const double* wd, int wind_stride,
                                    const float* yaw, float* power, float* o_ws, float* o_wd, float* load, int B,
                                    const WfEnvArgs* env, const float* pair_tab, const int* pair_first,
                                    const WfGroupArgs* grp, hipStream_t s, int* grid_out) {
  const WfVariant& v = kVariants[variant];
  const bool use_tab = pair_tab && v.fn_tab;
  const int wpb = use_tab ? kTabWaves : 4;
  const int envs_per_block = wpb * (64 / v.G);
  WfGroupArgs ga = *grp;
  // grouped launch: the farm list is padded per group to whole blocks (ga.n_blocks of them, some possibly unused)
  const int n_farms = ga.env_end ? ga.env_end - ga.env_base : B;  // (a mixed launch serves a range of the batch)
  const int grid = ga.blk_group ? (ga.n_slots + envs_per_block - 1) / envs_per_block : (n_farms + envs_per_block - 1) / envs_per_block;
  if (grid_out) *grid_out = grid;
  WfConsts cc = *c;
  WfEnvArgs ea;
  if (env) ea = *env; else memset(&ea, 0, sizeof(ea));
  void* args[] = {&cc, &tab, &gx, &gy, &gidx, &geom_stride, &ws, &wd, &wind_stride, &yaw, &power, &o_ws, &o_wd, &load, &B, &ea, &pair_tab, &pair_first, &ga};
  const void* fn = use_tab ? (cc.veer_on ? (wind_stride == 0 ? v.fn_veer_tab : v.fn_veer_tab_ws) : (wind_stride == 0 ? v.fn_tab : v.fn_tab_ws))
                           : (cc.veer_on ? v.fn_veer : (cc.mirror_core_n <= 1 ? v.fn : v.fn_all));
  if (!use_tab) pair_tab = nullptr;
  return hipLaunchKernel(fn, dim3(grid), dim3(64 * wpb), args, 0, s);
}

#define WF_STEP_ARGS                                                                                                  \
  int variant, const WfConsts *c, const WfTables *tab, const double *gx, const double *gy, const int *gidx,          \
      int geom_stride, const double *ws, const double *wd, int wind_stride, const float *yaw, float *power,          \
      float *o_ws, float *o_wd, float *load, int B, const WfEnvArgs *env, const float *pair_tab,                     \
      const int *pair_first, const WfGroupArgs *grp, hipStream_t s, int *grid_out
#define WF_STEP_PASS(v_)                                                                                              \
  v_, c, tab, gx, gy, gidx, geom_stride, ws, wd, wind_stride, yaw, power, o_ws, o_wd, load, B, env, pair_tab,         \
      pair_first, grp, s, grid_out

#if WF_KSET == 2
// second translation unit: its variants are reached through the dispatcher of the first
extern "C" int wfk2_num_variants() { return kNumLocal; }
extern "C" void wfk2_variant(int i, int* G, int* S, const void** fn) { local_variant(i, G, S, fn); }
extern "C" int wfk2_variant_has_table(int i) { return kVariants[i].fn_tab != nullptr; }
extern "C" const void* wfk2_variant_fn(int i, int kind) { return local_variant_fn(i, kind); }
extern "C" hipError_t wfk2_launch_step(WF_STEP_ARGS) { return local_launch_step(WF_STEP_PASS(variant)); }
#else
#if WF_KSET == 1
extern "C" int wfk2_num_variants();
extern "C" void wfk2_variant(int i, int* G, int* S, const void** fn);
extern "C" int wfk2_variant_has_table(int i);
extern "C" const void* wfk2_variant_fn(int i, int kind);
extern "C" hipError_t wfk2_launch_step(WF_STEP_ARGS);
#else
static int wfk2_num_variants() { return 0; }
static void wfk2_variant(int, int*, int*, const void**) {}
static int wfk2_variant_has_table(int) { return 0; }
static const void* wfk2_variant_fn(int, int) { return nullptr; }
static hipError_t wfk2_launch_step(WF_STEP_ARGS) { return hipErrorInvalidValue; }
#endif
// variant index space of the library: [0, kNumLocal) here, then the second translation unit's
extern "C" int wfk_num_variants() { return kNumLocal + wfk2_num_variants(); }
extern "C" void wfk_variant(int i, int* G, int* S, const void** fn) {
  if (i < kNumLocal) local_variant(i, G, S, fn); else wfk2_variant(i - kNumLocal, G, S, fn);
}
extern "C" int wfk_variant_has_table(int i) {
  return i < kNumLocal ? (kVariants[i].fn_tab != nullptr) : wfk2_variant_has_table(i - kNumLocal);
}
extern "C" int wfk_tab_waves() { return kTabWaves; }
extern "C" const void* wfk_variant_fn(int i, int kind) {
  return i < kNumLocal ? local_variant_fn(i, kind) : wfk2_variant_fn(i - kNumLocal, kind);
}

extern "C" hipError_t wfk_launch_geometry(int n_env, int N, const double* lx, const double* ly, const double* centre,
                                          int layout_mode, const int* layout_of, const int* layout_n, const double* wd, int wd_stride, double* gx,
                                          double* gy, int* gidx, int tie_block, int* farm_tie, int* any_tie, hipStream_t s) {
  const int threads = ((N + 63) / 64) * 64;
  if (tie_block > 0) {
    hipError_t e = hipMemsetAsync(any_tie, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(wf_geometry_kernel, dim3(n_env), dim3(threads), 0, s, N, lx, ly, centre, layout_mode, layout_of, layout_n, wd,
                     wd_stride, gx, gy, gidx, tie_block, farm_tie, any_tie);
  return hipGetLastError();
}

extern "C" hipError_t wfk_launch_step(WF_STEP_ARGS) {
  return variant < kNumLocal ? local_launch_step(WF_STEP_PASS(variant)) : wfk2_launch_step(WF_STEP_PASS(variant - kNumLocal));
}
#endif
