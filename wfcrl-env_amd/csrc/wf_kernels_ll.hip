// wf_kernels_ll.hip — the farm step, one target block at a time ("left-looking"): the kernel the headline runs on.
//
// Same model, same arithmetic and the same discontinuity handling as wf_step_kernel (wf_kernels.hip; references and
// [A.x] tags there); what differs is the order in which the (source, target) pairs of the triangular recurrence are
// visited, and with it what has to stay in registers:
//   * wf_step_kernel ("right-looking") applies a source to ALL downstream targets at once: the state of S target slots
//     (27 S floats) lives in VGPRs for the whole solve, which pins the S >= 4 variants at two waves per SIMD;
//   * here ONE target block of G S turbines (27 S floats per lane, S = 1 or 2) is in registers at a time.  Block J is first
//     swept by all sources of the earlier blocks, REPLAYED from a per-wave source log in device memory; then the block's own
//     sources run the sequential recurrence exactly as wf_step_kernel's slot 0 does (source phase, transverse pass, yaw-added
//     recovery, deficit pass) and append to the log; then the block's outputs are written.  The lane group can be narrow —
//     G = 2: thirty-two farms per wave share every per-source instruction — because the block, not the farm, has to fit the
//     registers.  Families: 2x2 (the headline: 4-turbine blocks, two waves per SIMD), 4x2, 4x1, 8x1, 16x1 (three waves per
//     SIMD with one slot); wf_dispatch.hip picks by batch size (rounds model + the handle's own timing).
//   * The source log is split by who reads it (round 4; struct SrcLog below has the field list):
//       HOT   8 B per source and farm  {Gy / Gwt, Gwt}: all the transverse pass needs — read by EVERY later block; sources
//             2m and 2m + 1 share one float4;
//       COLD  48 B per source and farm (three float4): what deflection / deficit / turbulence need — read only by the
//             blocks the source's wake can reach;
//       BOUND 16 B per source in wave-private LDS {k6, b6, n6}: a bound on 6.12 sigma_y(dx) + D/4 + |deflection| over the
//             farms of the wave (far_bound arithmetic in own_source); beyond it the nearest rotor-grid column gets exp2(-27)
//             of the amplitude — no effect on any float32 result (tests/test_hip_parity.py: skip on / off bit-identical);
//       SIDE  16 B, written and read only for split-TI sources (the three column TIs and dTI).
//     Both global parts are wave-major ([wave][source][farm]): a wave-instruction reads one contiguous run and a block's
//     records are whole 128-byte lines (no line is shared between a block being written and one being read).
//   * Replay of a staged chunk of 64 / (G S) sources (table path): one lane per (source, target) pair tests the pair against
//     the source's bound — the ballot, folded to a bit per source, is the chunk's NEAR LIST (farm-independent: the farms of a
//     wave share geometry); then the transverse pass over ALL sources of the chunk, two per iteration (5 instructions per grid
//     point, the next pair's hot record fetched one iteration ahead); then the deflection / deficit / turbulence pass over
//     the near sources only, the next near source's cold record fetched a step ahead into the other of two register sets.
//     The two passes touch disjoint state and keep their own source order: bit for bit the result of one loop.
//   * pair-table records are laid out per target block, [J][source i][target of J], so that the 64 records of a chunk
//     (64 / (G S) consecutive sources of one block) are one contiguous 11-KiB piece, staged into a double-buffered LDS slab
//     with global_load_lds_dwordx4 one chunk ahead (one __syncthreads() per chunk).  The transfers are issued by hand
//     (lds_dma16, wf_kernel_common.h: inline assembly the compiler's wait-count bookkeeping does not see), so nothing is waited for
//     at the first LDS read of an iteration any more (rounds 2-4: the builtin made every iteration start with s_waitcnt vmcnt(0));
//     the kernel waits where the data is needed — wf_dma_wait() in front of the chunk's closing barrier and at a block's first chunk.
//   * A wind per farm (TAB = false): no pair table — the transverse pass is evaluated on the fly from the farm's own float64
//     coordinates (apply_fly), same log, one loop; wf_set_wind sorts the launch slots by direction so that the farms of a
//     wave lie within a fraction of a degree and the wave-uniform skips take.
//   * the price of the order is the log traffic: 56 B written per source and farm, the hot part re-read once per later
//     block, the cold part once per later block in reach (2.4 GB per launch at HornsRev1 x 65536 = 14 x the algorithmic
//     bytes; it does not fit the L2 — DESIGN.md section 4).
// Precondition: no x' tie across a block boundary (a later block's source at dx = 0 from an earlier block's target owes
// that target its transverse velocities [A.3-4], which this order cannot deliver).  wf_pair_table_ll_kernel detects it
// per wind direction and raises a device flag; this kernel then leaves the launch to wf_step_kernel, which is always
// enqueued behind it with the opposite predicate (no host round trip).  Exact ties are what axis-aligned grid layouts
// have at wd = 270; HornsRev1/2 have none.
#include <hip/hip_runtime.h>

#include <cstring>
#include <type_traits>
#include <utility>

#include "wf_kernel_common.h"

namespace {

// f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}): slot loops whose index is a
// compile-time constant inside generic lambdas (register arrays must be indexed statically)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// sin / cos of the commanded yaw from the angle in radians: |yaw| <= 45 deg for every admissible command, where the
// Taylor polynomials below are exact to < 3e-9; larger angles (never produced by the env) take libm, wave-uniformly.
__device__ __forceinline__ void sincos_yaw(float x, float& s, float& c) {
  const float x2 = x * x;
  s = x * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 2.7557319e-6f, -1.9841270e-4f), 8.3333333e-3f), -1.6666667e-1f), 1.0f);
  c = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, -2.7557319e-7f, 2.4801587e-5f), -1.3888889e-3f), 4.1666667e-2f), -0.5f), 1.0f);
  if (__any(fabsf(x) > 0.8f)) {
    float s2, c2;
    sincosf(x, &s2, &c2);
    s = (fabsf(x) > 0.8f) ? s2 : s;
    c = (fabsf(x) > 0.8f) ? c2 : c;
  }
}

// max over the wave of a value that the G lanes of a farm hold alike (G a power of two), wave-uniform: v_max_f32 with DPP
// operands — quad swaps, half-row and row mirrors, then the row broadcasts that leave the total in lane 63 — six
// instructions and no LDS traffic (a __shfl_xor butterfly is five ds_bpermute with their address arithmetic)
template <int G>
__device__ __forceinline__ float wave_max(float v) {
  auto dpp = [](float x, auto ctrl, auto rows) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), decltype(ctrl)::value, decltype(rows)::value, 0xf, false));
  };
  using I = std::integral_constant<int, 0>;
  (void)sizeof(I);
  if constexpr (G < 2) v = fmaxf(v, dpp(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{}));   // quad_perm [1,0,3,2]
  if constexpr (G < 4) v = fmaxf(v, dpp(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{}));   // quad_perm [2,3,0,1]
  if constexpr (G < 8) v = fmaxf(v, dpp(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{}));  // row_half_mirror
  if constexpr (G < 16) v = fmaxf(v, dpp(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{})); // row_mirror
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{}));                       // row_bcast:15 into rows 1, 3
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{}));                       // row_bcast:31 into rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// What a source leaves behind for the target blocks after its own, in registers (SrcLog) and in the per-wave source log:
//   HOT  (8 bytes per farm):   {Gy / Gwt, Gwt} — the two circulations, all the transverse pass needs; read by EVERY later block;
//   COLD (48 bytes per farm):  three float4 {sy0d, sz0d, sM, tan_th0} {sy0v, x0d, kyd, pj} {x0v, kyv, +-ch_pref, 1 / x0v} —
//        what the deflection / deficit / turbulence pass needs; read only by the later blocks the source's wake can reach
//        (the far-source test below).  Whatever follows from these by one or two instructions (1.6 +- sM, 1 / (sy0d sz0d),
//        tan_th0 x0d, the near-wake sigma 0.501 D sqrt(ct / 2), the amplitude factor ct cos(yaw) D^2 / 8) is re-derived by
//        the reader; every stored float is read by it — a dead component of a load that is still in flight gets reused as
//        a scratch register, which makes the compiler wait for the load (the prefetch of the next record) on the spot.
//        The sign of ch_pref (> 0 by construction) carries the split-TI flag; the three column TIs and dTI such a source
//        needs live in a side array (WfLogSide).
// Both parts are wave-major — [wave][source i][farm of the wave] float2, [wave][source i][q = 0..2][farm] float4 — so that
// a lane's load of its farm's piece is part of one contiguous 64 / G x 8 or x 16 byte run per wave-instruction (the lanes
// of a farm ask for the same address).  A block's G S records are whole 128-byte lines in either part (S x 512 and
// S x 3072 bytes), so no line is shared between a block that is still being written and one that is being read.
//   BOUND (wave-private LDS, 16 bytes per source): {k6, b6, n6} with, over the farms of the wave,
//        6.12 sigma_y(dx) + D/4 + |deflection| - |ad + bd dx|  <=  max(k6 dx + b6, n6)   for every dx >= 0
//        — see far_bound() —: a later block whose targets ALL sit further than that from the source's centre line skips the
//        source's cold record and its whole deflection / deficit / turbulence pass.
struct SrcLog {
  float Gy, Gwt, sy0d, sz0d;        // circulations of the transverse pass (table path), deflection sigma_0
  float sM, tan_th0, sy0v, snw;     // sqrt(ct), deflection angle; deficit sigma_y0, near-wake sigma
  float kdef, x0d, kyd, pj;         // deficit amplitude factor; column 0: deflection near-wake length, expansion rate, log prefactor
  float x0v, kyv, ch_pref, ix0v;    // column 0: deficit near-wake length, expansion rate; +-Crespo-Hernandez prefactor; 1 / x0v
  float inv_s0d, d0;                // derived once per source step, not per target slot: 1 / (sy0d sz0d), tan_th0 x0d
};
struct ColdRec { float4 a, b, c; };  // the cold part as stored: {sy0d, sz0d, sM, tan_th0} {sy0v, x0d, kyd, pj} {x0v, kyv, ch_pref, ix0v}
static_assert(WF_LOG_HOT_FLOATS == 2 && WF_LOG_COLD_FLOATS == 12, "source log record");
struct WfLogSide { float TI0, TI1, TI2, dTI; };  // the source's column TIs before mixing, the yaw-added-recovery increment

}  // namespace

// ---------------------------------------------------------------------------------------------
// Pair table in target-block order.  One thread per (source i, target t); the record is the one wf_pair_table_kernel
// writes (wf_device.h: WF_PAIR_*), at  block_offset(J) + (i * G + t - J G) * 44,  J = t / G, for i < (J + 1) G.
// ---------------------------------------------------------------------------------------------
__global__ void wf_pair_table_ll_kernel(const WfPairConsts pc, int G, const double* __restrict__ gx,
                                        const double* __restrict__ gy, float* __restrict__ tab, size_t group_floats,
                                        int* __restrict__ cross_tie) {
  const int i = blockIdx.x, t = threadIdx.x, grp = blockIdx.y;
  const int nblk = (pc.N + G - 1) / G;
  if (t >= nblk * G) return;
  gx += (size_t)grp * pc.N;
  gy += (size_t)grp * pc.N;
  const int J = t / G;
  if (i >= (J + 1) * G) {  // a source of a later block: never applied to this target — unless it ties with it in x'
    if (t < pc.N && gx[t] - gx[i] >= 0.0) atomicOr(&cross_tie[grp], 1);
    return;
  }
  float* o = tab + (size_t)grp * group_floats + wf_ll_block_offset(J, pc.N, G) + ((size_t)i * G + (t - J * G)) * WF_PAIR_STRIDE;
  wf_pair_record(pc, gx, gy, i, t, o);
}

extern "C" size_t wfk_ll_table_floats(int N, int G) { return wf_ll_block_offset((N + G - 1) / G, N, G); }

extern "C" hipError_t wfk_launch_pair_table_ll(const WfPairConsts* pc, int G, int n_groups, const double* gx, const double* gy,
                                               float* tab, int* cross_tie, hipStream_t s) {
  const int nblk = (pc->N + G - 1) / G;
  const int threads = ((nblk * G + 63) / 64) * 64;
  hipError_t e = hipMemsetAsync(cross_tie, 0, sizeof(int) * (size_t)n_groups, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(wf_pair_table_ll_kernel, dim3(pc->N, n_groups), dim3(threads), 0, s, *pc, G, gx, gy, tab,
                     wfk_ll_table_floats(pc->N, G), cross_tie);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// The farm step, one target block at a time
// ---------------------------------------------------------------------------------------------
// G: lanes per farm; S: target slots per lane — a block is the G S consecutive (sorted) turbines  t = J G S + p G + sub
// (S = 2 halves the re-reads of the source log per farm at the same lane-group width, i.e. with as many farms per wave
// sharing the per-source phase, for 27 more state registers).  UWS: one wind speed for the whole batch (its derived
// constants live in SGPRs).  WPB waves per block; all of them walk the same chunk sequence, one barrier per chunk.
// TAB: pair-table path (one wind direction per launch group); false: a wind per farm — the transverse pass is evaluated
// on the fly from the farm's own sorted geometry (source coordinates re-read from gx / gy, the targets' kept in
// registers), nothing is staged and no barrier is needed after the start.  MC1 (on the fly only): compile-time skip of
// the ground-mirror vortex cores that are exactly 1.0f in float32 (WfConsts::mirror_core_n <= 1).
#ifdef WF_LL_STAMP  // debug build (tools/ll_stamps.py): wave cycles per phase, summed over the launch
__device__ unsigned long long wf_ll_stamp[16];
#define WF_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define WF_ACC(k, a, b) st_acc[k] += (b) - (a)
#else
#define WF_T(v)
#define WF_ACC(k, a, b)
#endif
// VEER: wind_veer != 0 [FLORIS gauss.py rCalt]: the rotated Gaussian is not even in z - HH — 9 instead of 6 SOSFS sums per slot,
// one exponential per grid row (wf_kernel_common.h: column_deficit_veer); instantiated for the throughput families only.
// Waves per SIMD the register allocator is asked to make room for: three with one slot (first version of the kernel,
// G = 8: 2 -> 1.87 ms, 3 -> 1.60 ms, 4 -> 2.9 ms with 168 B of spills; G = 4: 3 -> 1.24 ms, 4 -> 1.35 ms with 96 B of
// scratch; HornsRev1 x 65536), two with two slots (54 state registers).
// OCC2 (one-slot families on the table path): the same kernel compiled for TWO waves per SIMD instead of three.  Since the
// hot records go through LDS (round 5) the one-slot shapes spill 22-36 registers at the 168 of three waves per SIMD and carry a
// private segment; at 200 registers they do not, and a batch that reaches no third block per CU anyway runs 3-9 % faster from
// this build (profiles/r05_s1_occ2_ab.txt).  launch_ll picks by the launch's blocks per CU; the three-wave build keeps the
// batches whose third block is its whole point.
template <int G, int S, bool UWS, bool TAB, bool MC1, int WPB, bool VEER = false, bool OCC2 = false>
__global__ __launch_bounds__(64 * WPB, ((S == 1 && !OCC2) ? 3 : 2) * 4 / WPB) void wf_step_ll_kernel(
    const WfConsts c, const WfTables* __restrict__ tab, const int* __restrict__ gidx, const double* __restrict__ ws_in,
    const double* __restrict__ wd_in, int wind_stride, const float* __restrict__ yaw_in, float* __restrict__ o_power,
    float* __restrict__ o_ws, float* __restrict__ o_wd, float* __restrict__ o_load, int B, const WfEnvArgs ea,
    const float* __restrict__ ll_tab, size_t group_floats, const int* __restrict__ cross_tie, float* __restrict__ src_log,
    size_t log_cold_offset, size_t log_side_offset, int n_pad, const WfGroupArgs ga, const double* __restrict__ gx,
    const double* __restrict__ gy) {
  constexpr int EPW = 64 / G;   // farms per wave
  constexpr int GS = G * S;     // turbines per block
  constexpr int CH = 64 / GS;   // sources per staged chunk (64 records)
  constexpr int CHUNK_FLOATS = TAB ? 64 * WF_PAIR_STRIDE : 4;
  __shared__ TableLds T;
  __shared__ __attribute__((aligned(16))) float prow[2][CHUNK_FLOATS];
  __shared__ unsigned risk_lds[WPB][EPW];
  __shared__ int env_lds[WPB][EPW];  // farm index of every farm slot of the block (-1 - index: results dropped)
  extern __shared__ __attribute__((aligned(16))) float yaw_lds[];  // [WPB][n_pad] float4 far bounds, then (table path) [WPB][2][HOT_F4] float4 hot records

  if (ga.res_zero && blockIdx.x == 0 && threadIdx.x == 0) *ga.res_zero = 0;  // the re-solve counter of the NEXT step (before any early return)
  int grp = 0;
  if (ga.blk_group) {
    grp = ga.blk_group[(blockIdx.x * (WPB * EPW)) / ga.blk_unit];
    if (grp < 0) return;  // whole block, before any barrier
    grp = (grp + ga.shift) % ga.mod;
  }
  if constexpr (TAB) {
    if (cross_tie[grp]) return;  // this direction has an x' tie across a block boundary: wf_step_kernel serves it
    ll_tab += (size_t)grp * group_floats;
  }

  // chunk q of this direction's table -> LDS buffer q & 1 (the chunks of all target blocks are contiguous, in the order
  // they are consumed)
  auto stage_chunk = [&](int q) {
    const char* g0 = reinterpret_cast<const char*>(ll_tab) + (size_t)q * (CHUNK_FLOATS * 4);
    char* l0 = reinterpret_cast<char*>(&prow[q & 1][0]);
    for (int ch = (int)(threadIdx.x >> 6); ch < CHUNK_FLOATS * 4 / 1024; ch += WPB)
      lds_dma16(g0 + ch * 1024 + (threadIdx.x & 63) * 16, l0 + ch * 1024, false);
  };
  if constexpr (TAB) stage_chunk(0);
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
  const int eiw = lane / G;
  const int slot = (blockIdx.x * WPB + wave) * EPW + eiw;  // launch slot of the farm: indexes the source log
  int env_raw = slot + ga.env_base;  // (env_base: a mixed launch's farm range; the source log is indexed by the slot)
  if (ga.perm) env_raw = ga.perm[env_raw];
  bool env_ok = env_raw >= 0 && env_raw < (ga.env_end ? ga.env_end : B);
  const int env = env_ok ? env_raw : (B - 1);
  // a wind per farm: a farm whose own geometry has an x' tie across a block boundary is left to wf_step_kernel, which
  // is enqueued behind this kernel for exactly those farms (its results here are computed and dropped)
  if constexpr (!TAB) env_ok = env_ok && !cross_tie[env];
  if (sub == 0) risk_lds[wave][eiw] = 0u;
  const int N = c.N;
  const int nblk = (N + GS - 1) / GS;

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
  const float ovh = 0.5f * c.guard_inv;
  const float ovs[3] = {uni(ovh * Ui[0] * c.inv_overlap_thr), uni(ovh * Ui[1] * c.inv_overlap_thr), uni(ovh * Ui[2] * c.inv_overlap_thr)};
  const float ovc = 0.5f - ovh;

  const size_t gofs = TAB ? (size_t)grp * N : (size_t)env * N;  // sorted geometry of the direction (group 0 for a shared wind) / of the farm
  const size_t yofs = (size_t)env * N;
  // The fused MDP transition as in wf_step_kernel, over the env state of the wave's farms: the wave walks the
  // flattened (farm of the wave, turbine) space 64 elements at a time — a wave-instruction reads / writes one or two contiguous
  // row pieces.  The commanded yaw itself is NOT kept in LDS any more (round 5; rounds 2-4: 40 KiB per block at 128 farms x 80
  // turbines): a block fetches its own turbines' yaw from the caller's array / the env state one block ahead (load_yaw below),
  // which left room for the hot-record buffers at two blocks per CU.
  const bool env_mode = ea.yaw_state != nullptr;
  int moves_new = 0;
  if (env_mode && ea.action) moves_new = ea.moves[env] + 1;
  if (sub == 0) env_lds[wave][eiw] = env_ok ? env : -1 - env;  // (a farm whose results are dropped still has a valid row to read)
  if (env_mode && ea.action) {
    int f = 0, o = lane;
    while (o >= N) { o -= N; ++f; }
    while (f < EPW) {
      const int e_raw = env_lds[wave][f];
      const bool okf = e_raw >= 0;
      const int e = okf ? e_raw : -1 - e_raw;
      const size_t oi = (size_t)e * N + o;
      float yw = ea.yaw_state[oi];
      float a = ea.action[oi];
      float acc = ea.acc[oi];
      const float frac = __fdiv_rn(__fdiv_rn(__fdiv_rn(acc, ea.rate), (float)(ea.moves[e] + 1)), ea.dt);
      if (frac >= ea.budget) a = 0.0f;
      if (ea.discrete) a = (a - 1.0f) * ea.yaw_step;
      if (!ea.discrete) a = fminf(fmaxf(a, -ea.yaw_step), ea.yaw_step);
      yw = fminf(fmaxf(yw + a, ea.yaw_lo), ea.yaw_hi);
      acc += fabsf(a);
      if (okf) {
        ea.yaw_state[oi] = yw;
        ea.acc[oi] = acc;
        if (ea.yaw_out) ea.yaw_out[oi] = yw;
      }
      o += 64;
      while (o >= N) { o -= N; ++f; }
    }
  }
  // the commanded yaw of turbine o (caller's order) of this lane's farm.  After a transition it is read back from the env state
  // this wave has just written — at agent scope: the transition above READ those lines, the vector L1 is not updated by this
  // CU's stores, and a plain load could hit the old value.  (A farm whose results are dropped — padding slots — was not written
  // and solves on the old yaw: nobody sees it.)
  const float* const yaw_src = (env_mode ? ea.yaw_state : yaw_in) + yofs;
  const bool yaw_fresh = env_mode && ea.action;
  auto load_yaw = [&](int o) { return yaw_fresh ? __hip_atomic_load(yaw_src + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : yaw_src[o]; };
  // (the move counters are read above by every lane that holds an element of the farm, written here by the farm's first
  // lane: the same wave, in program order)
  if (env_mode && ea.action && sub == 0 && env_ok) ea.moves[env] = moves_new;
  if constexpr (TAB) wf_dma_wait();  // (chunk 0 of the table)
  __syncthreads();

  // ---- per-turbine state of the ONE target block in registers (as Slots<1> of wf_step_kernel) ----
  constexpr int NE = VEER ? 9 : 6;
  float esq[S][NE], V[S][9], W[S][9], TI[S][3];
  const float amb0 = fsqrt(c.amb2);

  // transverse velocities of one source on this lane's target [A.3-4]: record = 9 float4 {aV, bV, aW, bW}
  auto apply_tab = [&](auto PP, const float4* pr, float Gy, float Gwt) {
    constexpr int p = decltype(PP)::value;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float4 cf[3] = {pr[3 * j], pr[3 * j + 1], pr[3 * j + 2]};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int q = j * 3 + k;
        V[p][q] = fmaf(Gwt, cf[k].y, fmaf(Gy, cf[k].x, V[p][q]));
        const float ww = fmaf(Gwt, cf[k].w, Gy * cf[k].z);
        W[p][q] += fmaxf(ww, 0.0f);  // W[W<0] = 0, quirk (5)
      }
    }
  };

  // The same for a LOGGED source, whose hot record holds {rho = Gy / Gwt, Gwt}:  max(Gy aW + Gwt bW, 0) = Gwt max(rho aW + bW, 0)
  // for Gwt > 0 — the wake-rotation circulation gam_wr (a - a^2) ubar is positive — and Gwt min(rho aW + bW, 0) for Gwt < 0
  // (a rotor wind speed driven negative by an unphysically tight layout: the reference keeps computing, so does this):
  // both are  Gwt med3(rho aW + bW, 0, copysign(inf, Gwt)).  Five instead of six instructions per grid point — two FMAs for
  // V, FMA + med3 + FMA for W.  (Transverse velocities switched off: Gwt = 0 and rho = 0 in the record: nothing is added.)
  auto apply_tab_ratio = [&](auto PP, const float4* pr, float rho, float Gwt) {
    constexpr int p = decltype(PP)::value;
    const float lim = copysignf(__builtin_inff(), Gwt);
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const float4 cf = pr[q];
      V[p][q] = fmaf(Gwt, fmaf(rho, cf.x, cf.y), V[p][q]);
      W[p][q] = fmaf(Gwt, __builtin_amdgcn_fmed3f(fmaf(rho, cf.z, cf.w), 0.0f, lim), W[p][q]);  // W[W<0] = 0, quirk (5)
    }
  };

  // the same on the fly (a wind per farm): wf_step_kernel's apply_fly — 7 + 7 distinct vortex offsets per grid column,
  // circulations Gt = gam_top Gy, Gb = -gam_bot Gy (the tip vortices share their farm-dependent factor)
  auto apply_fly = [&](auto PP, float dx, float dy, float Gy, float Gwt) {
    constexpr int p = decltype(PP)::value;
    const float Gt = c.gam_top * Gy, Gb = -c.gam_bot * Gy;
    float dec[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) dec[k] = frcp(fmaf(c.decay_a[k], dx, 1.0f));
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float yL = dy + c.yoff[j];
      const float yL2 = yL * yL;
      const float Ey = fexp2(-yL2 * c.exp_c);
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
          A[mi - 2] = fmaf(Gwt, pr - pm, A[mi - 2]);  // rotation, real - mirror
          Bw[mi - 2] = fmaf(Gwt, tr - tm, Bw[mi - 2]);
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        V[p][j * 3 + k] = fmaf(A[k], dec[k], V[p][j * 3 + k]);
        W[p][j * 3 + k] += fmaxf(-yL * Bw[k] * dec[k], 0.0f);  // W[W<0] = 0, quirk (5)
      }
    }
  };

  // deflection + deficit + SOSFS + wake-added turbulence of one source on this lane's target [A.3-3, 6, 7, 8]
  // (the body of wf_step_kernel's pass 2 for one slot; R holds the source's constants, ex = {dx, dy, tipow, bits})
  // (on the fly: ex.z is 1 / 0 for the 15 D reach and ex.w carries bit 3 only; the lateral gates and the
  // Crespo-Hernandez distance factor are evaluated where they are needed, from the float64 y' of target and source)
  auto pass2 = [&](auto PP, const SrcLog& R, const float* side, bool side_from_log, const float4 ex, bool act,
                   double yt_d = 0.0, double yi_d = 0.0) {
    constexpr int p = decltype(PP)::value;
    if (!act) return;
    const float dx = ex.x, dy = ex.y;
    const bool in15 = ex.z > 0.0f;
    int bits = __float_as_int(ex.w);
    const float lin = fmaf(c.bd, dx, c.ad);
    const float amp_on = (bits & 8) ? 1.0f : 0.0f;
    SrcConsts sc;
    sc.sy0d = R.sy0d; sc.sz0d = R.sz0d; sc.inv_s0d = R.inv_s0d; sc.lnA = 1.6f + R.sM; sc.lnB = 1.6f - R.sM; sc.sM = R.sM;
    sc.tan_th0 = R.tan_th0; sc.sy0v = R.sy0v; sc.snw = R.snw; sc.kdef = R.kdef;
    const float d0 = R.d0, ix0v = R.ix0v;
    float e1[3], e0[3];
    float e2[VEER ? 3 : 1];  // with veer: the deficit of row k = 2 (e0 is row 0)
    const bool same = !__any(R.ch_pref < 0.0f);  // the sign of ch_pref flags a split-TI source
    if constexpr (VEER) {
      // (general form per column; the split-TI side record is handled as below)
      const bool mine = R.ch_pref < 0.0f;
      WfLogSide X = {0.0f, 0.0f, 0.0f, 0.0f};
      if (!same && mine) {
        if (side_from_log) {
          X.TI0 = __hip_atomic_load(side, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          X.TI1 = __hip_atomic_load(side + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          X.TI2 = __hip_atomic_load(side + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          X.dTI = __hip_atomic_load(side + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          X.TI0 = side[0]; X.TI1 = side[1]; X.TI2 = side[2]; X.dTI = side[3];
        }
      }
      const float s_c = fsqrt(fmaxf(1.0f - R.sM * R.sM, 0.0f));
      const float om_sc = R.sM * R.sM * frcp(1.0f + s_c);
      const float b2om = c.beta2 * om_sc, b2om_d = c.beta2_d * om_sc;
      const float x0num_d = R.x0d * fmaf(c.alpha4_d, X.TI0, b2om_d);
      const float x0num_v = R.x0v * fmaf(c.alpha4, X.TI0 + X.dTI, b2om);
      const float pfac = R.pj * R.kyd;
      const float tis[3] = {X.TI0, X.TI1, X.TI2};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        ColConsts k;
        if (j == 0 || same || !mine) {
          k.x0d = R.x0d; k.kyd = R.kyd; k.d0 = d0; k.pj = R.pj; k.x0v = R.x0v; k.ix0v = ix0v; k.kyv = R.kyv;
        } else {
          k.x0d = x0num_d * frcp(fmaf(c.alpha4_d, tis[j], b2om_d));
          k.kyd = fmaf(c.ka_d, tis[j], c.kb_d);
          k.d0 = sc.tan_th0 * k.x0d;
          k.pj = pfac * frcp(k.kyd);
          k.x0v = x0num_v * frcp(fmaf(c.alpha4, tis[j] + X.dTI, b2om));
          k.ix0v = frcp(k.x0v);
          k.kyv = fmaf(c.ka, tis[j] + X.dTI, c.kb);
        }
        column_deficit_veer(c, sc, k, dx, dy + c.off[j], lin, amp_on, e0[j], e1[j], e2[j]);
      }
    } else if (same) {
      const bool far = dx >= R.x0v;
      const float up = dx * ix0v;
      const float xf = dx - R.x0v;
      const float sy = far ? fmaf(R.kyv, xf, sc.sy0v) : fmaf(up, sc.sy0v - sc.snw, sc.snw);
      const float xs = fmaxf(dx - R.x0d, 0.0f);
      const float syd = fmaf(R.kyd, xs, sc.sy0d), szd = fmaf(R.kyd, xs, sc.sz0d);
      const float s = fsqrt(syd * szd * sc.inv_s0d);
      const float arg = sc.lnA * fmaf(1.6f, s, -sc.sM) * frcp(sc.lnB * fmaf(1.6f, s, sc.sM));
      const float d_far = fmaf(R.pj, flog2(arg), d0);
      const float delta = ((dx > R.x0d) ? d_far : dx * sc.tan_th0) + lin;
      // Far-pair skip.  More than 6.12 sigma_y + D/4 off the wake's centre line, the nearest grid column gets exp2(-27) =
      // 7.5e-9 of the amplitude: below the resolution of 1 - sqrt(esq) in float32 and far below the overlap threshold, so
      // the deficit, the SOSFS update and the wake-added TI of this pair are all no-ops (the TI candidate is the ambient
      // value the running maximum starts from; the clamped overlap terms are exactly 0: no flag).  Wave-uniform: on the
      // table path the farms of a wave share the geometry; on the fly they do once wf_set_wind has put farms of like
      // direction into one wave.  c.far_k = 6.12 (1e30 with the skip disabled: wf_kernel_choice::far_skip = 0 — never taken).
      // Sources whose wake cannot reach ANY target of the block never get here: far_bound() / the replay loop.
      if (!__any(fabsf(dy - delta) < fmaf(c.far_k, sy, c.off[2]))) return;
      const float sz = far ? fmaf(R.kyv, xf, c.sz0v) : fmaf(up, c.sz0v - sc.snw, sc.snw);
      const float isy = frcp(sy), isz = frcp(sz);
      const float xarg = sc.kdef * isy * isz;
      const float C = (xarg >= 1.0f) ? 1.0f : xarg * frcp(1.0f + fsqrt(fmaxf(1.0f - xarg, 0.0f)));
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
      // split-TI source (rare): the other two columns' constants are re-derived from the log — the numerators of the
      // near-wake lengths and the log prefactor follow from column 0's values
      // (wave-uniform branch: SOME farm of the wave has a split-TI source; the lanes of the other farms keep column 0's
      // constants for all three columns, and only the split farms have a side record)
      const bool mine = R.ch_pref < 0.0f;
      WfLogSide X = {0.0f, 0.0f, 0.0f, 0.0f};
      if (!mine) {
      } else if (side_from_log) {  // agent-scope loads: served by L2, where the writer's store went
        X.TI0 = __hip_atomic_load(side, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        X.TI1 = __hip_atomic_load(side + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        X.TI2 = __hip_atomic_load(side + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        X.dTI = __hip_atomic_load(side + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        X.TI0 = side[0]; X.TI1 = side[1]; X.TI2 = side[2]; X.dTI = side[3];
      }
      const float s_c = fsqrt(fmaxf(1.0f - R.sM * R.sM, 0.0f));       // sM^2 = ct
      const float om_sc = R.sM * R.sM * frcp(1.0f + s_c);
      const float b2om = c.beta2 * om_sc, b2om_d = c.beta2_d * om_sc;
      const float x0num_d = R.x0d * fmaf(c.alpha4_d, X.TI0, b2om_d);
      const float x0num_v = R.x0v * fmaf(c.alpha4, X.TI0 + X.dTI, b2om);
      const float pfac = R.pj * R.kyd;
      const float tis[3] = {X.TI0, X.TI1, X.TI2};
      auto col_consts = [&](int j) {
        ColConsts k;
        if (j == 0 || !mine) {
          k.x0d = R.x0d; k.kyd = R.kyd; k.d0 = d0; k.pj = R.pj; k.x0v = R.x0v; k.ix0v = ix0v; k.kyv = R.kyv;
        } else {
          k.x0d = x0num_d * frcp(fmaf(c.alpha4_d, tis[j], b2om_d));
          k.kyd = fmaf(c.ka_d, tis[j], c.kb_d);
          k.d0 = sc.tan_th0 * k.x0d;
          k.pj = pfac * frcp(k.kyd);
          k.x0v = x0num_v * frcp(fmaf(c.alpha4, tis[j] + X.dTI, b2om));
          k.ix0v = frcp(k.x0v);
          k.kyv = fmaf(c.ka, tis[j] + X.dTI, c.kb);
        }
        return k;
      };
      // One evaluation of the column-independent part (deflection, sigmas, amplitude) per DISTINCT set of constants:
      // a turbine partly inside an upstream lateral gate has one or two columns covered, so two of its three TIs agree
      // — two evaluations instead of three (wave-uniform: the farms of a wave share the pattern on the table path)
      const bool need1 = __any(mine && X.TI1 != X.TI0), need2 = __any(mine && X.TI2 != X.TI0);
      const bool same12 = !__any(mine && X.TI2 != X.TI1);
      const ColWake w0 = column_wake(c, sc, col_consts(0), dx, lin, amp_on);
      ColWake w1 = w0, w2 = w0;
      if (need1) w1 = column_wake(c, sc, col_consts(1), dx, lin, amp_on);
      if (need2) {
        if (need1 && same12) w2 = w1;
        else w2 = column_wake(c, sc, col_consts(2), dx, lin, amp_on);
      }
      column_rows(w0, dy + c.off[0], e1[0], e0[0]);
      column_rows(w1, dy + c.off[1], e1[1], e0[1]);
      column_rows(w2, dy + c.off[2], e1[2], e0[2]);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if constexpr (VEER) {
        esq[p][3 * j] = fmaf(e0[j], e0[j], esq[p][3 * j]);
        esq[p][3 * j + 1] = fmaf(e1[j], e1[j], esq[p][3 * j + 1]);
        esq[p][3 * j + 2] = fmaf(e2[j], e2[j], esq[p][3 * j + 2]);
      } else {
        esq[p][2 * j] = fmaf(e0[j], e0[j], esq[p][2 * j]);
        esq[p][2 * j + 1] = fmaf(e1[j], e1[j], esq[p][2 * j + 1]);
      }
    }
    // wake-added TI [A.3-8]: only within 15 D downstream and 2 D laterally (float64 decisions)
    float tipow = ex.z;
    if constexpr (TAB) {
      if (!__any(in15 && (bits & 7))) return;
    } else {
      if (!__any(in15 && (fabsf(dy) < c.twoD + c.off[2] + 1.0f))) return;  // float32 prefilter with a margin
      const double twoD_d = 8.0 * c.q_d;
      bits |= (fabs(yi_d - (yt_d - c.q_d)) < twoD_d) ? 1 : 0;
      bits |= (fabs(yi_d - yt_d) < twoD_d) ? 2 : 0;
      bits |= (fabs(yi_d - (yt_d + c.q_d)) < twoD_d) ? 4 : 0;
      const float dxp = (bits & 8) ? dx : dx + 1.0f;
      tipow = fexp2(c.ch_down * flog2(dxp * c.invD));
    }
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
    const float ti = fabsf(R.ch_pref) * tipow;
    const float tia = in15 ? ti * (cnt * (1.0f / 9.0f)) : 0.0f;
    const float cand = fsqrt(fmaf(tia, tia, c.amb2));
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float cm = (bits & (1 << j)) ? cand : 0.0f;
      TI[p][j] = __uint_as_float(max(__float_as_uint(TI[p][j]), __float_as_uint(cm)));
    }
  };

  float psum = 0.0f, lsum = 0.0f;  // per-lane partial sums for the fused reward
  const int n_real = ga.n_real ? ga.n_real[env] : N;  // turbines the farm really has (padded layouts: WfGroupArgs)
#ifdef WF_LL_STAMP
  unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  WF_T(st_begin);
#endif
  int q = 0;                        // running chunk index (LDS buffer q & 1)
  // Cache-line discipline of the log (the vector L1 is not updated by this CU's own stores): a farm's records start on
  // a 128-byte line (n_pad is a multiple of G, G is even), so a line belongs to ONE block and is never read before that
  // block has written it.  The 16-byte side records do share lines across blocks; they are read past the L1.
  // the wave's parts of the log: record i of the hot part at i * EPW float2, of the cold part at i * 3 EPW float4
  const size_t wave_rec0 = (size_t)(slot - eiw) * n_pad;  // (records before this wave's, per farm slot)
  // hot part: the records of sources 2m and 2m + 1 are the two halves of ONE float4 per farm — [wave][m][farm] float4 —, so
  // that the transverse pass of the table path fetches two sources with one load (a block's sources are whole pairs: G S
  // is even, and so is every chunk's count of logged sources)
  float2* const log_hot = reinterpret_cast<float2*>(src_log + wave_rec0 * WF_LOG_HOT_FLOATS);
  float4* const log_cold = reinterpret_cast<float4*>(src_log + log_cold_offset + wave_rec0 * WF_LOG_COLD_FLOATS) + eiw;
  float4* const bndL = reinterpret_cast<float4*>(yaw_lds) + (size_t)wave * n_pad;  // far bounds
  // Table path (round 5): the HOT records of a chunk's logged sources are staged into wave-private LDS by LDS-DMA a whole chunk
  // (or a block's output phase) ahead of their use, like the chunk's pair records.  Rounds 2-4 fetched them into registers one
  // loop iteration ahead — all the compiler's s_waitcnt vmcnt(0) in front of every iteration's first LDS read leaves room for —
  // and the transverse pass waited for the log: 6 % of the launch (profiles/r05_ablate_log.txt: the pass without the load).
  // A chunk's CH / 2 source pairs x EPW farms x 16 bytes are one contiguous run of the wave-major log: whole 1-KiB
  // wave-instructions (a run shorter than 1 KiB reads on into the following records, which nobody looks at).
  constexpr int HOT_INSTR = (CH / 2) * EPW * 16 >= 1024 ? (CH / 2) * EPW * 16 / 1024 : 1;  // wave-instructions per chunk
  constexpr int HOT_F4 = HOT_INSTR * 64;                                                   // float4 per wave and buffer
  float4* const hotL = reinterpret_cast<float4*>(yaw_lds) + (size_t)WPB * n_pad + (size_t)wave * (2 * HOT_F4);
  auto stage_hot = [&](int qq, int i_first) {  // the hot records of sources i_first .. i_first + CH - 1 -> buffer qq & 1
    const char* g0 = reinterpret_cast<const char*>(reinterpret_cast<const float4*>(log_hot) + (size_t)(i_first >> 1) * EPW);
    char* l0 = reinterpret_cast<char*>(hotL + (qq & 1) * HOT_F4);
#pragma unroll
    for (int ch = 0; ch < HOT_INSTR; ++ch) lds_dma16(g0 + ch * 1024 + lane * 16, l0 + ch * 1024, true);
    // (agent scope, served by the L2.  The run may hold records that are not written yet — the chunk's own
    // sources, the records behind a short run — and the vector L1 is not updated by this CU's later stores: a line cached
    // now would be stale when a later block stages it again.  Every record is read once per block: nothing is lost.)
  };
  auto hot_index = [&](int i) { return 2 * ((size_t)(i >> 1) * EPW + eiw) + (i & 1); };
  auto load_hot = [&](int i) { return log_hot[hot_index(i)]; };
  auto load_cold = [&](int i) {
    const float4* lp = log_cold + (size_t)i * (3 * EPW);
    ColdRec r;
    r.a = lp[0]; r.b = lp[EPW]; r.c = lp[2 * EPW];
    return r;
  };
  // the in-register record of a replayed source from its two parts (pass2 does not read Gy / Gwt)
  const float snw_f = c.near_c * 0.70710678118f;
  auto unpack = [&](const float2 h, const ColdRec& r) {
    SrcLog R;
    R.Gy = h.x * h.y; R.Gwt = h.y;  // (hot record: {Gy / Gwt, Gwt})
    R.sy0d = r.a.x; R.sz0d = r.a.y; R.sM = r.a.z; R.tan_th0 = r.a.w;
    R.sy0v = r.b.x; R.x0d = r.b.y; R.kyd = r.b.z; R.pj = r.b.w;
    R.x0v = r.c.x; R.kyv = r.c.y; R.ch_pref = r.c.z; R.ix0v = r.c.w;
    R.snw = snw_f * R.sM;                               // 0.501 D sqrt(ct / 2)
    R.kdef = (R.sM * R.sM) * (R.sy0v * c.kdef_sy0v);   // ct cos(yaw) D^2 / 8, cos(yaw) = sy0v / (sz0v [cos veer])
    R.inv_s0d = frcp(R.sy0d * R.sz0d);
    R.d0 = R.tan_th0 * R.x0d;
    return R;
  };
  float* const logx = src_log + log_side_offset + (size_t)slot * n_pad * WF_LOG_SIDE_FLOATS;
  // caller's (unsorted) index of this lane's turbines — the current block's, the next block's, the one after — and the next
  // block's commanded yaw: indices are fetched two blocks ahead, the yaw behind them one block ahead
  int oidx[S], oidx_nx[S], oidx_n2[S];
  float yaw_nx[S];
#pragma unroll
  for (int p = 0; p < S; ++p) {
    oidx_nx[p] = gidx[gofs + min(p * G + sub, N - 1)];
    oidx_n2[p] = gidx[gofs + min(GS + p * G + sub, N - 1)];
  }
#pragma unroll
  for (int p = 0; p < S; ++p) yaw_nx[p] = load_yaw(oidx_nx[p]);
  for (int J = 0; J < nblk; ++J) {
    int tt[S];
    bool tvalid[S];
    float yaw_t[S], sg_t[S], cg_t[S];
    double xt_d[TAB ? 1 : S] = {}, yt_d[TAB ? 1 : S] = {};  // on the fly: this lane's targets' sorted coordinates (float64:
#pragma unroll
    for (int p = 0; p < S; ++p) {
      tt[p] = J * GS + p * G + sub;
      tvalid[p] = tt[p] < N;
      if constexpr (!TAB) {
        xt_d[p] = tvalid[p] ? gx[gofs + tt[p]] : -1.0e300;  // padding is never downstream of anything
        yt_d[p] = gy[gofs + (tvalid[p] ? tt[p] : 0)];
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) { V[p][k] = 0.0f; W[p][k] = 0.0f; }
#pragma unroll
      for (int k = 0; k < NE; ++k) esq[p][k] = 0.0f;
#pragma unroll
      for (int j = 0; j < 3; ++j) TI[p][j] = amb0;
      // (the caller's index of this lane's turbine was fetched a block ahead: a load issued here would be waited for at
      // the first LDS read of the replay, which the compiler guards with s_waitcnt vmcnt(0))
      oidx[p] = oidx_nx[p];
      yaw_t[p] = yaw_nx[p];
      oidx_nx[p] = oidx_n2[p];
      oidx_n2[p] = gidx[gofs + min(tt[p] + 2 * GS, N - 1)];
      yaw_nx[p] = load_yaw(oidx_nx[p]);
      sincos_yaw(yaw_t[p] * kDeg2Rad, sg_t[p], cg_t[p]);  // this lane's turbines: source constants (own block), power output
    }

    const int first_own = J * GS;                 // sources [0, first_own) come from the log, [first_own, n_src) are this block's
    const int n_src = min(N, first_own + GS);
    const int n_chunks = (n_src + CH - 1) / CH;
    // the first logged source's record is fetched ahead; every later one while its predecessor is being applied
    float2 hot_nx = make_float2(0.0f, 0.0f);   // on the fly: the hot record of the next source
    ColdRec cold_nx = {};             // (on the fly; the table path fetches cold records per chunk, for the near sources only)
    double xs_nx = 0.0, ys_nx = 0.0;  // on the fly: the sorted coordinates of that source come with its record
    if (first_own > 0) {
      if constexpr (!TAB) {
        hot_nx = load_hot(0);
        cold_nx = load_cold(0);
        xs_nx = gx[gofs];
        ys_nx = gy[gofs];
      }
    }

    // on the fly: the {dx, dy, 15 D reach, bit 3} of (source at (xi, yi), this lane's target in slot p), decided in float64
    auto fly_record = [&](auto PP, double xi, double yi) {
      constexpr int p = decltype(PP)::value;
      constexpr int q = TAB ? 0 : p;
      float4 r;
      r.x = (float)(xt_d[q] - xi);
      // the lateral offset from the float64 coordinates, as the pair table has it: the difference of two float32
      // y' - yc (2.4e-4 m apart at 2.6 km) is 6e-7 of a 400 m offset — harmless on its own, but on an aligned grid the
      // transverse velocities of symmetric neighbours cancel and leave that error standing (wd 3e-4 deg, power 3e-4 on
      // a 112-turbine grid at exactly 360 deg: tests/tools/wd_error_probe.py)
      r.y = (float)(yt_d[q] - yi);
      r.z = (xt_d[q] <= xi + c.fifteenD_d) ? 1.0f : 0.0f;
      r.w = __int_as_float((xt_d[q] > xi + 0.1) ? 8 : 0);
      return r;
    };

    // ---- one source of THIS block: the sequential recurrence, as wf_step_kernel's slot 0.  PS: the slot the source lives
    // in; recs: its G S records in the staged chunk ----------------------------------------------------------------
    auto own_source = [&](auto PS, int i, const float* recs) {
      constexpr int ps = decltype(PS)::value;
      const int src = gbase + ((i - first_own) - ps * G);  // owner lane
      // The own-source step is a dependent chain (~4 300 cycles of latency); the SIMD's other wave is usually in its replay,
      // which is throughput: with equal priority the two alternate and every instruction of the chain also waits for its turn.
      // Raised priority for the chain: HornsRev1 x 65 536 0.902 -> 0.861 ms on one box (profiles/r05_setprio_ab.txt).
      __builtin_amdgcn_s_setprio(3);
      WF_T(so_0);
      // A. the source's state
      float vsum = 0.0f;
#pragma unroll
      for (int k2 = 0; k2 < 9; ++k2) vsum += V[ps][k2];
      const float m3 = __shfl(cube_sum(esq[ps]), src);
      const float Vmean = __shfl(vsum, src) * (1.0f / 9.0f);
      float TIs[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) TIs[j] = __shfl(TI[ps][j], src);
      const float yaw_i = __shfl(yaw_t[ps], src);
      const float cg = __shfl(cg_t[ps], src), sg = __shfl(sg_t[ps], src);
      // B. circulations [A.3-1, A.3-4]
      WF_T(so_1);
      WF_ACC(6, so_0, so_1);
      SrcLog Sc;
      const float ubar = fcbrt_pos(m3 * (1.0f / 9.0f));
      unsigned trisk;
      const float ct = table_ct(c, T, ubar, trisk) * cg;
      if (trisk) atomicOr(&risk_lds[wave][eiw], trisk);
      const float sq1 = fsqrt(1.0f - ct * cg);
      const float a = 0.5f * ct * frcp(1.0f + sq1);
      const float Gwr = c.gam_wr * (a - a * a) * ubar;
      const float gt = c.gam_top * ws * ct, gb = c.gam_bot * ws * ct;
      const float scg = sg * cg * c.sw_tv;
      Sc.Gy = scg * ct * ws;
      Sc.Gwt = Gwr * c.sw_tv;
      WF_T(so_2);
      WF_ACC(7, so_1, so_2);
      // C. pass 1: the source's own slot (lanes upstream of it carry dx < 0 in their record), every later slot of the
      // block, and the lanes of EARLIER slots that tie with it in x' (dx = 0 counts as downstream here [A.3-4])
      float4 ex[S];
      double xi_d = 0.0, yi_d = 0.0;
      if constexpr (!TAB) {  // the source's own coordinates (every lane of the group reads the same two words)
        xi_d = gx[gofs + i];
        yi_d = gy[gofs + i];
      }
      auto pass1_slot = [&](auto PP) {
        constexpr int p = decltype(PP)::value;
        if constexpr (TAB) {
          const float* rec = recs + (p * G + sub) * WF_PAIR_STRIDE;
          ex[p] = *reinterpret_cast<const float4*>(rec + WF_PAIR_DX);  // {dx, dy, tipow, decision bits}
          const bool act1 = (p > ps) ? tvalid[p] : (ex[p].x >= 0.0f);
          if (p >= ps || __any(act1)) {
            if (act1) apply_tab(PP, reinterpret_cast<const float4*>(rec), Sc.Gy, Sc.Gwt);
          }
        } else {
          ex[p] = fly_record(PP, xi_d, yi_d);
          const bool act1 = (p > ps) ? tvalid[p] : (ex[p].x >= 0.0f);
          if (p >= ps || __any(act1)) {
            if (act1) apply_fly(PP, ex[p].x, ex[p].y, Sc.Gy, Sc.Gwt);
          }
        }
      };
      static_for<S>(pass1_slot);
      WF_T(so_3);
      WF_ACC(8, so_2, so_3);
      float vbar = 0.0f, wbar = 0.0f;
#pragma unroll
      for (int k2 = 0; k2 < 9; ++k2) { vbar += V[ps][k2]; wbar += W[ps][k2]; }
      vbar = __shfl(vbar, src) * (1.0f / 9.0f);
      wbar = __shfl(wbar, src) * (1.0f / 9.0f);
      // B2. steering + deflection constants [A.3-2, A.3-3]
      float val = c.sw_steer * (Vmean - Gwr * c.ks_core) * frcp(gt * c.ks_top - gb * c.ks_bot);
      val = fminf(fmaxf(val, -1.0f), 1.0f);
      const float asv = __any(fabsf(val) > 0.3f) ? asinf(val) : asin_small(val);
      const float gd = -(yaw_i * kDeg2Rad + 0.5f * asv);
      const float c2h = fsqrt(fmaxf(fmaf(-val, val, 1.0f), 0.0f));
      const float chh = fsqrt(0.5f * (1.0f + c2h));
      const float shh = 0.5f * val * frcp(chh);
      const float cgd = fmaf(cg, chh, -sg * shh);
      const float s_cc = fsqrt(1.0f - ct * cgd), s_c = fsqrt(1.0f - ct);
      const float om_scc = ct * cgd * frcp(1.0f + s_cc);
      const float om_sc = ct * frcp(1.0f + s_c);
      Sc.sM = fsqrt(ct);
      const float E0 = fmaf(om_sc, om_sc, fmaf(-c.e0c1, om_sc, c.e0c2));
      Sc.sz0d = 0.5f * c.D * fsqrt((1.0f + s_cc) * frcp(2.0f * (1.0f + s_c)));
      Sc.sy0d = Sc.sz0d * cgd;
      if constexpr (VEER) Sc.sy0d *= c.cos_veer;
      const float th0 = c.dm03 * gd * frcp(cgd) * om_scc;
      {
        const float t2 = th0 * th0;
        const float poly = th0 * fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, 0.0088632355f, 0.0218694885f), 0.0539682540f),
                                                          0.1333333333f), 0.3333333333f), 1.0f);
        Sc.tan_th0 = poly;
        if (__any(fabsf(th0) > 0.35f)) {
          const float rev = th0 * 0.15915494309189535f;
          const float hw = __builtin_amdgcn_sinf(rev) * frcp(__builtin_amdgcn_cosf(rev));
          Sc.tan_th0 = (fabsf(th0) > 0.35f) ? hw : poly;
        }
      }
      const float pfac = th0 * E0 * (1.0f / 5.2f) * fsqrt(Sc.sy0d * Sc.sz0d * frcp(ct)) * kLn2;
      const float x0num_d = c.D * cgd * (1.0f + s_cc) * (1.0f / 1.41421356237f);
      // D. yaw-added recovery [A.3-5] and deficit constants [A.3-6]
      const float I0 = TIs[0];
      const float uI = ubar * I0;
      const float mix2 = (vbar * vbar + wbar * wbar) * (1.0f / 3.0f);
      const float inv_ubar = frcp(ubar);
      const float Itot = fsqrt(fmaf(uI, uI, mix2)) * inv_ubar;
      const float Imix = mix2 * inv_ubar * inv_ubar * frcp(Itot + I0);
      WfLogSide X;
      X.dTI = c.gch_gain * Imix;
      if (lane == src) {
#pragma unroll
        // (the stored TI: max(ambient, TI + dTI), as FLORIS' maximum over all turbines at the end of the source step leaves it —
        // see wf_kernels.hip; the passes of this source go on with TI + dTI)
        for (int j = 0; j < 3; ++j) TI[ps][j] = fmaxf(TI[ps][j] + X.dTI, amb0);
      }
      Sc.sy0v = c.sz0v * cg;
      if constexpr (VEER) Sc.sy0v *= c.cos_veer;
      Sc.snw = c.near_c * fsqrt(0.5f * ct);
      Sc.kdef = ct * cg * c.kdef;
      Sc.ch_pref = c.ch_c * fexp2(c.ch_ai * flog2(a));
      const float x0num_v = c.D * cg * (1.0f + s_c) * (1.0f / 1.41421356237f);
      const float b2om = c.beta2 * om_sc, b2om_d = c.beta2_d * om_sc;
      Sc.x0d = x0num_d * frcp(fmaf(c.alpha4_d, TIs[0], b2om_d));
      Sc.kyd = fmaf(c.ka_d, TIs[0], c.kb_d);
      Sc.pj = pfac * frcp(Sc.kyd);
      Sc.x0v = x0num_v * frcp(fmaf(c.alpha4, TIs[0] + X.dTI, b2om));
      Sc.kyv = fmaf(c.ka, TIs[0] + X.dTI, c.kb);
      X.TI0 = TIs[0]; X.TI1 = TIs[1]; X.TI2 = TIs[2];
      const bool split = !((TIs[0] == TIs[1]) && (TIs[1] == TIs[2]));
      Sc.ch_pref = split ? -Sc.ch_pref : Sc.ch_pref;  // the flag travels in the sign
      Sc.ix0v = frcp(Sc.x0v);
      Sc.inv_s0d = frcp(Sc.sy0d * Sc.sz0d);
      Sc.d0 = Sc.tan_th0 * Sc.x0d;
      // the later blocks replay this source from the log
      if (J + 1 < nblk) {
        if (sub == 0) {
          // (Gwt == 0: transverse velocities switched off, Gy is 0 as well)
          log_hot[hot_index(i)] = make_float2(Sc.Gwt != 0.0f ? Sc.Gy * frcp(Sc.Gwt) : 0.0f, Sc.Gwt);
          float4* lp = log_cold + (size_t)i * (3 * EPW);           // EPW * 16 each
          lp[0] = make_float4(Sc.sy0d, Sc.sz0d, Sc.sM, Sc.tan_th0);
          lp[EPW] = make_float4(Sc.sy0v, Sc.x0d, Sc.kyd, Sc.pj);
          lp[2 * EPW] = make_float4(Sc.x0v, Sc.kyv, Sc.ch_pref, Sc.ix0v);
          if (split) *reinterpret_cast<float4*>(logx + (size_t)i * WF_LOG_SIDE_FLOATS) = make_float4(X.TI0, X.TI1, X.TI2, X.dTI);
        }
        // Far bound of this source over the farms of the wave (far_bound): for every dx >= 0 and every farm
        //   sigma_y(dx) <= max(kyv dx + (sy0v - kyv x0v), max(snw, sy0v))   [far wake: equality; near wake: sigma_y lies
        //                                                                    between snw and sy0v]
        //   |deflection - (ad + bd dx)| <= |tan_th0 x0d| + 2.2 |pj|         [near wake: |dx tan_th0| <= |tan_th0| x0d; far
        //        wake: d0 + pj log2(arg) with arg rising from 1 at sigma = sigma_0 to (1.6 + sM) / (1.6 - sM), whose log2
        //        is <= 2.115 for sM = sqrt(ct) <= 1 — ct is clipped to 0.9999 — and the wake only widens: kyd >= 0]
        // hold; with per-column constants (split TI) the growth rate is taken at the largest column TI, the near-wake
        // length and the log prefactor at the smallest (both fall with TI), and the near-wake credit is dropped.
        // (wind veer: the Gaussian is rotated — r = a yy^2 - 2 b yy zz + c zz^2 >= yy^2 / (2 sigma_max^2), the smaller
        // eigenvalue of the form — so the bound takes the larger of the two widths: sigma_z, which starts from sz0v >= sy0v and
        // grows at the same rate)
        const float s0 = VEER ? c.sz0v : Sc.sy0v;
        float kyv_m = Sc.kyv, bb = fmaf(-Sc.kyv, Sc.x0v, s0);
        float db = fabsf(Sc.tan_th0 * Sc.x0d) + 2.2f * fabsf(Sc.pj);
        if (__any(split)) {
          const float TImax = fmaxf(TIs[0], fmaxf(TIs[1], TIs[2])), TImin = fminf(TIs[0], fminf(TIs[1], TIs[2]));
          const float x0d_m = x0num_d * frcp(fmaf(c.alpha4_d, TImin, b2om_d));
          const float pj_m = pfac * frcp(fmaf(c.ka_d, TImin, c.kb_d));
          kyv_m = fmaf(c.ka, TImax + X.dTI, c.kb);
          bb = split ? s0 : bb;
          db = fabsf(Sc.tan_th0 * x0d_m) + 2.2f * fabsf(pj_m);
        }
        const float dbo = db + c.off[2];
        float k6 = c.far_k * kyv_m, b6 = fmaf(c.far_k, bb, dbo), n6 = fmaf(c.far_k, fmaxf(Sc.snw, s0), dbo);
        k6 = wave_max<G>(k6);
        b6 = wave_max<G>(b6);
        n6 = wave_max<G>(n6);
        if (!c.far_on) { k6 = 0.0f; b6 = 0.0f; n6 = 3.0e38f; }  // never far
        if (lane == 0) bndL[i] = make_float4(k6, b6, n6, 0.0f);
      }
      WF_T(so_4);
      WF_ACC(9, so_3, so_4);
      // E. pass 2: strictly downstream lanes of the source's own slot, every later slot (an earlier slot's turbines are at
      // or upstream of the source: nothing to do, a tie leaves deficit and TI untouched)
      const float xs_[4] = {X.TI0, X.TI1, X.TI2, X.dTI};
      static_for<S>([&](auto PP) {
        constexpr int p = decltype(PP)::value;
        if constexpr (p == ps) pass2(PP, Sc, xs_, false, ex[p], ex[p].x > 0.0f, yt_d[TAB ? 0 : p], yi_d);
        else if constexpr (p > ps) pass2(PP, Sc, xs_, false, ex[p], tvalid[p], yt_d[TAB ? 0 : p], yi_d);
      });
      WF_T(so_5);
      WF_ACC(10, so_4, so_5);
      __builtin_amdgcn_s_setprio(0);
    };

    for (int cq = 0; cq < n_chunks; ++cq, ++q) {
      if constexpr (TAB) {
        // (a block's first chunk: its hot records were staged behind the previous block's sources, in front of that block's
        // outputs — long landed; waited for here, before this chunk's own transfers are issued)
        if (cq == 0 && first_own > 0) wf_dma_wait();
        if (cq + 1 < n_chunks || J + 1 < nblk) stage_chunk(q + 1);  // lands in the other buffer while this chunk is consumed
        // the next chunk's hot records, when it has logged sources and is a chunk of THIS block (everything it replays was
        // logged by earlier blocks); the first chunk of the NEXT block replays this block's sources too: staged behind them
        if (cq + 1 < n_chunks && (cq + 1) * CH < first_own) stage_hot(q + 1, (cq + 1) * CH);
      }
      const float* buf = &prow[q & 1][0];
      const int i0 = cq * CH;
      const int k_log = min(max(first_own - i0, 0), CH);  // records [0, k_log) of the chunk belong to earlier blocks
      const int k_end = min(n_src - i0, CH);
      // ---- sources of earlier blocks: replayed from the log on this block's targets ----------------------
      // Every logged source applies its transverse velocities (hot record: two circulations); its deflection / deficit /
      // turbulence pass and the cold record that needs are skipped when NO target of the block, in any farm of the wave, is
      // within the source's far bound (bndL, written with the record: own_source).  The two passes touch disjoint state
      // (V, W / deficit sums, TI) and each keeps its own order of sources, so they run as two loops per chunk: the results
      // are bit for bit those of one loop.
      WF_T(st_a);
      auto within_bound = [&](const float4 bnd, float dx, float dy) {  // lin = ad + bd dx: the model's linear deflection offset
        return fabsf(dy) < fmaxf(fmaf(bnd.x, dx, bnd.y), bnd.z) + fabsf(fmaf(c.bd, dx, c.ad));
      };
      if constexpr (TAB) {
        if (k_log > 0) {
          // Which logged sources of this chunk are within reach of the block?  On the table path the farms of a wave share
          // the pair geometry and the bound is the wave's own, so the answer does not depend on the farm: ONE lane per
          // (source, target of the block) pair of the chunk — 64 / GS sources x GS targets = 64 lanes — tests its pair;
          // the ballot, folded to one bit per source (at bit k GS), is the whole chunk's answer.
          unsigned long long near_bits, pair_bits;  // per source (at bit k GS) / per (source, target of the block) pair
          {
            const int kk = lane / GS, rr = lane % GS;
            const float2 e = *reinterpret_cast<const float2*>(buf + (kk * GS + rr) * WF_PAIR_STRIDE + WF_PAIR_DX);
            const float4 bnd = bndL[min(i0 + kk, n_pad - 1)];
            pair_bits = __ballot(kk < k_log && (J * GS + rr) < N && within_bound(bnd, e.x, e.y));
            near_bits = pair_bits;
#pragma unroll
            for (int sh = 1; sh < GS; sh <<= 1) near_bits |= near_bits >> sh;
            unsigned long long rep = 0ull;
#pragma unroll
            for (int kq = 0; kq < CH; ++kq) rep |= 1ull << (kq * GS);
            near_bits &= rep;
          }
          // the first near source's cold record is fetched now and lands while the transverse pass runs
          ColdRec cold_nx = {};
          if (near_bits) cold_nx = load_cold(i0 + (__builtin_ctzll(near_bits) / GS));
          // -- transverse pass: every logged source of the chunk, two per iteration (their hot records are one float4,
          // staged into wave-private LDS by hand-issued LDS-DMA a chunk ahead: the loop holds no global load at all).  Two
          // sources per iteration, ~230 instructions.  No lane mask: every real turbine of this block is at or downstream of an earlier block's
          // source (dx >= 0), and the lanes beyond N (last block only) carry all-zero records. --
          static_assert(GS % 2 == 0 && CH % 2 == 0, "logged sources come in pairs");
#pragma unroll 1
          for (int k = 0; k < k_log; k += 2) {
            const float4 hot2 = hotL[(q & 1) * HOT_F4 + (k >> 1) * EPW + eiw];  // {rho, Gwt} of sources i0 + k, i0 + k + 1
            float4 cf[S][9];
#pragma unroll
            for (int p = 0; p < S; ++p)
#pragma unroll
              for (int qq = 0; qq < 9; ++qq)
                cf[p][qq] = *reinterpret_cast<const float4*>(buf + (k * GS + p * G + sub) * WF_PAIR_STRIDE + 4 * qq);
            static_for<S>([&](auto PP) { apply_tab_ratio(PP, cf[decltype(PP)::value], hot2.x, hot2.y); });
#pragma unroll
            for (int p = 0; p < S; ++p)
#pragma unroll
              for (int qq = 0; qq < 9; ++qq)
                cf[p][qq] = *reinterpret_cast<const float4*>(buf + ((k + 1) * GS + p * G + sub) * WF_PAIR_STRIDE + 4 * qq);
            static_for<S>([&](auto PP) { apply_tab_ratio(PP, cf[decltype(PP)::value], hot2.z, hot2.w); });
          }
          WF_T(st_ab);
          WF_ACC(11, st_a, st_ab);  // (the transverse pass's share of the replay)
          // -- deflection / deficit / turbulence pass: the near sources only; the next one's cold record is fetched while the
          // current one is applied, into the OTHER of two register sets used alternately (a 2 x unrolled loop: one set
          // copied into the other costs a move per float and step — 12 of ~200 instructions) --
          auto near_step = [&](int k, const ColdRec& cold, ColdRec& nxt) {
            const int i = i0 + k;
            float4 exs[S];
#pragma unroll
            for (int p = 0; p < S; ++p)
              exs[p] = *reinterpret_cast<const float4*>(buf + (k * GS + p * G + sub) * WF_PAIR_STRIDE + WF_PAIR_DX);
            asm volatile("" ::: "memory");
            // unconditional (the last near source re-reads its own record, which is in the cache): a conditional load leaves
            // "keeps its value" on the other path, a register copy per float and iteration.  (Issued behind the step's first LDS
            // reads for historical reasons — rounds 2-4's builtin LDS-DMA made the compiler wait for every outstanding load there;
            // with the hand-issued transfers the position no longer matters: measured equal either way, profiles/r05_hotlds_ablation.txt.)
#ifdef WF_EXP_NOCOLD  // timing experiment only (wrong results): the deficit pass without its cold-record loads
            nxt = cold;
#else
            nxt = load_cold(near_bits ? i0 + (__builtin_ctzll(near_bits) / GS) : i);
#endif
            const SrcLog Sl = unpack(make_float2(0.0f, 0.0f), cold);
            const float* side = logx + (size_t)i * WF_LOG_SIDE_FLOATS;
            // ... on the slots that hold a target within the source's bound (round 5: the pair test already knows which; a near
            // source usually reaches ONE of a block's slots — a wake is a row wide, a slot's turbines sit in two rows — and the
            // other slot used to pay the deflection arithmetic (root, reciprocal, logarithm) just to find itself out of reach)
            static_for<S>([&](auto PP) {
              constexpr int p = decltype(PP)::value;
              if ((pair_bits >> (k * GS + p * G)) & ((1ull << G) - 1ull)) pass2(PP, Sl, side, true, exs[p], tvalid[p]);
            });
          };
          ColdRec cold_b = {};
          // (the deficit pass waits for its cold records: ahead of the other wave's transverse pass and outputs, behind its
          // own-source chain — HornsRev1 x 65 536 another -2.4 %, profiles/r05_setprio_ab.txt)
          __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
          while (near_bits) {
            const int k0 = __builtin_ctzll(near_bits) / GS;
            near_bits &= near_bits - 1ull;
            near_step(k0, cold_nx, cold_b);
            if (!near_bits) break;
            const int k1 = __builtin_ctzll(near_bits) / GS;
            near_bits &= near_bits - 1ull;
            near_step(k1, cold_b, cold_nx);
          }
          __builtin_amdgcn_s_setprio(0);
        }
      } else {
        // a wind per farm: the pair geometry comes from the farm's own float64 coordinates (the source's travel with its
        // record), so the far test is on THIS source, after the fact: it saves the pass, not the cold record
#pragma unroll 1
        for (int k = 0; k < k_log; ++k) {
          const int i = i0 + k;
          const float2 hot = hot_nx;       // fetched during the previous iteration
          const ColdRec cold = cold_nx;
          const int inx = min(i + 1, first_own - 1);
          const float* side = logx + (size_t)i * WF_LOG_SIDE_FLOATS;
          const double xs_d = xs_nx, ys_d = ys_nx;
          hot_nx = load_hot(inx);
          cold_nx = load_cold(inx);
          xs_nx = gx[gofs + inx];
          ys_nx = gy[gofs + inx];
          const float4 bnd = bndL[i];
          float4 exr[S];
          int nr = 0;
          static_for<S>([&](auto PP) {
            constexpr int p = decltype(PP)::value;
            exr[p] = fly_record(PP, xs_d, ys_d);
            nr |= (int)tvalid[p] & (int)within_bound(bnd, exr[p].x, exr[p].y);
          });
          const bool near = __any(nr != 0);
          const SrcLog Sl = unpack(hot, cold);
          static_for<S>([&](auto PP) {
            constexpr int p = decltype(PP)::value;
            if (tvalid[p]) apply_fly(PP, exr[p].x, exr[p].y, hot.x * hot.y, hot.y);  // Gy = rho Gwt
            if (near) pass2(PP, Sl, side, true, exr[p], tvalid[p], yt_d[TAB ? 0 : p], ys_d);
          });
        }
      }
      // ---- sources of this block ------------------------------------------------------------------------
      WF_T(st_b);
      WF_ACC(0, st_a, st_b);
#pragma unroll 1
      for (int k = k_log; k < k_end; ++k) {
        const int i = i0 + k;
        const float* recs = buf + (k * GS) * WF_PAIR_STRIDE;
        const int slot_of_source = (i - first_own) / G;
        static_for<S>([&](auto PS) {
          if (slot_of_source == decltype(PS)::value) own_source(PS, i, recs);
        });
      }
      WF_T(st_c);
      WF_ACC(1, st_b, st_c);
      if constexpr (TAB) {
        wf_dma_wait();    // this wave's transfers for the next chunk (pair records, hot records) have landed ...
        __syncthreads();  // ... and so have everyone else's; everyone is done with this chunk
      }
      WF_T(st_d);
      WF_ACC(2, st_c, st_d);
    }
    if constexpr (TAB) {
      // the next block's first chunk replays this block's sources too: its hot records are staged now, behind their log writes
      // (same wave, program order), and land while this block's outputs are computed
      if (J + 1 < nblk) stage_hot(q, 0);
    }
    WF_T(st_e);

    // ---- outputs [A.4] of block J --------------------------------------------------------------
#pragma unroll
    for (int p = 0; p < S; ++p) {
      if (!tvalid[p]) continue;
      const int o = oidx[p];
      float U[9], m3 = 0.0f, mu = 0.0f, mv = 0.0f, mw = 0.0f, adir = 0.0f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if constexpr (VEER) {
#pragma unroll
          for (int k = 0; k < 3; ++k) U[3 * j + k] = Ui[k] * (1.0f - fsqrt(esq[p][3 * j + k]));
        } else {
          const float ue = 1.0f - fsqrt(esq[p][2 * j]), uc = 1.0f - fsqrt(esq[p][2 * j + 1]);
          U[3 * j] = Ui[0] * ue; U[3 * j + 1] = Ui[1] * uc; U[3 * j + 2] = Ui[2] * ue;
        }
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        m3 = fmaf(U[k] * U[k], U[k], m3);
        mu += U[k]; mv += V[p][k]; mw += W[p][k];
      }
      if (o_wd) {
        bool small = true;
        float rr[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          rr[k] = V[p][k] * frcp(U[k]);
          small = small && (U[k] > 0.0f) && (fabsf(rr[k]) <= 0.25f);
        }
        if (__all(small)) {
#pragma unroll
          for (int k = 0; k < 9; ++k) adir += atan_small(rr[k]);
        } else {
#pragma unroll
          for (int k = 0; k < 9; ++k) adir += atan2f(V[p][k], U[k]);
        }
      }
      mu *= (1.0f / 9.0f); mv *= (1.0f / 9.0f); mw *= (1.0f / 9.0f);
      float su = 0.0f, sv = 0.0f, sw = 0.0f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float du = U[k] - mu, dv = V[p][k] - mv, dw = W[p][k] - mw;
        su = fmaf(du, du, su); sv = fmaf(dv, dv, sv); sw = fmaf(dw, dw, sw);
      }
      const float wsp = fcbrt_pos(m3 * (1.0f / 9.0f));
      {  // (a rotor-grid speed that is not positive: WF_RISK_NEGATIVE_SPEED, see wf_kernels.hip)
        float umin = U[0];
#pragma unroll
        for (int k = 1; k < 9; ++k) umin = fminf(umin, U[k]);
        if (!(umin > 0.0f)) atomicOr(&risk_lds[wave][eiw], (unsigned)WF_RISK_NEGATIVE_SPEED);
      }
      const float veff = c.dens_f * wsp * fexp2(c.pw * flog2(cg_t[p]));
      float pslope;
      const float pwr = c.rho * table_pw(c, T, veff, pslope);
      if (c.rho * fabsf(pslope) * veff > c.knee_kappa * fmaxf(pwr, 1.0e3f))
        atomicOr(&risk_lds[wave][eiw], (unsigned)WF_RISK_POWER_KNEE);
      float4 l;
      l.x = (TI[p][0] + TI[p][1] + TI[p][2]) * (1.0f / 3.0f);
      l.y = fsqrt(su * (1.0f / 9.0f));
      l.z = fsqrt(sv * (1.0f / 9.0f));
      l.w = fsqrt(sw * (1.0f / 9.0f));
      const bool real = o < n_real;  // (a placeholder of a padded layout: zeros out, nothing into the reward)
      psum += real ? pwr : 0.0f;
      lsum += real ? (l.x + l.y) + (l.z + l.w) : 0.0f;
      if (env_ok) {
        const size_t oo = yofs + o;
        if (o_power) o_power[oo] = real ? (ea.power_mw ? pwr * 1.0e-6f : pwr) : 0.0f;
        if (o_ws) o_ws[oo] = real ? wsp : 0.0f;
        if (o_wd) o_wd[oo] = real ? wd - adir * (kRad2Deg / 9.0f) : 0.0f;
        if (o_load) reinterpret_cast<float4*>(o_load)[oo] = real ? l : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
    }
    WF_T(st_f);
    WF_ACC(3, st_e, st_f);
  }  // J
#ifdef WF_LL_STAMP
  if (lane == 0) {
    st_acc[4] = __builtin_readcyclecounter() - st_begin;
    for (int k = 0; k < 5; ++k) atomicAdd(&wf_ll_stamp[k], st_acc[k]);
    atomicAdd(&wf_ll_stamp[5], 1ull);
    for (int k = 6; k < 12; ++k) atomicAdd(&wf_ll_stamp[k], st_acc[k]);  // parts of an own-source step; 11: transverse pass of the replay (tools/ll_stamps.py)
  }
#endif

  if (ga.risk_flags && sub == 0 && env_ok) {
    const int rf = (int)risk_lds[wave][eiw];
    ga.risk_flags[env] = rf;
    if (ga.res_list) {  // the float64 re-solve's work list (wf_device.h: WfGroupArgs)
      ga.flags_raw[env] = rf;
      if (rf & ga.res_mask) ga.res_list[atomicAdd(ga.res_count, 1)] = env;
    }
  }
  if (ea.reward) {
#pragma unroll
    for (int w = G / 2; w >= 1; w >>= 1) {
      psum += __shfl_xor(psum, w);
      lsum += __shfl_xor(lsum, w);
    }
    if (sub == 0 && env_ok) {
      const float invN = __fdiv_rn(1.0f, (float)n_real);
      const float wr = ea.ws_prev ? (float)ea.ws_prev[env] : ws;
      const float r = psum * invN * 1.0e-3f * frcp(wr * wr * wr) - ea.load_coef * lsum * invN * 0.25f;
      ea.reward[env] = r;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Launch
// ---------------------------------------------------------------------------------------------
constexpr int kLLWaves = 4;

template <int G, int S, bool TAB, bool MC1, bool VEER = false>
static hipError_t launch_ll(const WfConsts* c, const WfTables* tab, const int* gidx, const double* ws, const double* wd,
                            int wind_stride, const float* yaw, float* power, float* o_ws, float* o_wd, float* load, int B,
                            const WfEnvArgs* env, const float* ll_tab, const int* cross_tie, float* src_log, size_t log_records,
                            const WfGroupArgs* grp, const double* gx, const double* gy, hipStream_t s) {
  constexpr int fpb = kLLWaves * (64 / G);
  WfGroupArgs ga = *grp;
  const int n_farms = ga.env_end ? ga.env_end - ga.env_base : B;  // (a mixed launch serves a range of the batch)
  const int grid = ga.blk_group ? (ga.n_slots + fpb - 1) / fpb : (n_farms + fpb - 1) / fpb;
  WfConsts cc = *c;
  WfEnvArgs ea;
  if (env) ea = *env; else memset(&ea, 0, sizeof(ea));
  size_t group_floats = wfk_ll_table_floats(cc.N, G * S);
  int n_pad = ((cc.N + G * S - 1) / (G * S)) * (G * S);
  // dynamic LDS: the far bounds of every wave (16 bytes per source)
  // ... then (table path) two buffers per wave for the hot records of a chunk's logged sources (stage_hot)
  constexpr int hot_bytes = 32768 / (G * G * S) >= 1024 ? 32768 / (G * G * S) : 1024;  // (CH / 2) x EPW x 16 B, whole 1-KiB DMA instructions
  const size_t dyn_lds = sizeof(float4) * (size_t)kLLWaves * n_pad + (TAB ? (size_t)kLLWaves * 2 * hot_bytes : 0);
  // the log allocation holds log_records (farm slot, source) records: hot part, cold part, side records (wf_device.h)
  size_t log_cold_offset = log_records * WF_LOG_HOT_FLOATS, log_side_offset = log_records * WF_LOG_FLOATS;
  void* args[] = {&cc, &tab, &gidx, &ws, &wd, &wind_stride, &yaw, &power, &o_ws, &o_wd, &load, &B, &ea, &ll_tab,
                  &group_floats, &cross_tie, &src_log, &log_cold_offset, &log_side_offset, &n_pad, &ga, &gx, &gy};
  const void* fn = (const void*)&wf_step_ll_kernel<G, S, false, TAB, MC1, kLLWaves, VEER>;
  if constexpr (TAB) {
    if (wind_stride == 0) fn = (const void*)&wf_step_ll_kernel<G, S, true, TAB, MC1, kLLWaves, VEER>;
  }
  if constexpr (S == 1 && (MC1 || !TAB) && !VEER) {  // no third block per CU in this launch: the spill-free two-wave build
    // CU count of the CURRENT device (the handle's: launch_step_f32 runs under WF_ON_DEVICE), cached per device — not once per
    // instantiation from whichever device launched first: wf_get_kernel_info decides with the handle's own count
    static int n_cu_of[64] = {};
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
      n_cu = n_cu_of[dev];
      if (n_cu == 0 && hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) n_cu_of[dev] = n_cu;
    }
    if (n_cu > 0 && grid <= 2 * n_cu)
      fn = wind_stride == 0 ? (const void*)&wf_step_ll_kernel<G, S, true, TAB, MC1, kLLWaves, VEER, true>
                            : (const void*)&wf_step_ll_kernel<G, S, false, TAB, MC1, kLLWaves, VEER, true>;
  }
  return hipLaunchKernel(fn, dim3(grid), dim3(64 * kLLWaves), args, dyn_lds, s);
}

extern "C" int wfk_ll_farms_per_block(int G) { return kLLWaves * (64 / G); }
#ifdef WF_LL_STAMP
extern "C" int wfk_ll_stamps(unsigned long long* out, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(wf_ll_stamp), sizeof(wf_ll_stamp));
  if (e == hipSuccess && reset) {
    unsigned long long z[16] = {};
    e = hipMemcpyToSymbol(HIP_SYMBOL(wf_ll_stamp), z, sizeof(z));
  }
  return (int)e;
}
#endif

// (G, S) instantiations of the table path: one slot per lane at every width, two slots at G = 4 (eight turbines per block
// like G = 8, S = 1, with sixteen instead of eight farms per wave sharing the per-source phase).  On the fly (a wind
// per farm): the two variants the rounds model ever picks, each with and without the mirror-core shortcut.
#define WF_LL_DISPATCH(G_, S_, CALL) \
  if (G == G_ && S == S_) return CALL(G_, S_)
extern "C" hipError_t wfk_launch_step_ll(int G, int S, const WfConsts* c, const WfTables* tab, const int* gidx, const double* ws,
                                         const double* wd, int wind_stride, const float* yaw, float* power, float* o_ws,
                                         float* o_wd, float* load, int B, const WfEnvArgs* env, const float* ll_tab,
                                         const int* cross_tie, float* src_log, size_t log_records,
                                         const WfGroupArgs* grp, hipStream_t s) {
#define WF_LL_LAUNCH(G_, S_) launch_ll<G_, S_, true, true>(c, tab, gidx, ws, wd, wind_stride, yaw, power, o_ws, o_wd, load, B, env, ll_tab, cross_tie, src_log, log_records, grp, nullptr, nullptr, s)
#define WF_LL_LAUNCH_VEER(G_, S_) launch_ll<G_, S_, true, true, true>(c, tab, gidx, ws, wd, wind_stride, yaw, power, o_ws, o_wd, load, B, env, ll_tab, cross_tie, src_log, log_records, grp, nullptr, nullptr, s)
  if (c->veer_on) {  // wind veer: the throughput families only (wfk_ll_has_veer)
    WF_LL_DISPATCH(4, 1, WF_LL_LAUNCH_VEER);
    WF_LL_DISPATCH(4, 2, WF_LL_LAUNCH_VEER);
    WF_LL_DISPATCH(2, 2, WF_LL_LAUNCH_VEER);
    return hipErrorInvalidValue;
  }
  WF_LL_DISPATCH(4, 1, WF_LL_LAUNCH);
  WF_LL_DISPATCH(8, 1, WF_LL_LAUNCH);
  WF_LL_DISPATCH(16, 1, WF_LL_LAUNCH);
  WF_LL_DISPATCH(4, 2, WF_LL_LAUNCH);
  WF_LL_DISPATCH(2, 2, WF_LL_LAUNCH);
  return hipErrorInvalidValue;
}

// (G, S) shapes instantiated with wind veer: table path 4x1, 4x2, 2x2; on the fly 4x2
extern "C" int wfk_ll_has_veer(int G, int S, int table) { return table ? ((G == 4 && S <= 2) || (G == 2 && S == 2)) : (G == 4 && S == 2); }

extern "C" int wfk_ll_has_fly(int G, int S) { return (G == 4 && S <= 2) || (G == 8 && S == 1) || (G == 2 && S == 2); }

// a wind per farm: gx / gy [B][N] sorted coordinates of every farm, farm_tie [B] the per-farm cross-block-tie flags
extern "C" hipError_t wfk_launch_step_ll_fly(int G, int S, const WfConsts* c, const WfTables* tab, const int* gidx, const double* gx,
                                             const double* gy, const double* ws, const double* wd, const float* yaw,
                                             float* power, float* o_ws, float* o_wd, float* load, int B, const WfEnvArgs* env,
                                             const int* farm_tie, float* src_log, size_t log_records,
                                             const WfGroupArgs* grp, hipStream_t s) {
#define WF_LL_LAUNCH_FLY(G_, S_)                                                                                              \
  (c->mirror_core_n <= 1                                                                                                      \
       ? launch_ll<G_, S_, false, true>(c, tab, gidx, ws, wd, 1, yaw, power, o_ws, o_wd, load, B, env, nullptr, farm_tie, src_log, \
                                        log_records, grp, gx, gy, s)                                                       \
       : launch_ll<G_, S_, false, false>(c, tab, gidx, ws, wd, 1, yaw, power, o_ws, o_wd, load, B, env, nullptr, farm_tie,      \
                                         src_log, log_records, grp, gx, gy, s))
  if (c->veer_on) {
    if (G == 4 && S == 2)
      return launch_ll<4, 2, false, false, true>(c, tab, gidx, ws, wd, 1, yaw, power, o_ws, o_wd, load, B, env, nullptr, farm_tie, src_log,
                                                 log_records, grp, gx, gy, s);
    return hipErrorInvalidValue;
  }
  WF_LL_DISPATCH(4, 2, WF_LL_LAUNCH_FLY);
  WF_LL_DISPATCH(2, 2, WF_LL_LAUNCH_FLY);
  WF_LL_DISPATCH(4, 1, WF_LL_LAUNCH_FLY);
  WF_LL_DISPATCH(8, 1, WF_LL_LAUNCH_FLY);
  return hipErrorInvalidValue;
}

// (occ2: the launch reaches no third block per CU — launch_ll then takes the two-wave build of a one-slot family)
extern "C" hipError_t wfk_ll_func_attributes(int G, int S, int shared_speed, int table, int veer, int occ2, hipFuncAttributes* a) {
#define WF_LL_ATTR_VEER(G_, S_) hipFuncGetAttributes(a, shared_speed ? (const void*)&wf_step_ll_kernel<G_, S_, true, true, true, kLLWaves, true> : (const void*)&wf_step_ll_kernel<G_, S_, false, true, true, kLLWaves, true>)
  if (veer) {
    if (!table) return (G == 4 && S == 2) ? hipFuncGetAttributes(a, (const void*)&wf_step_ll_kernel<4, 2, false, false, true, kLLWaves, true>) : hipErrorInvalidValue;
    WF_LL_DISPATCH(4, 1, WF_LL_ATTR_VEER);
    WF_LL_DISPATCH(4, 2, WF_LL_ATTR_VEER);
    WF_LL_DISPATCH(2, 2, WF_LL_ATTR_VEER);
    return hipErrorInvalidValue;
  }
#define WF_LL_ATTR(G_, S_) hipFuncGetAttributes(a, shared_speed ? (const void*)&wf_step_ll_kernel<G_, S_, true, true, true, kLLWaves> : (const void*)&wf_step_ll_kernel<G_, S_, false, true, true, kLLWaves>)
#define WF_LL_ATTR_FLY(G_, S_) hipFuncGetAttributes(a, (const void*)&wf_step_ll_kernel<G_, S_, false, false, true, kLLWaves>)
#define WF_LL_ATTR_FLY_OCC2(G_, S_) hipFuncGetAttributes(a, (const void*)&wf_step_ll_kernel<G_, S_, false, false, true, kLLWaves, false, true>)
  if (!table) {
    if (occ2) {
      WF_LL_DISPATCH(4, 1, WF_LL_ATTR_FLY_OCC2);
      WF_LL_DISPATCH(8, 1, WF_LL_ATTR_FLY_OCC2);
    }
    WF_LL_DISPATCH(4, 2, WF_LL_ATTR_FLY);
    WF_LL_DISPATCH(2, 2, WF_LL_ATTR_FLY);
    WF_LL_DISPATCH(4, 1, WF_LL_ATTR_FLY);
    WF_LL_DISPATCH(8, 1, WF_LL_ATTR_FLY);
    return hipErrorInvalidValue;
  }
#define WF_LL_ATTR_OCC2(G_, S_) hipFuncGetAttributes(a, shared_speed ? (const void*)&wf_step_ll_kernel<G_, S_, true, true, true, kLLWaves, false, true> : (const void*)&wf_step_ll_kernel<G_, S_, false, true, true, kLLWaves, false, true>)
  if (occ2) {
    WF_LL_DISPATCH(4, 1, WF_LL_ATTR_OCC2);
    WF_LL_DISPATCH(8, 1, WF_LL_ATTR_OCC2);
    WF_LL_DISPATCH(16, 1, WF_LL_ATTR_OCC2);
  }
  WF_LL_DISPATCH(4, 1, WF_LL_ATTR);
  WF_LL_DISPATCH(8, 1, WF_LL_ATTR);
  WF_LL_DISPATCH(16, 1, WF_LL_ATTR);
  WF_LL_DISPATCH(4, 2, WF_LL_ATTR);
  WF_LL_DISPATCH(2, 2, WF_LL_ATTR);
  return hipErrorInvalidValue;
}
