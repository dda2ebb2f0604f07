// wf_resolve.hip — float64 farm solve on the device for the farms the float32 kernels flag (include/wfstep.h:
// wf_set_risk_resolve); mode 2 solves every farm (validation).
//
// The reference evaluates the whole path in float64 (reference wfcrl/interface.py:564 `fi.calculate_wake`, FLORIS 3.5
// sequential solver; SURVEY.md Appendix A, tags [A.x] below).  The float32 step kernels cannot reproduce a float64
// decision inside their own rounding band (the overlap count "deficit * Uinit > 0.05" [A.3-8]) nor the result on an
// ill-conditioned piece of the turbine tables; they raise a per-farm risk flag there.  This file makes the 1e-4 contract
// unconditional: the flagged farms are compacted on the device (no host round trip) and solved again in float64, the
// comparison taken exactly as FLORIS takes it, and their outputs overwritten.
//
// Two kernels (wfk_launch_resolve: the four-wave kernel behind every step of modes 0 / 1, whatever the list's length; the one-wave
// kernel for mode 2 — every farm — at batches beyond one residency of the other):
//   wf_resolve4_kernel  one farm per block: four waves with roles, up to four such blocks per CU — or, on a short list (the
//                       re-solve is then ONE farm's latency), a 512-thread launch whose four more HELPER waves share the pair
//                       passes of the level stages.  The farm's state — per sorted turbine 9 sums of squared deficits, 9 V, 9 W,
//                       3 column TIs, float64 — lives in LDS, turbine-major; a lane is not tied to a turbine.  A SEQUENTIAL
//                       stage solves one source: waves 0-2 take a rotor-grid column each (transverse pass; deflection / deficit
//                       pass; turbulence pass — three block barriers), wave 3 the source-only chain of steering, deflection and
//                       deficit constants beside the transverse pass, and the NEXT source's rotor speed, thrust and circulations
//                       beside the other two passes (speculated from the deficit sums, confirmed by a bit comparison).  A LEVEL
//                       stage (round 6: Lvl4Shared) solves three to eight consecutive sources that put no deficit on each
//                       other at once, every (source, target, column) pair in its own lane, the sums still taken in source
//                       order; its phases hand over through LDS counters and flags, two block barriers per stage.
//   wf_resolve_kernel   one farm per WAVE (64-thread blocks, one per SIMD), the same state in LDS, no block barrier inside the
//                       solve, no levels: a third more farms per CU and second than the four-wave kernel's SEQUENTIAL stages, at
//                       2.2 x their latency (rounds 3-5 and the first half of round 6: the kernel for lists beyond one residency).
// Per-source constants are derived once per farm and handed to the pair passes through LDS; every phase of a stage starts
// behind a compiler barrier and is free of calls and spills (no private segment: tests/test_abi.py).
// Only exactness-preserving algebra is used (vortex core 1 - exp(-(y^2+z^2)/eps^2) with the z factor a constant);
// sums are taken in a different order than NumPy takes them: results agree with the CPU oracle to ~1e-13 relative.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>

#include "wf_device.h"
#include "wf_resolve.h"
#include "wf_f64_math.h"

// Several turbine definitions per farm (include/wfstep.h: wf_set_turbine_types): this file is compiled a second time with
// RES_MT = 1 (wf_resolve_mt.hip) into kernels that take a turbine's thrust / power table, TSR, pP and reference density from
// the definition the turbine belongs to.  The definitions share the rotor (diameter, hub height), so the geometry constants
// stay per handle; what differs enters at two places only — the source's thrust coefficient and wake-rotation circulation,
// and the turbine's power.  The build without definitions (RES_MT = 0) is the code it always was: the macros below expand
// to the single table.
#ifndef RES_MT
#define RES_MT 0
#endif
// The file is compiled in two PARTS (round 6): 1 = the one-wave-per-farm kernel, the list kernel and the launch logic, under
// the default flags; 2 = the four-wave kernel (wf_resolve4.hip, wf_resolve4_mt.hip) WITHOUT machine LICM — hoisted out of the
// stage loop, the ~50 float64 literals of its phases held a register pair each across the whole solve (256 VGPRs, 60-90
// spilled SGPRs, and with the level stages a private segment); rematerialised where they are used the kernel needs 128.  The
// one-wave kernel loses 10-30 % without the hoisting (profiles/r06_levels_ab.txt), hence two translation units.
#ifndef RES_PART
#define RES_PART 1
#endif
#define WF_RES_GRID_PER_CU 4  // persistent one-wave blocks per CU: one per SIMD (WF_RES_OCC)
#if RES_MT
#define RES_NT WF_MAX_TYPES
#define wf_list_all_kernel wf_list_all_mt_kernel
#define wf_resolve_kernel wf_resolve_mt_kernel
#define wf_resolve4_kernel wf_resolve4_mt_kernel
#define wfk_launch_resolve wfk_launch_resolve_mt
#if RES_PART == 2
#define wfk_launch_resolve4 wfk_launch_resolve4_mt
#endif
#define RES_TY(t) (reinterpret_cast<const int*>(res_dyn + (t) * RES_TS + 35)[0])  // (the record's first padding word)
#define RES_TOFS(ty) ((ty) * WF_TABLE_PAD)
#define RES_TN(S, ty) ((S).ty_n[ty])
#define RES_TC(S, k, ty, single) ((S).ty_c[ty][k])
#else
#define RES_NT 1
#define RES_TY(t) 0
#define RES_TOFS(ty) 0
#define RES_TN(S, ty) ((S).c.n_table)
#define RES_TC(S, k, ty, single) (single)
#endif

namespace {



// scipy interp1d(linear, bounds_error=False, fill_value=(lo, hi)) on the LDS copy of a table column; sl: the segment slopes
__device__ inline double interp_fill(double xq, int n, const double* xs, const double* ys, const double* sl, double lo, double hi) {
  if (xq < xs[0]) return lo;
  if (xq > xs[n - 1]) return hi;
  if (xq == xs[n - 1]) return ys[n - 1];
  int j = 0;  // last knot <= xq, at most n - 2: bisection (the dependent LDS probes of a linear scan cost 5 k cycles per call)
  for (int step = 32; step >= 1; step >>= 1) {
    const int k = j + step;
    if (k <= n - 2 && xq >= xs[k]) j = k;
  }
  return sl[j] * (xq - xs[j]) + ys[j];
}

// the same for a query that is uniform over the wave (the source's rotor wind speed): every lane tests one knot, the
// segment index is the population count of the ballot — one LDS read instead of six dependent ones
__device__ inline double interp_fill_uniform(double xq, int n, const double* xs, const double* ys, const double* sl, double lo, double hi) {
  const int l = threadIdx.x & 63;
  const unsigned long long m = __ballot(l < n && xq >= xs[l < n ? l : 0]);
  if (xq < xs[0]) return lo;
  if (xq > xs[n - 1]) return hi;
  if (xq == xs[n - 1]) return ys[n - 1];
  int j = __popcll(m) - 1;
  j = j < 0 ? 0 : (j > n - 2 ? n - 2 : j);
  return sl[j] * (xq - xs[j]) + ys[j];
}

}  // namespace

#if RES_PART == 1
// every farm -> list, count = B; raw = copy of the flags as the float32 kernels raised them (mode 2: the flagged-farms list of
// modes 0 / 1 is written by the step kernels themselves, wf_device.h: WfGroupArgs::res_list)
__global__ void wf_list_all_kernel(const int* __restrict__ flags, int B, int* __restrict__ list, int* __restrict__ count,
                                   int* __restrict__ raw) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b == 0) *count = B;
  if (b >= B) return;
  raw[b] = flags[b];
  list[b] = b;
}

#endif  // RES_PART == 1
// The kernels below contain NO function call and no register spill: a kernel with a private segment (a stack for out-of-line
// library routines, callee-saved register saves, spills) costs ~20 us per LAUNCH on this chip against ~2.6 us without one
// (tools/ubench/scratch_switch.hip, profiles/r05_scratch_switch.txt) — two such launches behind every step were the whole
// price of the re-solve on small farms.  So the rare out-of-range arguments (a yaw beyond 45 deg, a steering term beyond 0.3,
// velocities driven negative by an unphysically tight layout) are served by inline general-range routines built on the lean
// ones, chosen wave-uniformly.
// A phase of a source step starts behind a compiler barrier: with the phases inlined into one loop body, the model constants
// and the per-source record (all in LDS) would otherwise be loaded once, ahead of the loop, and held in registers across every
// phase — ~370 live values
#ifndef RES_SCHED_LIMIT
#define RES_SCHED_LIMIT 0  // 1: the 14 reciprocal chains of a rotor-grid column are interleaved two at a time instead of all at once (30 registers
                           // less; 3 % slower: profiles/r05_resolve_ab.txt)
#endif
#define RES_PHASE_FENCE asm volatile("" ::: "memory")
// ... and with its own opaque copy of the thread index: everything derived from it (wave, lane, a dozen LDS addresses per phase)
// is recomputed where it is used instead of being hoisted out of the source loop and held — or spilled — across it
__device__ __forceinline__ int res_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}
// a literal the compiler must materialise where it is used: hoisted out of the stage loop, every float64 literal of every
// phase holds a register pair across the whole solve (~50 pairs), and the allocator spills them instead of rematerialising
__device__ __forceinline__ double res_lit(double x) {
  asm volatile("" : "+v"(x));
  return x;
}
#define LOG_F64(x) log_any(x)
#define POW_F64(x, y) pow_any(x, y)

// which phase functions are real functions (A/B builds: -DRES_PASS_INLINE=1 / -DRES_SRC_INLINE=1)
#ifndef RES_PASS_INLINE
#define RES_PASS_INLINE 1
#endif
#ifndef RES_SRC_INLINE
#define RES_SRC_INLINE 1
#endif
#if RES_PASS_INLINE
#define RES_PASS_FN __device__ __forceinline__
#else
#define RES_PASS_FN __device__ __noinline__
#endif



// One farm per WAVE (64-thread blocks).  The farm's state — per turbine 9 sums of squared deficits, 9 V, 9 W, 3 column
// TIs, float64 — lives in LDS, structure-of-arrays over the sorted turbine index; a lane is not tied to a turbine:
// for source i the lanes take the turbines the source can reach, t = first + lane (first = the start of the source's x'
// tie group for the transverse velocities [A.3-4], i + 1 for deficit and wake-added turbulence), 64 at a time, so the
// triangle of the recurrence costs 96 instead of 160 wave passes at N = 80 and nothing is done twice: the per-source
// constants are derived once per farm (by every lane of the one wave).  No block-level barrier inside the solve: the
// LDS operations of a wave execute in order.
// The phases of a source step talk through LDS only (the per-source constants too), with a compiler barrier at every
// phase boundary: written as one body with the constants in registers the allocator kept ~370 values live (the
// per-source constants across both passes, the device library's temporaries) and spilled inside the source loop at any
// occupancy above one wave per SIMD; as real (non-inlined) functions the calling convention's register saves cost 180
// scratch accesses per source.
// (First version, in the history: a thread per turbine, state in registers, two __syncthreads per source, every wave
// re-deriving the source constants: 1.95 ms for 1394 HornsRev1 farms; the inlined one-wave version 1.42 ms; this one 1.37 ms.
// The four-wave kernel further down has the lower latency per farm (0.73 against 1.04 ms) but takes four wave slots per
// farm: this kernel serves the counts beyond half a residency of that one.)
#ifndef WF_RES_OCC
// ONE wave per SIMD (round 5; rounds 3-4: two).  The kernel is a single wave's dependent float64 chain: a second wave on the
// SIMD buys ~1.3 x, not 2 x, and the hardware does not spread 64-thread blocks evenly — with 813 farms on 1024 SIMDs some SIMDs
// got two farms and the launch took their time: 1.08 ms against 0.70 with one wave per SIMD enforced (profiles/r05_resolve_ab.txt).
// Beyond 4 farms per CU the persistent blocks take their next farm: 3 rounds of 0.8 ms for 3000 farms against 3.9 ms before.
#define WF_RES_OCC 1
#endif
#if RES_PART == 1
struct SrcShared {  // what a source leaves for the two passes over its targets
  double x_i, y_i, ct, ai, ubar, Vmean, TIs[3], dTI;
  double Gt, Gb, Gw;  // circulations / (2 pi): top, bottom, wake rotation (commanded yaw)
  double cgd, s_cc, s_c, th0, tan_th0, M0, E0, sM, sz0d, sy0d, is0d, lnAB, sz0v, sy0v, snw, kdef, ch_pref, cgv;
  int same, first_tv;
};
struct ResShared {
  WfResolveConsts c;
  double tws[RES_NT * WF_TABLE_PAD], tct[RES_NT * WF_TABLE_PAD], tpw[RES_NT * WF_TABLE_PAD];
  double tcs[RES_NT * WF_TABLE_PAD], tps[RES_NT * WF_TABLE_PAD];  // segment slopes of the thrust / power columns (one division per segment and launch)
#if RES_MT
  double ty_c[RES_NT][WF_TYPE_CONSTS];  // per definition: 1 / TSR, pP / 3, (air / ref density)^(1/3), ref density
  int ty_n[RES_NT];                     // table entries
#endif
  double ws, wd, Uinf, Uinit[3];
  double dec_a[3];  // 4 nu_k ws / Uinf: decay_k = eps^2 / (dec_a[k] dx + eps^2)  [A.3-4]
  int N, n_pad, veer_on, mcore;
  WfResolveArgs a;  // the launch arguments (read from here inside the farm loop: a dozen pointers less to keep in registers)
  SrcShared s;
};
__shared__ ResShared R;
#endif  // RES_PART == 1
extern __shared__ double res_dyn[];  // per sorted turbine: x', y', cos / sin / radians of the commanded yaw, the 30 state values; then the int arrays
// Turbine-major (round 5; rounds 3-4: one array per quantity at stride n_pad): a quantity of turbine t is at a CONSTANT offset
// from t's record, so a phase's dozens of state addresses are one base plus immediates — with the run-time stride the compiler
// hoisted every (5 + q) n_pad product out of the source loop and spilled them.  The record is 35 doubles padded to 38 = 304
// bytes: a multiple of 16 (the compiler merges neighbouring doubles into ds_read / ds_write_b128, which this runtime serves
// only at 16-byte alignment — a 280-byte record put every odd turbine's merged accesses 8 bytes off) and 76 words = 12 banks
// mod 64: sixteen consecutive turbines cover 32 banks with their 8-byte accesses, the full LDS rate.
#define RES_TS 38

#define RES_XS(t) res_dyn[(t) * RES_TS]
#define RES_YS(t) res_dyn[(t) * RES_TS + 1]
#define RES_CG(t) res_dyn[(t) * RES_TS + 2]
#define RES_SG(t) res_dyn[(t) * RES_TS + 3]
#define RES_GR(t) res_dyn[(t) * RES_TS + 4]
#define RES_ST(q, t) res_dyn[(t) * RES_TS + 5 + (q)]  // wake2 q = 0..8, V 9..17, W 18..26, TI 27..29
#define RES_TIE(t) (reinterpret_cast<int*>(res_dyn + RES_TS * R.n_pad)[(t)])

#if RES_SRC_INLINE
#define RES_SRC_FN __device__ __forceinline__
#else
#define RES_SRC_FN __device__ __noinline__
#endif
#if RES_PART == 1
// ---- the source's state and circulations [A.3-1, A.3-2, A.3-4] ----
RES_SRC_FN void res_source_begin(int i) {
  RES_PHASE_FENCE;
  const WfResolveConsts& c = R.c;
  const double cg = RES_CG(i), sg = RES_SG(i);
  double m3 = 0.0, vs = 0.0;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const double u = R.Uinit[q % 3] - sqrt_nn(RES_ST(q, i));
    m3 += u * u * u;
    vs += RES_ST(9 + q, i);
  }
  const double m3m = m3 * (1.0 / 9.0);
  const double ubar = __any(!(m3m > 1.0e-6)) ? cbrt_any(m3m) : cbrt_pos(m3m);
  [[maybe_unused]] const int ty = RES_TY(i);
  double ct_tab = interp_fill_uniform(ubar, RES_TN(R, ty), R.tws + RES_TOFS(ty), R.tct + RES_TOFS(ty), R.tcs + RES_TOFS(ty), 0.0001, 0.9999);
  ct_tab = fmin(fmax(ct_tab, 0.0001), 0.9999);
  const double ct = ct_tab * cg;
  const double ai = 0.5 * rcp64(cg) * (1.0 - sqrt_nn(1.0 - ct * cg));
  const double G_wr = (0.25 * kTwoPi) * c.D * (ai - ai * ai) * ubar * RES_TC(R, 0, ty, c.inv_TSR);
  const double gam_top = (kTwoPi / 16.0) * c.D * c.vel_top * R.Uinf * ct;
  const double gam_bot = (kTwoPi / 16.0) * c.D * c.vel_bot * R.Uinf * ct;
  const double sc = sg * cg;
  if (res_tid() == 0) {
    SrcShared& s = R.s;
    s.x_i = RES_XS(i); s.y_i = RES_YS(i); s.ct = ct; s.ai = ai; s.ubar = ubar; s.Vmean = vs * (1.0 / 9.0);
    s.TIs[0] = RES_ST(27, i); s.TIs[1] = RES_ST(28, i); s.TIs[2] = RES_ST(29, i);
    s.Gt = sc * gam_top * (1.0 / kTwoPi); s.Gb = -sc * gam_bot * (1.0 / kTwoPi); s.Gw = G_wr * (1.0 / kTwoPi);
    s.first_tv = RES_TIE(i);
    // secondary steering [A.3-2]: the three means on the source's own grid are geometry constants
    const double v_top = gam_top * c.k_top, v_bot = -gam_bot * c.k_bot, v_core = G_wr * c.k_core;
    s.cgv = 2.0 * (s.Vmean - v_core) * rcp64(v_top + v_bot);  // (val: parked here until res_source_finish overwrites it)
  }
}

// ---- 4. transverse velocities (commanded yaw) on every turbine at or downstream of the source, ties included; per grid
// column the 7 + 7 distinct vertical offsets of the three vortices and their ground mirrors ----
RES_PASS_FN void res_transverse_pass() {
  RES_PHASE_FENCE;
  const WfResolveConsts& c = R.c;
  const int lane = res_tid(), N = R.N;
  const double x_i = R.s.x_i, y_i = R.s.y_i, Gt = R.s.Gt, Gb = R.s.Gb, Gw = R.s.Gw;
  const double qd = c.off[2], neps = c.num_eps, twoHH = 2.0 * c.HH, eps2 = c.eps2, ieps2 = c.inv_eps2;
  const bool mcore = R.mcore != 0;
  for (int base = R.s.first_tv; base < N; base += 64) {
    const int t = base + lane;
    if (t >= N) continue;
    const double dx = RES_XS(t) - x_i, y_t = RES_YS(t);
    double dec[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) dec[k] = eps2 * rcp64(fma(R.dec_a[k], dx, eps2));
#pragma unroll 1
    for (int j = 0; j < 3; ++j) {  // (a real loop: the state is addressed in LDS, nothing needs a static index; the three columns interleaved gained nothing)
      double Vj[3], Wj[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) { Vj[k] = RES_ST(9 + j * 3 + k, t); Wj[k] = RES_ST(18 + j * 3 + k, t); }
      const double yL = (y_t + c.off[j] - y_i) + neps;
      const double yL2 = yL * yL;
      const double Ey = exp_lean(-yL2 * ieps2);
      double Av[3] = {0.0, 0.0, 0.0}, Bw[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int m = 0; m < 7; ++m) {
        const double zc = (double)(m - 3) * qd + neps, zm = zc + twoHH;
        const double tr = (1.0 - Ey * c.ezc[m]) * rcp64(yL2 + zc * zc);   // core / r of a real vortex at offset zc
        double tm = rcp64(yL2 + zm * zm);                                   // ... of a mirror vortex at zm
        if (mcore) tm *= 1.0 - Ey * c.ezm7[m];  // (1 - Ey ezm == 1.0 exactly unless the hub is very low)
        const double pr = zc * tr, pm = zm * tm;
        if (m <= 2) {  // real top (k = m), mirror bottom (k = m)
          Av[m] += Gt * pr - Gb * pm;
          Bw[m] += Gt * tr - Gb * tm;
        }
        if (m >= 4) {  // real bottom (k = m - 4), mirror top (k = m - 4)
          Av[m - 4] += Gb * pr - Gt * pm;
          Bw[m - 4] += Gb * tr - Gt * tm;
        }
        if (m >= 2 && m <= 4) {  // wake rotation, real - mirror (k = m - 2)
          Av[m - 2] += Gw * (pr - pm);
          Bw[m - 2] += Gw * (tr - tm);
        }
        // (the scheduler interleaves all 14 reciprocal chains of a column otherwise: 250 registers; the other waves of
        // the SIMD hide the latency of one chain at a time)
        if (RES_SCHED_LIMIT && (m & 1)) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double w = -yL * Bw[k] * dec[k];
        RES_ST(9 + j * 3 + k, t) = Vj[k] + Av[k] * dec[k];
        RES_ST(18 + j * 3 + k, t) = Wj[k] + ((w < 0.0) ? 0.0 : w);  // quirk (5) [A.6]
      }
    }
  }
}

// ---- 2, 5 and the source-only part of 3 + 6: steering, yaw-added recovery, deflection / deficit constants ----
RES_SRC_FN void res_source_finish(int i) {
  RES_PHASE_FENCE;
  const WfResolveConsts& c = R.c;
  const SrcShared& s0 = R.s;
  const double cg = RES_CG(i), sg = RES_SG(i), ct = s0.ct, ubar = s0.ubar, D = c.D;
  double val = s0.cgv;  // parked by res_source_begin
  val = fmin(fmax(val, -1.0), 1.0);
  const double asv = __any(fabs(val) > 0.3) ? asin_any(val) : asin_small(val);
  const double g_off = c.sw_steer ? 0.5 * asv : 0.0;  // radians added to the commanded yaw
  double dTI = 0.0;
  {  // 5. yaw-added recovery [A.3-5] (the source's own transverse contribution is in V / W now)
    double vsum = 0.0, wsum = 0.0;
#pragma unroll
    for (int q = 0; q < 9; ++q) { vsum += RES_ST(9 + q, i); wsum += RES_ST(18 + q, i); }
    const double I = s0.TIs[0];
    const double k_tke = (ubar * I) * (ubar * I) * 1.5;
    const double vbar = vsum * (1.0 / 9.0), wbar = wsum * (1.0 / 9.0);
    const double I_tot = sqrt_nn((2.0 / 3.0) * 0.5 * (2.0 * k_tke + vbar * vbar + wbar * wbar)) * rcp64(ubar);
    if (c.sw_yar) dTI = c.gch_gain * (I_tot - I);
  }
  // cosd(-g_eff) = cos(g + d), d = asin(val) / 2: half-angle formulas instead of a second cosine
  const double c2d = sqrt_nn(fmax(1.0 - val * val, 0.0)), cd = sqrt_pos(0.5 * (1.0 + c2d)), sd = 0.5 * val * rcp64(cd);
  const double cgd = c.sw_steer ? cg * cd - sg * sd : cg;
  const double gd_rad = -(RES_GR(i) + g_off);  // -(g + d) in radians
  const double s_cc = sqrt_nn(1.0 - ct * cgd), s_c = sqrt_nn(1.0 - ct);
  const double th0 = c.dm * (0.3 * gd_rad * rcp64(cgd)) * (1.0 - s_cc);
  const double tan_th0 = __any(fabs(th0) > 0.5) ? tan_any(th0) : tan_small(th0);
  const double C0 = 1.0 - s_c;
  const double M0 = C0 * (2.0 - C0);
  const double i1sc = rcp64(1.0 + s_c);
  const double sz0d = D * 0.5 * sqrt_pos((ct * cgd * rcp64(2.0 * (1.0 - s_cc))) * i1sc);
  const double sy0d = sz0d * cgd * c.cos_veer;
  const double sM = sqrt_pos(M0);
  const double sz0v = D * 0.5 * sqrt_pos((ct * rcp64(2.0 * (1.0 - s_c))) * i1sc);
  const int lane = res_tid();
  // (stored: max(ambient, TI + dTI) — FLORIS' maximum(sqrt(ti_added^2 + ambient^2), TI) over all turbines at the end of the
  // source step lifts a TI that a NEGATIVE rotor-mean speed drove below ambient; the deficit pass goes on with TI + dTI)
  if (lane < 3) RES_ST(27 + lane, i) = fmax(s0.TIs[lane] + dTI, c.amb);
  if (lane == 0) {
    SrcShared& s = R.s;
    s.dTI = dTI; s.cgd = cgd; s.s_cc = s_cc; s.s_c = s_c; s.th0 = th0; s.tan_th0 = tan_th0; s.M0 = M0;
    s.E0 = C0 * C0 - c.e0c1 * C0 + c.e0c2;
    s.sM = sM; s.sz0d = sz0d; s.sy0d = sy0d; s.is0d = rcp64(sy0d * sz0d); s.lnAB = (1.6 + sM) * rcp64(1.6 - sM);
    s.sz0v = sz0v; s.sy0v = sz0v * cg * c.cos_veer; s.snw = c.near_c * sqrt_pos(ct * 0.5); s.kdef = ct * cg * D * D * 0.125;
    s.ch_pref = c.ch_constant * POW_F64(s0.ai, c.ch_ai) * c.ch_amb_pow;
    s.cgv = cg;  // cosd(-g)
    s.same = (s0.TIs[0] == s0.TIs[1]) && (s0.TIs[1] == s0.TIs[2]);
  }
}

// ---- 3 + 6 + 7 + 8 on the turbines behind the source: deflection (TI before mixing, effective yaw), deficit (TI after
// mixing, commanded yaw), SOSFS, Crespo-Hernandez with the overlap count taken as FLORIS takes it ----
RES_PASS_FN void res_deficit_pass(int i) {
  RES_PHASE_FENCE;
  const WfResolveConsts& c = R.c;
  const SrcShared& s = R.s;
  const int lane = res_tid(), N = R.N;
  const bool veer_on = R.veer_on != 0, same = s.same != 0;
  const double x_i = s.x_i, y_i = s.y_i, D = c.D;
  const double q2 = c.off[2] * c.off[2];
  for (int base = i + 1; base < N; base += 64) {
    const int t = base + lane;
    if (t >= N) continue;
    const double x_t = RES_XS(t), y_t = RES_YS(t);
    const double dx = x_t - x_i;
    const double lin = c.ad + c.bd * dx;
    int cnt = 0;
    double delta = 0.0, amp = 0.0, isy2 = 0.0, isz2 = 0.0;
#pragma unroll 1
    for (int j = 0; j < 3; ++j) {
      if (j == 0 || !same) {  // (one column serves all three when the source's TIs agree)
        // source-side constants of this column [A.3-3, A.3-6]
        const double TIpre = s.TIs[j];
        const double x0d_rel = c.D * s.cgd * (1.0 + s.s_cc) * rcp64(c.sqrt2 * (4.0 * c.defl_alpha * TIpre + 2.0 * c.defl_beta * (1.0 - s.s_c)));
        const double x0d = x0d_rel + x_i;
        const double ix0d_rel = rcp64(x0d_rel);
        const double kyd = c.defl_ka * TIpre + c.defl_kb;
        const double d0 = s.tan_th0 * x0d_rel;
        const double TIq = TIpre + s.dTI;
        const double x0v_rel = c.D * s.cgv * (1.0 + s.s_c) * rcp64(c.sqrt2 * (4.0 * c.alpha * TIq + 2.0 * c.beta * (1.0 - s.s_c)));
        const double x0v = x0v_rel + x_i;
        const double ix0v_rel = rcp64(x0v_rel);
        const double kyv = c.ka * TIq + c.kb;
        // this turbine's column: deflection -> delta; deficit -> amplitude and the Gaussian's 1 / (2 sigma^2)
        double d_near = (dx * ix0d_rel) * d0 + lin;
        if (!(x_t <= x0d)) d_near = 0.0;  // [x >= x_i] holds here
        double d_far = 0.0;
        if (x_t > x0d) {
          const double pfar = s.th0 * s.E0 * (1.0 / 5.2) * sqrt_pos(s.sy0d * s.sz0d * rcp64(kyd * kyd * s.M0));
          const double sy = kyd * (x_t - x0d) + s.sy0d, sz = kyd * (x_t - x0d) + s.sz0d;
          const double sg_ = sqrt_pos(sy * sz * s.is0d);
          const double ln_arg = s.lnAB * (1.6 * sg_ - s.sM) * rcp64(1.6 * sg_ + s.sM);
          d_far = d0 + pfar * LOG_F64(ln_arg) + lin;
        }
        delta = d_near + d_far;
        amp = 0.0; isy2 = 0.0; isz2 = 0.0;
        double sy = 0.0, sz = 0.0;
        bool on = false;
        if (x_t > x_i + 0.1 && x_t < x0v) {  // the masks as FLORIS takes them on the coordinates
          const double up = dx * ix0v_rel, dn = (x0v - x_t) * ix0v_rel;
          sy = dn * s.snw + up * s.sy0v;
          sz = dn * s.snw + up * s.sz0v;
          on = true;
        } else if (x_t >= x0v) {
          sy = kyv * (x_t - x0v) + s.sy0v;
          sz = kyv * (x_t - x0v) + s.sz0v;
          on = true;
        }
        if (on) {
          const double isy = rcp64(sy), isz = rcp64(sz);
          double dd = 1.0 - s.kdef * isy * isz;
          dd = fmin(fmax(dd, 0.0), 1.0);
          amp = 1.0 - sqrt_nn(dd);
          isy2 = 0.5 * isy * isy;
          isz2 = 0.5 * isz * isz;
        }
      }
      const double yy = (y_t + c.off[j]) - y_i - delta;
      double def[3];
      if (!veer_on) {  // r = yy^2 / (2 sy^2) + zz^2 / (2 sz^2), zz = -q, 0, +q
        const double e1 = amp * exp_lean(-(yy * yy) * isy2);
        const double e0 = e1 * exp_lean(-q2 * isz2);
        def[0] = e0; def[1] = e1; def[2] = e0;
      } else {  // FLORIS rCalt [gauss.py]: the Gaussian rotated by the veer angle
        const double ca = c.cos2_veer * isy2 + c.sin2_veer * isz2;
        const double cb = 0.5 * c.sin_2veer * (isz2 - isy2);
        const double cc = c.sin2_veer * isy2 + c.cos2_veer * isz2;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double zz = c.off[k];
          def[k] = amp * exp_lean(-(ca * yy * yy - 2.0 * cb * yy * zz + cc * zz * zz));
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double dU = def[k] * R.Uinit[k];
        if (dU > c.overlap_thr) ++cnt;  // the comparison as FLORIS takes it [A.3-8]
        RES_ST(j * 3 + k, t) = fma(dU, dU, RES_ST(j * 3 + k, t));  // 7. SOSFS [A.3-7]: the sum of squares, root taken where needed
      }
    }
    // 8. Crespo-Hernandez + overlap gating [A.3-8]
    const bool reach = (x_t > x_i) && (x_t <= x_i + 15.0 * D);
    bool gate[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) gate[j] = reach && (fabs(y_i - (y_t + c.off[j])) < 2.0 * D);
    if (gate[0] || gate[1] || gate[2]) {
      const double dxp = (dx <= 0.1) ? dx + 1.0 : dx;  // dx > -0.1 holds for every downstream turbine
      double ti = s.ch_pref * POW_F64(dxp * c.inv_D, c.ch_down);
      if (isnan(ti) || (isinf(ti) && ti > 0)) ti = 0.0;
      const double ti_added = ((double)cnt * (1.0 / 9.0)) * ti;
      const double cand = sqrt_pos(ti_added * ti_added + c.amb * c.amb);
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (gate[j] && cand > RES_ST(27 + j, t)) RES_ST(27 + j, t) = cand;
    }
  }
}

#endif  // RES_PART == 1
// ---- outputs [A.4] of one turbine (sorted index t, caller's index o) from the farm's state in LDS ----
// (tb: the turbine's record in LDS; the nine rotor-grid velocities are kept in registers: one root per grid point)
__device__ __forceinline__ void res_turbine_outputs(const WfResolveConsts& c, const WfResolveArgs& a, const double* tb,
                                                    size_t oo, bool real, const double* Uinit, double wd, double cg_t, int n_tab,
                                                    const double* tws, const double* tpw, const double* tps, double pP3,
                                                    double dens_cbrt, double rho_ref, double& psum, double& lsum) {
  double u[9], m3 = 0.0, mu = 0.0, mv = 0.0, mw = 0.0;
  bool small = true;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    u[q] = Uinit[q % 3] - sqrt_nn(tb[5 + q]);
    const double v = tb[14 + q];
    m3 += u[q] * u[q] * u[q];
    mu += u[q]; mv += v; mw += tb[23 + q];
    small = small && (u[q] > 0.0) && (fabs(v) <= 0.1 * u[q]);
  }
  double dir = 0.0;
  if (__all(small)) {
#pragma unroll
    for (int q = 0; q < 9; ++q) dir += wd - atan_small(tb[14 + q] * rcp64(u[q])) * (1.0 / kDeg);
  } else {
#pragma unroll 1
    for (int q = 0; q < 9; ++q) dir += wd - atan2_any(tb[14 + q], u[q]) * (1.0 / kDeg);
  }
  mu *= (1.0 / 9.0); mv *= (1.0 / 9.0); mw *= (1.0 / 9.0);
  double su = 0.0, sv = 0.0, sw = 0.0;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const double du = u[q] - mu, dv = tb[14 + q] - mv, dw = tb[23 + q] - mw;
    su = fma(du, du, su); sv = fma(dv, dv, sv); sw = fma(dw, dw, sw);
  }
  const double m3m = m3 * (1.0 / 9.0);
  const double wsp = __any(!(m3m > 1.0e-6)) ? cbrt_any(m3m) : cbrt_pos(m3m);
  const double veff = dens_cbrt * (wsp * POW_F64(cg_t, pP3));
  const double pw = rho_ref * interp_fill(veff, n_tab, tws, tpw, tps, 0.0, 0.0);
  const double l0 = (tb[32] + tb[33] + tb[34]) * (1.0 / 3.0);
  const double l1 = sqrt_nn(su * (1.0 / 9.0)), l2 = sqrt_nn(sv * (1.0 / 9.0)), l3 = sqrt_nn(sw * (1.0 / 9.0));
  psum += real ? pw : 0.0;
  lsum += real ? fabs(l0) + fabs(l1) + fabs(l2) + fabs(l3) : 0.0;
  if (a.o_power) a.o_power[oo] = real ? (a.power_mw ? (float)pw * 1.0e-6f : (float)pw) : 0.0f;
  if (a.o_ws) a.o_ws[oo] = real ? (float)wsp : 0.0f;
  if (a.o_wd) a.o_wd[oo] = real ? (float)(dir * (1.0 / 9.0)) : 0.0f;
  if (a.o_load) reinterpret_cast<float4*>(a.o_load)[oo] = real ? make_float4((float)l0, (float)l1, (float)l2, (float)l3) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

#if RES_PART == 1
// ---- outputs [A.4] in the caller's turbine order; the farm's reward ----
RES_PASS_FN void res_outputs(const WfResolveArgs& a, int b, size_t gofs) {
  RES_PHASE_FENCE;
  const WfResolveConsts& c = R.c;
  const int lane = res_tid(), N = R.N;
  double psum = 0.0, lsum = 0.0;
  const int n_real = a.n_real ? a.n_real[b] : N;  // turbines the farm really has (padded layouts)
  for (int t = lane; t < N; t += 64) {
    const int o = a.gidx[gofs + t];
    [[maybe_unused]] const int ty = RES_TY(t);
    res_turbine_outputs(c, a, res_dyn + t * RES_TS, (size_t)b * N + o, o < n_real, R.Uinit, R.wd, RES_CG(t), RES_TN(R, ty),
                        R.tws + RES_TOFS(ty), R.tpw + RES_TOFS(ty), R.tps + RES_TOFS(ty), RES_TC(R, 1, ty, c.pP3),
                        RES_TC(R, 2, ty, c.dens_cbrt), RES_TC(R, 3, ty, c.rho_ref), psum, lsum);  // (a placeholder of a padded layout: zeros out, nothing into the reward)
  }
  if (a.reward) {  // reference simple_env.py:78-84 on the float64 values
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
      psum += __shfl_xor(psum, w);
      lsum += __shfl_xor(lsum, w);
    }
    if (lane == 0) {
      const double wr = a.ws_prev ? a.ws_prev[b] : R.ws;
      a.reward[b] = (float)(psum / n_real / 1.0e6 * 1.0e3 / (wr * wr * wr) - (double)a.load_coef * lsum / (4.0 * n_real));
    }
  }
}

#endif  // RES_PART == 1
// the table columns (and their segment slopes) of every definition -> LDS, by all `nthreads` threads of the block
template <class Shared>
__device__ __forceinline__ void res_stage_tables(Shared& S, const WfResolveConsts& c_arg, const WfResolveArgs& a_arg, int tid, int nthreads) {
#if RES_MT
  const int n_types = a_arg.n_types;
  for (int k = tid; k < n_types; k += nthreads) {
    S.ty_n[k] = (int)a_arg.type_consts[k * (WF_TYPE_CONSTS + 1)];
    for (int j = 0; j < WF_TYPE_CONSTS; ++j) S.ty_c[k][j] = a_arg.type_consts[k * (WF_TYPE_CONSTS + 1) + 1 + j];
  }
#else
  const int n_types = 1;
#endif
  for (int ty = 0; ty < n_types; ++ty) {
    const double* tb = a_arg.tab64 + (size_t)ty * 3 * WF_TABLE_PAD;
#if RES_MT
    const int n = (int)a_arg.type_consts[ty * (WF_TYPE_CONSTS + 1)];
#else
    const int n = c_arg.n_table;
#endif
    for (int k = tid; k < n; k += nthreads) {
      const int d = RES_TOFS(ty) + k;
      S.tws[d] = tb[k];
      S.tct[d] = tb[WF_TABLE_PAD + k];
      S.tpw[d] = tb[2 * WF_TABLE_PAD + k];
      if (k + 1 < n) {  // segment slopes (scipy interp1d: slope * (x - x_lo) + y_lo)
        const double dxk = tb[k + 1] - tb[k];
        S.tcs[d] = (tb[WF_TABLE_PAD + k + 1] - tb[WF_TABLE_PAD + k]) / dxk;
        S.tps[d] = (tb[2 * WF_TABLE_PAD + k + 1] - tb[2 * WF_TABLE_PAD + k]) / dxk;
      }
    }
  }
}

#if defined(WF_RES_STAMP) && !RES_MT  // debug build (tools/res_stamps.py): wave cycles per phase, summed over the launch
__device__ unsigned long long wf_res_stamp[8];
// the four-wave kernel (tools/res4_stamps.py): [wave 0 | wave 3][work phase 1, wait 1, work 2, wait 2, work 3, wait 3], then farms
__device__ unsigned long long wf_res4_stamp[16];
#define RES4_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define RES4_ACC(k, a, b) st4[k] += (b) - (a)
#define RES_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define RES_ACC(k, a, b) st_acc[k] += (b) - (a)
#else
#define RES_T(v)
#define RES_ACC(k, a, b)
#endif
#if !defined(WF_RES_STAMP) || RES_MT
#define RES4_T(v)
#define RES4_ACC(k, a, b)
#define RES4_LACC(k, a, b)
#define RES4_FT(v)
#define RES4_FACC(k, a, b)
#else
// level stages (tools/res4_stamps.py): [wave 0 | wave 3][transverse of the members, wait, chain, transverse of the rest, wait, deficit, wait, turbulence | check + next begin, wait]
__device__ unsigned long long wf_res4_lstamp[20];
__device__ unsigned long long wf_res4_fstamp[16];  // inside the level's transverse pass (thread 0): per part {prologue, terms, hand-over, passes}
#define RES4_FT(v) const unsigned long long v = __builtin_readcyclecounter()
#define RES4_FACC(k, a, b) if (tid == 0) atomicAdd(&wf_res4_fstamp[k], (b) - (a))
#define RES4_LACC(k, a, b) stl[k] += (b) - (a)
#endif
#if RES_PART == 1
__global__ __launch_bounds__(64, WF_RES_OCC) void wf_resolve_kernel(const WfResolveConsts c_arg, const WfResolveArgs a_arg, int n_pad, int min_count) {
  const int lane = threadIdx.x;
  const int N = c_arg.N;
  if (lane == 0) {
    R.c = c_arg;
    R.a = a_arg;
    R.N = N; R.n_pad = n_pad; R.veer_on = c_arg.sin2_veer != 0.0; R.mcore = c_arg.mirror_core;
  }
  res_stage_tables(R, c_arg, a_arg, lane, 64);
  const int n_list = *a_arg.count;
  if (n_list < min_count) return;  // (few enough farms for one residency of the four-wave kernel below: it serves them)
  for (int li = blockIdx.x; li < n_list; li += gridDim.x) {
    const int b = a_arg.list[li];
    RES_T(t_farm);
    __syncthreads();  // (one wave: orders the constants / the previous farm's last reads before the new contents)
    RES_PHASE_FENCE;
    const WfResolveArgs& a = R.a;
    size_t gofs = 0;
    if (a.farm_group) gofs = (size_t)((a.farm_group[b] + a.shift) % a.mod) * N;
    else gofs = (size_t)b * a.geom_stride;
    const float* yaw_b = (a.yaw_state ? a.yaw_state : a.yaw_in) + (size_t)b * N;
    if (lane == 0) {
      const double ws = a.ws[(size_t)b * a.wind_stride];
      double wd = fmod(a.wd[(size_t)b * a.wind_stride], 360.0);  // reference interface.py:664 (Python's %)
      if (wd < 0.0) wd += 360.0;
      R.ws = ws; R.wd = wd; R.Uinf = ws * c_arg.uinf1;  // inflow [A.2]
      for (int k = 0; k < 3; ++k) { R.Uinit[k] = ws * c_arg.shearf[k]; R.dec_a[k] = 4.0 * (c_arg.nu1[k] * ws) / R.Uinf; }
    }
    for (int t = lane; t < N; t += 64) {
      const double g = (double)yaw_b[a.gidx[gofs + t]];
      double sg, cg;
      if (__any(fabs(g) > 45.0)) sincos_any(g * kDeg, sg, cg);  // (never an admissible yaw command)
      else sincos_small(g * kDeg, sg, cg);
      RES_XS(t) = a.gx[gofs + t]; RES_YS(t) = a.gy[gofs + t]; RES_CG(t) = cg; RES_SG(t) = sg; RES_GR(t) = g * kDeg;
#if RES_MT
      reinterpret_cast<int*>(res_dyn + t * RES_TS + 35)[0] = a.type_of[a.gidx[gofs + t]];
#endif
#pragma unroll 1
      for (int q = 0; q < 27; ++q) res_dyn[t * RES_TS + 5 + q] = 0.0;
      for (int j = 0; j < 3; ++j) res_dyn[t * RES_TS + 32 + j] = c_arg.amb;
    }
    __syncthreads();
    for (int t = lane; t < N; t += 64) {  // start of the turbine's x' tie group (sorted order: ties are contiguous)
      int f = t;
      while (f > 0 && RES_XS(f - 1) == RES_XS(t)) --f;
      RES_TIE(t) = f;
    }
    __syncthreads();
#ifdef WF_RES_STAMP
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
#endif
    RES_T(t_a);
    RES_ACC(4, t_farm, t_a);
    for (int i = 0; i < N; ++i) {
      RES_T(t0);
      res_source_begin(i);
      RES_T(t1);
      if (c_arg.sw_tv) res_transverse_pass();
      RES_T(t2);
      res_source_finish(i);
      RES_T(t3);
      if (i + 1 < N) res_deficit_pass(i);
      RES_T(t4);
      RES_ACC(0, t0, t1); RES_ACC(1, t1, t2); RES_ACC(2, t2, t3); RES_ACC(3, t3, t4);
    }
    RES_T(t_b);
    res_outputs(a, b, gofs);
    RES_T(t_c);
    RES_ACC(5, t_b, t_c);
    if (lane == 0) a.flags[b] = 0;
#ifdef WF_RES_STAMP
    if (lane == 0) {
      for (int k = 0; k < 6; ++k) atomicAdd(&wf_res_stamp[k], st_acc[k]);
      atomicAdd(&wf_res_stamp[6], 1ull);
    }
#endif
  }
}

#endif  // RES_PART == 1
#if RES_PART == 2
// ---------------------------------------------------------------------------------------------------------------------
// The same solve with FOUR waves per farm (built in round 3 for flagged counts within one residency of the chip: there the
// re-solve is pure latency — one farm's 80-stage chain, whatever the count — and spreading a source step over four waves
// shortens it; since round 6, with level stages, the kernel for a flagged list of ANY length: its persistent blocks walk the
// list, and it is the only launch behind a step — wfk_launch_resolve).
#ifndef WF_RES4_OCC
#define WF_RES4_OCC 4          // waves per SIMD the register allocator makes room for (four 4-wave blocks, or two blocks with their helper waves, per CU): 128 VGPRs since round 6 (without machine LICM the kernel needs no more — with it, at 256, only two 4-wave blocks fitted a CU and a list beyond 512 farms needed a second round)
#endif
// One farm per block (256 threads: the four waves below; 512 on a short list: four helper waves besides, see the kernel).  The farm's state — per turbine 9 sums of squared deficits, 9 V, 9 W, 3 column TIs,
// float64 — lives in LDS, structure-of-arrays over the sorted turbine index; a lane is not tied to a turbine: for source i
// the lanes take the turbines the source can reach, t = first + lane (first = the start of the source's x' tie group for
// the transverse velocities [A.3-4], i + 1 for deficit and wake-added turbulence), 64 at a time, so the triangle of the
// recurrence costs 96 instead of 160 passes at N = 80.
// The kernel is LATENCY-bound (a few thousand flagged farms are one to two waves per SIMD: every float64 operation of the
// 80-stage dependent chain costs its full latency), so a source step is spread over the block's four waves:
//   waves 0-2  one rotor-grid column each (lateral offset -D/4, 0, +D/4): the transverse pass and the deflection /
//              deficit / SOSFS pass of that column on every reachable turbine;
//   wave 3     the source-only chain of steering, deflection and deficit constants (asin, tan, six roots, the
//              Crespo-Hernandez prefactor) BESIDE the transverse pass, which does not need it;
//   all waves  the source's state and circulations (redundantly: each wave keeps its own copy, no barrier) and the
//              yaw-added recovery once the transverse velocities of all three columns are in.
// Three block barriers per source (a SEQUENTIAL stage; level stages: below): after the transverse pass / the constant chain; after
// the deficit pass (the overlap count of a turbine is the sum of its three columns' counts, exchanged through LDS); after the
// turbulence update.
// The phases are separate NON-INLINED functions that talk through LDS (the per-source constants too): inlined into one
// body the register allocator kept ~370 values live and spilled inside the source loop.
// (History: a thread per turbine, state in registers, two __syncthreads per source, every wave re-deriving the source
// constants: 1.95 ms for 1394 HornsRev1 farms; one wave per farm with the state in LDS: 1.37 ms; profiles/archive/r03_*.)
struct Src4Shared {  // what res4_source_begin leaves (one copy per wave)
  double x_i, y_i, ct, ai, ubar, Vmean, val, TIs[3];
  double Gt, Gb, Gw;  // circulations / (2 pi): top, bottom, wake rotation (commanded yaw)
  int first_tv;
};
struct Fin4Shared {  // the source-only constants of deflection, deficit and wake-added turbulence (written by wave 3)
  double cgd, s_cc, s_c, th0, tan_th0, M0, E0, sM, sz0d, sy0d, is0d, lnAB, sz0v, sy0v, snw, kdef, ch_pref, cgv;
};
// LEVELS (round 6).  Consecutive sources of the sort order that put no deficit and no turbulence on each other — the
// turbines of a column of a grid farm, an x' tie group — need not wait for each other: their rotor speeds, thrusts and
// circulations depend on the sources BEFORE the level only, and the one coupling that does run through the level (the
// transverse velocities of member k at the rotors of the members behind it, which enter their steering and their
// yaw-added recovery) is a sum whose terms are all known once the circulations are.  A level of L = 3 .. 8 members is
// therefore ONE stage of five phases instead of L stages of three:
//   begin       (wave 3, a member per lane)   rotor speed, thrust, circulations of every member — a stage AHEAD, beside the
//                                             previous stage's turbulence passes
//   transverse  (all waves)                   every (member, target, column) pair in its own lane: 64 / L targets per
//                                             wave pass instead of the N - i of a single source (40 of 64 lanes on average
//                                             at N = 80, 8 near the end of the farm); first the chunks that hold the members
//                                             themselves (a fixed share per wave), then the others (drawn from a counter)
//   chain       (wave 3, a member per lane)   steering, recovery, deflection / deficit / turbulence constants, the columns'
//                                             constants: what needs only the members' states while the member chunks are in
//                                             work, the rest behind a COUNT of their finished passes (no barrier)
//   deficit     (all waves)                   pairs as above, behind the chain's FLAG (no barrier: a wave that runs out of
//                                             transverse passes starts here while others finish theirs); a block barrier after
//   turbulence  (all waves but 3)             pairs, the level's check first (wave 0); wave 3 derives the next stage's
//                                             state(s); the stage's second block barrier
// (helper waves 4-7 of a 512-thread launch take part in every "all waves" above.)
// What a target receives from the members of a level is added in MEMBER ORDER (a lane holds one member's contribution and
// leaves it in a wave-private LDS buffer; one lane group per value adds the members' terms of its target in order: see
// RES_HAND_DOUBLES), so every sum is taken in the order of the sequential solve: the results are the SAME BITS as the
// sequential stages' (tests/test_resolve_gpu.py: levels on against levels off).
// Which turbines may share a level is decided from the geometry alone (res4_level_lengths: laterally 8.6 wake widths
// apart, where exp() has taken the deficit below half an ulp of the free stream) — a heuristic, not a proof: at the end of the
// stage the mean cube of every member's rotor speeds is recomputed from the final deficit sums and compared bit for bit
// with the one its state was derived from, and a member's column TIs must not have been raised by a member ahead of it.
// A farm that fails the check is solved again without levels (never seen on the repo's layouts; counted, wfk_res_level_stats).
#define RES_LMAX 8
#define RES4_MAX_WAVES 8  // four waves with roles + four helper waves (level stages of a short list)
#ifndef RES4_SCHED_LIMIT
#define RES4_SCHED_LIMIT 0  // the sequential stage's transverse pass: 0 all 14 reciprocal chains at once, 2 in two batches (8 + 6), 1 two at a time
#endif
#ifndef RES_LV_SCHED_LIMIT
#define RES_LV_SCHED_LIMIT 2
#endif
struct Lvl4Shared {  // what res4_level_begin leaves: written a stage AHEAD, hence two copies (R4.lv, by level parity)
  Src4Shared s[RES_LMAX];
  double m3[RES_LMAX], vtb[RES_LMAX], vcore[RES_LMAX];
  // the lane layout of the level's pair passes (RES4_LV_LANES), divided out once: targets per wave pass (64 / L), the first target
  // of the transverse passes and their chunks — all, the first that holds a member, how many hold members — and the chunks of the
  // deficit / turbulence passes
  int T, tmin, n_ch_tv, c0, n_own, n_ch_df;
};
struct Col4Shared {  // the deflection / deficit constants of ONE rotor-grid column of a member [A.3-3, A.3-6]: they depend on the
  double x0d, ix0d_rel, kyd, d0, pfar, x0v, ix0v_rel, kyv;  // column's TI — derived once by the chain, not by every pair pass
};
struct Lvl4Work {  // what the level stage itself writes and reads (one copy: a stage's last barrier lies between its readers and the next level's writers)
  Fin4Shared f[RES_LMAX];
  double dTI[RES_LMAX];
  double i1sc[RES_LMAX], iubar[RES_LMAX];  // 1 / (1 + sqrt(1 - Ct)), 1 / rotor speed: from the chain's first part to its second
  union {  // (the chain has read the sums by the time it writes the columns' constants: one wave, in program order)
    struct {
      double before[RES_LMAX][9];  // V of member m's rotor as source m finds it (after the members ahead of it)
      double own[RES_LMAX][18];    // V, W of member m's rotor after its own transverse pass
    };
    Col4Shared col[RES_LMAX][3];
  };
};
// The member-to-member hand-over of a pair pass's running sums goes through a wave-private LDS buffer (first version: a chain of
// L - 1 ds_bpermute steps — 35-39 % of a level stage, profiles/r06_handover_ablation.txt): every lane leaves its term there,
// and for each of the pass's three values ONE lane group (group k for value k; a level has at least three members) adds the L
// terms of its target in member order — L reads issued at once and L dependent adds instead of L - 1 round trips through the
// crossbar for all values at once.  The sums are the same sums in the same order: the same bits.  Laid out value-major
// ([value][lane]: a lane's three terms are 512 bytes apart — side by side the compiler would merge two of them into a
// ds_write_b128 that is 16-byte aligned for every other lane only, which this runtime does not serve: see RES_TS).
#define RES_HAND_DOUBLES 3
// (the terms are in LDS before any lane of the wave reads them: the wave's writes have completed, and the compiler keeps the order)
#define RES_HAND_FENCE asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// (measured in round 6: as a compiler barrier alone — the LDS serves a wave's instructions in the order they were issued — the
// results are the same bits and the time is the same; the wait is kept)
struct Res4Shared {
  WfResolveConsts c;
  double tws[RES_NT * WF_TABLE_PAD], tct[RES_NT * WF_TABLE_PAD], tpw[RES_NT * WF_TABLE_PAD];
  double tcs[RES_NT * WF_TABLE_PAD], tps[RES_NT * WF_TABLE_PAD];  // segment slopes of the thrust / power columns
#if RES_MT
  double ty_c[RES_NT][WF_TYPE_CONSTS];
  int ty_n[RES_NT];
#endif
  double ws, wd, Uinf, Uinit[3];
  double dec_a[3];  // 4 nu_k ws / Uinf
  WfResolveArgs a;  // the launch arguments (read from here inside the farm loop)
  double red[RES4_MAX_WAVES][2];
  int N, n_pad, veer_on, mcore;
  int nw;  // waves at work on the block's farm: 4, or 8 with the helper waves of a short list (see wf_resolve4_kernel)
  Src4Shared s[2];  // by stage parity: wave 3 writes the NEXT source's copy while the others still read this one's
  Fin4Shared f;
  double own[18];  // V (0..8) and W (9..17) of the source's own turbine after its transverse pass (see res4_transverse_pass)
  Lvl4Shared lv[2];  // a LEVEL stage's members (round 6, below), by level parity: wave 3 derives the NEXT level's members while the others still read this one's
  Lvl4Work lw;
  double lvl_a, lvl_b;  // two turbines may share a level when |dy'| >= lvl_a + lvl_b dx' (or dx' == 0)
  int levels_on, lv_fail;
  int wp_tv, wp_df, wp_tb;  // next wave pass of a level stage's pair passes (the waves draw them: whoever is free takes the next)
  int mem_done;  // wave passes over the chunks that hold the level's members that are through (the chain waits for all 3 n_own)
  int chain_done;  // the level's chain is through: its constants are in LDS (the deficit passes wait for it)
  double hand[RES4_MAX_WAVES][64 * RES_HAND_DOUBLES + 2];  // per wave: a term per lane and value (see RES_HAND_DOUBLES); [192] holds 0.0: the term of a member the level does not have
};
__shared__ Res4Shared R4;

#define RES4_XS(t) res_dyn[(t) * RES_TS]
#define RES4_YS(t) res_dyn[(t) * RES_TS + 1]
#define RES4_CG(t) res_dyn[(t) * RES_TS + 2]
#define RES4_SG(t) res_dyn[(t) * RES_TS + 3]
#define RES4_GR(t) res_dyn[(t) * RES_TS + 4]
#define RES4_ST(q, t) res_dyn[(t) * RES_TS + 5 + (q)]  // wake2 q = 0..8, V 9..17, W 18..26, TI 27..29
#define RES4_TIE(t) (reinterpret_cast<int*>(res_dyn + RES_TS * R4.n_pad)[(t)])
#define RES4_LVL(t) (reinterpret_cast<int*>(res_dyn + RES_TS * R4.n_pad)[R4.n_pad + (t)])  // members of the level that starts at t (1: a sequential stage)
// (the overlap counts live from a stage's deficit pass to its turbulence pass: a sequential stage's and a level stage's share the space)
#define RES4_CNT(j, t) (reinterpret_cast<int*>(res_dyn + RES_TS * R4.n_pad)[(2 + (j)) * R4.n_pad + (t)])  // overlap count of column j
#define RES4_LCNT(m, j, t) (reinterpret_cast<unsigned char*>(reinterpret_cast<int*>(res_dyn + RES_TS * R4.n_pad) + 2 * R4.n_pad)[((m) * 3 + (j)) * R4.n_pad + (t)])  // overlap count (0 .. 3) of member m, column j: a byte each
#define RES4_DYN_BYTES(n_pad) (sizeof(double) * RES_TS * (size_t)(n_pad) + sizeof(int) * 2 * (size_t)(n_pad) + 3 * RES_LMAX * (size_t)(n_pad))

// ---- the source's state and circulations [A.3-1, A.3-2, A.3-4] ----
// Wave 3 only, one source AHEAD (round 5): source i + 1's rotor speed, thrust and circulations need its turbine's deficits
// (final after the deficit pass of source i) and transverse velocities (final after the transverse pass of source i), so
// wave 3 — idle during the turbulence pass of stage i — derives them there, into the other parity's copy; the block barrier
// that ends the stage publishes them.  Before, every wave derived them at the START of stage i + 1: a cube root, a table
// probe, a root and two reciprocals of pure latency on each of the farm's N stages.  The three column TIs of the source are
// NOT final at that point (the turbulence pass is writing them): wave 3 snapshots them at the start of the stage proper
// (res4_source_chain), ahead of the barrier behind which the recovery and the deficit pass read them.
// ... and SPECULATIVELY one phase earlier still: wave 3 is idle during the deficit pass of stage i as well, and source i
// usually does not reach turbine i + 1 at all (neighbours in the sort order stand side by side: a Gaussian 10 sigma off
// adds less than an ulp to the deficit sums).  So it derives source i + 1's state THERE, from the sums as they stand
// before — or torn by — that pass, and beside the turbulence pass only recomputes the mean cube of the rotor speeds from
// the final sums: everything derived is a function of that one number, so if it is bit for bit the speculated one the
// published state stands, otherwise it is derived again as before.  The stage's last phase shrinks from 3 700 cycles to the
// turbulence pass's own 1 500 whenever the speculation holds; the results are the same bits either way.
// the mean cube of the rotor-grid speeds of turbine i as its deficit sums stand now
RES_SRC_FN double res4_rotor_m3(int i) {
  RES_PHASE_FENCE;
  double m3 = 0.0;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const double u = R4.Uinit[q % 3] - sqrt_nn(RES4_ST(q, i));
    m3 += u * u * u;
  }
  return m3 * (1.0 / 9.0);
}
RES_SRC_FN void res4_source_begin(int tid, int i, double m3m) {
  RES_PHASE_FENCE;
  const WfResolveConsts& c = R4.c;
  const double cg = RES4_CG(i), sg = RES4_SG(i);
  double vs = 0.0;
#pragma unroll
  for (int q = 0; q < 9; ++q) vs += RES4_ST(9 + q, i);
  const double ubar = __any(!(m3m > 1.0e-6)) ? cbrt_any(m3m) : cbrt_pos(m3m);
  [[maybe_unused]] const int ty = RES_TY(i);
  const double lo_ct = res_lit(0.0001), hi_ct = res_lit(0.9999);
  double ct_tab = interp_fill_uniform(ubar, RES_TN(R4, ty), R4.tws + RES_TOFS(ty), R4.tct + RES_TOFS(ty), R4.tcs + RES_TOFS(ty), lo_ct, hi_ct);
  ct_tab = fmin(fmax(ct_tab, lo_ct), hi_ct);
  const double ct = ct_tab * cg;
  const double ai = 0.5 * rcp64(cg) * (1.0 - sqrt_nn(1.0 - ct * cg));
  const double G_wr = (0.25 * kTwoPi) * c.D * (ai - ai * ai) * ubar * RES_TC(R4, 0, ty, c.inv_TSR);
  const double gam_top = (kTwoPi / 16.0) * c.D * c.vel_top * R4.Uinf * ct;
  const double gam_bot = (kTwoPi / 16.0) * c.D * c.vel_bot * R4.Uinf * ct;
  const double sc = sg * cg;
  if ((tid & 63) == 0) {
    Src4Shared& s = R4.s[i & 1];
    s.x_i = RES4_XS(i); s.y_i = RES4_YS(i); s.ct = ct; s.ai = ai; s.ubar = ubar; s.Vmean = vs * (1.0 / 9.0);
    s.Gt = sc * gam_top * (1.0 / kTwoPi); s.Gb = -sc * gam_bot * (1.0 / kTwoPi); s.Gw = G_wr * (1.0 / kTwoPi);
    s.first_tv = RES4_TIE(i);
    // secondary steering [A.3-2]: the three means on the source's own grid are geometry constants
    const double v_top = gam_top * c.k_top, v_bot = -gam_bot * c.k_bot, v_core = G_wr * c.k_core;
    s.val = 2.0 * (s.Vmean - v_core) * rcp64(v_top + v_bot);
  }
}

// ---- 4. transverse velocities (commanded yaw) of grid column j on every turbine at or downstream of the source, ties
// included; the 7 + 7 distinct vertical offsets of the three vortices and their ground mirrors ----
// The new V / W of the SOURCE's own turbine go to a side buffer, not into the state: the other waves may still be reading
// that turbine's V in res4_source_begin (its rotor mean feeds the steering) — res4_recovery commits them after the barrier.
RES_PASS_FN void res4_transverse_pass(int tid, int i, int j) {
  RES_PHASE_FENCE;
  const WfResolveConsts& c = R4.c;
  const Src4Shared& s = R4.s[i & 1];
  const int lane = tid & 63, N = R4.N;
  const double x_i = s.x_i, y_i = s.y_i, Gt = s.Gt, Gb = s.Gb, Gw = s.Gw;
  const double qd = c.off[2], neps = c.num_eps, twoHH = 2.0 * c.HH, eps2 = c.eps2, ieps2 = c.inv_eps2;
  const bool mcore = R4.mcore != 0;
  for (int base = s.first_tv; base < N; base += 64) {
    const int t = base + lane;
    if (t >= N) continue;
    const double dx = RES4_XS(t) - x_i, y_t = RES4_YS(t);
    double dec[3], Vj[3], Wj[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      dec[k] = eps2 * rcp64(fma(R4.dec_a[k], dx, eps2));
      Vj[k] = RES4_ST(9 + j * 3 + k, t);
      Wj[k] = RES4_ST(18 + j * 3 + k, t);
    }
    const double yL = (y_t + c.off[j] - y_i) + neps;
    const double yL2 = yL * yL;
    const double Ey = exp_lean(-yL2 * ieps2);
    double Av[3] = {0.0, 0.0, 0.0}, Bw[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int m = 0; m < 7; ++m) {
      const double zc = (double)(m - 3) * qd + neps, zm = zc + twoHH;
      const double tr = (1.0 - Ey * c.ezc[m]) * rcp64(yL2 + zc * zc);   // core / r of a real vortex at offset zc
      double tm = rcp64(yL2 + zm * zm);                                   // ... of a mirror vortex at zm
      if (mcore) tm *= 1.0 - Ey * c.ezm7[m];  // (1 - Ey ezm == 1.0 exactly unless the hub is very low)
      const double pr = zc * tr, pm = zm * tm;
      if (m <= 2) {  // real top (k = m), mirror bottom (k = m)
        Av[m] += Gt * pr - Gb * pm;
        Bw[m] += Gt * tr - Gb * tm;
      }
      if (m >= 4) {  // real bottom (k = m - 4), mirror top (k = m - 4)
        Av[m - 4] += Gb * pr - Gt * pm;
        Bw[m - 4] += Gb * tr - Gt * tm;
      }
      if (m >= 2 && m <= 4) {  // wake rotation, real - mirror (k = m - 2)
        Av[m - 2] += Gw * (pr - pm);
        Bw[m - 2] += Gw * (tr - tm);
      }
      // (above two waves per SIMD the 14 interleaved reciprocal chains of a column would not fit the registers)
      if (RES4_SCHED_LIMIT == 2 ? (m == 3) : (RES4_SCHED_LIMIT && (m & 1))) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double w = -yL * Bw[k] * dec[k];
      const double vn = Vj[k] + Av[k] * dec[k], wn = Wj[k] + ((w < 0.0) ? 0.0 : w);  // quirk (5) [A.6]
      if (t == i) {
        R4.own[j * 3 + k] = vn;
        R4.own[9 + j * 3 + k] = wn;
      } else {
        RES4_ST(9 + j * 3 + k, t) = vn;
        RES4_ST(18 + j * 3 + k, t) = wn;
      }
    }
  }
}

// ---- 2 and the source-only part of 3 + 6 + 8: steering, deflection / deficit / turbulence constants (wave 3, beside the
// transverse pass) ----
RES_SRC_FN void res4_source_chain(int tid, int i) {
  RES_PHASE_FENCE;
  const WfResolveConsts& c = R4.c;
  Src4Shared& s0 = R4.s[i & 1];
  if ((tid & 63) < 3) s0.TIs[tid & 63] = RES4_ST(27 + (tid & 63), i);  // (final since the barrier that ended the previous stage)
  const double cg = RES4_CG(i), sg = RES4_SG(i), ct = s0.ct, D = c.D;
  double val = s0.val;
  val = fmin(fmax(val, -1.0), 1.0);
  const double asv = __any(fabs(val) > 0.3) ? asin_any(val) : asin_small(val);
  const double g_off = c.sw_steer ? 0.5 * asv : 0.0;  // radians added to the commanded yaw
  // cosd(-g_eff) = cos(g + d), d = asin(val) / 2: half-angle formulas instead of a second cosine
  const double c2d = sqrt_nn(fmax(1.0 - val * val, 0.0)), cd = sqrt_pos(0.5 * (1.0 + c2d)), sd = 0.5 * val * rcp64(cd);
  const double cgd = c.sw_steer ? cg * cd - sg * sd : cg;
  const double gd_rad = -(RES4_GR(i) + g_off);  // -(g + d) in radians
  const double s_cc = sqrt_nn(1.0 - ct * cgd), s_c = sqrt_nn(1.0 - ct);
  const double th0 = c.dm * (0.3 * gd_rad * rcp64(cgd)) * (1.0 - s_cc);
  const double tan_th0 = __any(fabs(th0) > 0.5) ? tan_any(th0) : tan_small(th0);
  const double C0 = 1.0 - s_c;
  const double M0 = C0 * (2.0 - C0);
  const double i1sc = rcp64(1.0 + s_c);
  const double sz0d = D * 0.5 * sqrt_pos((ct * cgd * rcp64(2.0 * (1.0 - s_cc))) * i1sc);
  const double sy0d = sz0d * cgd * c.cos_veer;
  const double sM = sqrt_pos(M0);
  const double sz0v = D * 0.5 * sqrt_pos((ct * rcp64(2.0 * (1.0 - s_c))) * i1sc);
  if ((tid & 63) == 0) {
    Fin4Shared& f = R4.f;
    f.cgd = cgd; f.s_cc = s_cc; f.s_c = s_c; f.th0 = th0; f.tan_th0 = tan_th0; f.M0 = M0;
    f.E0 = C0 * C0 - c.e0c1 * C0 + c.e0c2;
    f.sM = sM; f.sz0d = sz0d; f.sy0d = sy0d; f.is0d = rcp64(sy0d * sz0d); f.lnAB = (1.6 + sM) * rcp64(1.6 - sM);
    f.sz0v = sz0v; f.sy0v = sz0v * cg * c.cos_veer; f.snw = c.near_c * sqrt_pos(ct * 0.5); f.kdef = ct * cg * D * D * 0.125;
    f.ch_pref = c.ch_constant * POW_F64(s0.ai, c.ch_ai) * c.ch_amb_pow;
    f.cgv = cg;  // cosd(-g)
  }
}

// ---- 5. yaw-added recovery [A.3-5]: the source's own transverse contribution is in V / W now (every wave; returns the
// increment of the source's TI, which waves 0-2 apply to their column) ----
RES_SRC_FN double res4_recovery(int tid, int i) {
  RES_PHASE_FENCE;
  const WfResolveConsts& c = R4.c;
  const int wave = tid >> 6;
  const Src4Shared& s0 = R4.s[i & 1];
  double vsum = 0.0, wsum = 0.0;
#pragma unroll
  for (int q = 0; q < 9; ++q) { vsum += R4.own[q]; wsum += R4.own[9 + q]; }
  const double I = s0.TIs[0], ubar = s0.ubar;
  const double k_tke = (ubar * I) * (ubar * I) * 1.5;
  const double vbar = vsum * (1.0 / 9.0), wbar = wsum * (1.0 / 9.0);
  const double I_tot = sqrt_nn((2.0 / 3.0) * 0.5 * (2.0 * k_tke + vbar * vbar + wbar * wbar)) * rcp64(ubar);
  const double dTI = c.sw_yar ? c.gch_gain * (I_tot - I) : 0.0;
  if (wave < 3 && (tid & 63) == 0) RES4_ST(27 + wave, i) = fmax(s0.TIs[wave] + dTI, c.amb);  // (stored: see res_source_finish)
  if (wave < 3 && (tid & 63) < 3) {  // commit the source's own column (nothing reads it before the next barrier)
    const int q = wave * 3 + (tid & 63);
    RES4_ST(9 + q, i) = R4.own[q];
    RES4_ST(18 + q, i) = R4.own[9 + q];
  }
  return dTI;
}

// ---- 3 + 6 + 7 of grid column j on the turbines behind the source: deflection (TI before mixing, effective yaw), deficit
// (TI after mixing, commanded yaw), SOSFS; the column's part of the overlap count, taken as FLORIS takes it ----
RES_PASS_FN void res4_deficit_pass(int tid, int i, int j, double dTI) {
  RES_PHASE_FENCE;
  const WfResolveConsts& c = R4.c;
  const Src4Shared& s = R4.s[i & 1];
  const Fin4Shared& f = R4.f;
  const int lane = tid & 63, N = R4.N;
  const bool veer_on = R4.veer_on != 0;
  const double x_i = s.x_i, y_i = s.y_i;
  const double q2 = c.off[2] * c.off[2];
  // source-side constants of this column [A.3-3, A.3-6]
  const double TIpre = s.TIs[j];
  const double x0d_rel = c.D * f.cgd * (1.0 + f.s_cc) * rcp64(c.sqrt2 * (4.0 * c.defl_alpha * TIpre + 2.0 * c.defl_beta * (1.0 - f.s_c)));
  const double x0d = x0d_rel + x_i;
  const double ix0d_rel = rcp64(x0d_rel);
  const double kyd = c.defl_ka * TIpre + c.defl_kb;
  const double d0 = f.tan_th0 * x0d_rel;
  const double pfar = f.th0 * f.E0 * (1.0 / 5.2) * sqrt_pos(f.sy0d * f.sz0d * rcp64(kyd * kyd * f.M0));
  const double TIq = TIpre + dTI;
  const double x0v_rel = c.D * f.cgv * (1.0 + f.s_c) * rcp64(c.sqrt2 * (4.0 * c.alpha * TIq + 2.0 * c.beta * (1.0 - f.s_c)));
  const double x0v = x0v_rel + x_i;
  const double ix0v_rel = rcp64(x0v_rel);
  const double kyv = c.ka * TIq + c.kb;
  for (int base = i + 1; base < N; base += 64) {
    const int t = base + lane;
    if (t >= N) continue;
    const double x_t = RES4_XS(t), y_t = RES4_YS(t);
    const double dx = x_t - x_i;
    const double lin = c.ad + c.bd * dx;
    // this turbine's column: deflection -> delta; deficit -> amplitude and the Gaussian's 1 / (2 sigma^2)
    double d_near = (dx * ix0d_rel) * d0 + lin;
    if (!(x_t <= x0d)) d_near = 0.0;  // [x >= x_i] holds here
    double d_far = 0.0;
    if (x_t > x0d) {
      const double sy = kyd * (x_t - x0d) + f.sy0d, sz = kyd * (x_t - x0d) + f.sz0d;
      const double sg_ = sqrt_pos(sy * sz * f.is0d);
      const double ln_arg = f.lnAB * (1.6 * sg_ - f.sM) * rcp64(1.6 * sg_ + f.sM);
      d_far = d0 + pfar * LOG_F64(ln_arg) + lin;
    }
    const double delta = d_near + d_far;
    double amp = 0.0, isy2 = 0.0, isz2 = 0.0, sy = 0.0, sz = 0.0;
    bool on = false;
    if (x_t > x_i + 0.1 && x_t < x0v) {  // the masks as FLORIS takes them on the coordinates
      const double up = dx * ix0v_rel, dn = (x0v - x_t) * ix0v_rel;
      sy = dn * f.snw + up * f.sy0v;
      sz = dn * f.snw + up * f.sz0v;
      on = true;
    } else if (x_t >= x0v) {
      sy = kyv * (x_t - x0v) + f.sy0v;
      sz = kyv * (x_t - x0v) + f.sz0v;
      on = true;
    }
    if (on) {
      const double isy = rcp64(sy), isz = rcp64(sz);
      double dd = 1.0 - f.kdef * isy * isz;
      dd = fmin(fmax(dd, 0.0), 1.0);
      amp = 1.0 - sqrt_nn(dd);
      isy2 = 0.5 * isy * isy;
      isz2 = 0.5 * isz * isz;
    }
    const double yy = (y_t + c.off[j]) - y_i - delta;
    double def[3];
    if (!veer_on) {  // r = yy^2 / (2 sy^2) + zz^2 / (2 sz^2), zz = -q, 0, +q
      const double e1 = amp * exp_lean(-(yy * yy) * isy2);
      const double e0 = e1 * exp_lean(-q2 * isz2);
      def[0] = e0; def[1] = e1; def[2] = e0;
    } else {  // FLORIS rCalt [gauss.py]: the Gaussian rotated by the veer angle
      const double ca = c.cos2_veer * isy2 + c.sin2_veer * isz2;
      const double cb = 0.5 * c.sin_2veer * (isz2 - isy2);
      const double cc = c.sin2_veer * isy2 + c.cos2_veer * isz2;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double zz = c.off[k];
        def[k] = amp * exp_lean(-(ca * yy * yy - 2.0 * cb * yy * zz + cc * zz * zz));
      }
    }
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double dU = def[k] * R4.Uinit[k];
      if (dU > c.overlap_thr) ++cnt;  // the comparison as FLORIS takes it [A.3-8]
      RES4_ST(j * 3 + k, t) = fma(dU, dU, RES4_ST(j * 3 + k, t));  // 7. SOSFS [A.3-7]: the sum of squares, root taken where needed
    }
    RES4_CNT(j, t) = cnt;
  }
}

// ---- 8. Crespo-Hernandez + overlap gating [A.3-8] of grid column j (the overlap count is the sum over the three columns) ----
RES_PASS_FN void res4_turbulence_pass(int tid, int i, int j) {
  RES_PHASE_FENCE;
  const WfResolveConsts& c = R4.c;
  const Src4Shared& s = R4.s[i & 1];
  const int lane = tid & 63, N = R4.N;
  const double x_i = s.x_i, y_i = s.y_i, D = c.D, ch_pref = R4.f.ch_pref;
  for (int base = i + 1; base < N; base += 64) {
    const int t = base + lane;
    if (t >= N) continue;
    const double x_t = RES4_XS(t), y_t = RES4_YS(t);
    const bool gate = (x_t > x_i) && (x_t <= x_i + 15.0 * D) && (fabs(y_i - (y_t + c.off[j])) < 2.0 * D);
    if (!gate) continue;
    const double dx = x_t - x_i;
    const int cnt = RES4_CNT(0, t) + RES4_CNT(1, t) + RES4_CNT(2, t);
    const double dxp = (dx <= 0.1) ? dx + 1.0 : dx;  // dx > -0.1 holds for every downstream turbine
    double ti = ch_pref * POW_F64(dxp * c.inv_D, c.ch_down);
    if (isnan(ti) || (isinf(ti) && ti > 0)) ti = 0.0;
    const double ti_added = ((double)cnt * (1.0 / 9.0)) * ti;
    const double cand = sqrt_pos(ti_added * ti_added + c.amb * c.amb);
    if (cand > RES4_ST(27 + j, t)) RES4_ST(27 + j, t) = cand;
  }
}

// ======================================================================================================================
// LEVEL stages (see Lvl4Shared).  Lane layout of the pair passes: T = 64 / L targets per wave pass, lane = ks T + tl —
// member ks of the level on target (chunk base + tl); lanes beyond L T idle.
#define RES4_LV_LANES(L)                                        \
  const int lane = tid & 63, T = lv.T;                          \
  int ks = 0;                                                   \
  _Pragma("unroll") for (int q_ = 1; q_ < RES_LMAX; ++q_) ks += (lane >= q_ * T) ? 1 : 0; \
  const int tl = lane - ks * T;                                 \
  const bool member = ks < (L);                                 \
  const int km = member ? ks : 0

// ---- begin: rotor speed, thrust, induction, circulations of every member (wave 3, member m in lane m) ----
RES_SRC_FN void res4_level_begin(int tid, Lvl4Shared& lv, int i0, int L) {
  RES_PHASE_FENCE;
  const WfResolveConsts& c = R4.c;
  const int m = tid & 63;
  const bool act = m < L;
  const int i = i0 + (act ? m : 0);
  const double m3m = res4_rotor_m3(i);
  const double cg = RES4_CG(i), sg = RES4_SG(i);
  double ubar = cbrt_pos(act && m3m > 1.0e-6 ? m3m : 1.0);
  if (__any(act && !(m3m > 1.0e-6))) {  // (the routine the sequential stage would have taken for THIS member: same bits)
    const double ua = cbrt_any(m3m);
    ubar = (m3m > 1.0e-6) ? ubar : ua;
  }
  [[maybe_unused]] const int ty = RES_TY(i);
  const double lo_ct = res_lit(0.0001), hi_ct = res_lit(0.9999);
  double ct_tab = interp_fill(ubar, RES_TN(R4, ty), R4.tws + RES_TOFS(ty), R4.tct + RES_TOFS(ty), R4.tcs + RES_TOFS(ty), lo_ct, hi_ct);
  ct_tab = fmin(fmax(ct_tab, lo_ct), hi_ct);
  const double ct = ct_tab * cg;
  const double ai = 0.5 * rcp64(cg) * (1.0 - sqrt_nn(1.0 - ct * cg));
  const double G_wr = (0.25 * kTwoPi) * c.D * (ai - ai * ai) * ubar * RES_TC(R4, 0, ty, c.inv_TSR);
  const double gam_top = (kTwoPi / 16.0) * c.D * c.vel_top * R4.Uinf * ct;
  const double gam_bot = (kTwoPi / 16.0) * c.D * c.vel_bot * R4.Uinf * ct;
  const double sc = sg * cg;
  if (act) {
    Src4Shared& s = lv.s[m];
    s.x_i = RES4_XS(i); s.y_i = RES4_YS(i); s.ct = ct; s.ai = ai; s.ubar = ubar;
    s.Gt = sc * gam_top * (1.0 / kTwoPi); s.Gb = -sc * gam_bot * (1.0 / kTwoPi); s.Gw = G_wr * (1.0 / kTwoPi);
    s.first_tv = RES4_TIE(i);
    const double v_top = gam_top * c.k_top, v_bot = -gam_bot * c.k_bot, v_core = G_wr * c.k_core;
    lv.vtb[m] = v_top + v_bot; lv.vcore[m] = v_core;
    lv.m3[m] = m3m;
    if (m == 0) {
      // (64 / L from a table, the quotients by T through a float reciprocal: turbine indices are far below 2^20 and T <= 21 — the
      // half added before the multiplication is worth more than any rounding; four integer divisions would be ~ 600 cycles of this
      // wave's chain)
      const int T = (int)((0x08090A0C10152040ull >> (8 * (L - 1))) & 0xFF), tmin = s.first_tv, N = R4.N;
      const float iT = __frcp_rn((float)T);
      const int c0 = (int)(((float)(i0 - tmin) + 0.5f) * iT);
      lv.T = T; lv.tmin = tmin; lv.n_ch_tv = (int)(((float)(N - tmin + T - 1) + 0.5f) * iT); lv.c0 = c0;
      lv.n_own = (int)(((float)(i0 + L - 1 - tmin) + 0.5f) * iT) - c0 + 1;
      lv.n_ch_df = (int)(((float)(N - i0 - 1 + T - 1) + 0.5f) * iT);
    }
  }
}

// ---- 4. transverse velocities: every (member, target, column) pair; what a target receives is added in member order ----
RES_PASS_FN void res4_level_transverse(int tid, Lvl4Shared& lv, int i0, int L, int part) {
  RES_PHASE_FENCE;
  RES4_FT(f0);
  const WfResolveConsts& c = R4.c;
  RES4_LV_LANES(L);
  const int wave = tid >> 6, N = R4.N;
  const Src4Shared& s = lv.s[km];
  const double x_i = s.x_i, y_i = s.y_i, Gt = s.Gt, Gb = s.Gb, Gw = s.Gw;
  const int first = s.first_tv, tmin = lv.tmin;
  const double qd = c.off[2], neps = c.num_eps, twoHH = 2.0 * c.HH, eps2 = c.eps2, ieps2 = c.inv_eps2;
  const bool mcore = R4.mcore != 0;
  // part 1: the chunks that hold the level's own members (the chain waits for them), a fixed share per wave; part 2: all
  // other chunks, drawn from a counter — wave 3 joins when its chain is done
  const int n_ch = lv.n_ch_tv, c0 = lv.c0, n_own = lv.n_own;
  const int n_wp = part == 1 ? 3 * n_own : 3 * (n_ch - n_own);
  const int nw = R4.nw;
  int n_mine = 0;
  for (int wp_s = wave;; wp_s += nw) {
    int wp = wp_s;
    if (part != 1) {
      if (lane == 0) wp = atomicAdd(&R4.wp_tv, 1);
      wp = __builtin_amdgcn_readfirstlane(wp);
    }
    if (wp >= n_wp) break;
    ++n_mine;
    RES4_FT(f1);
    int ch = wp / 3;
    if (part == 1) ch += c0;
    else if (ch >= c0) ch += n_own;
    const int j = wp % 3, t = tmin + ch * T + tl;
    const bool valid_t = member && t < N;
    const bool active = valid_t && t >= first;
    const int tt = t < N ? t : N - 1;
    double cv[3] = {0.0, 0.0, 0.0}, cw[3] = {0.0, 0.0, 0.0};
    if (active) {
      const double dx = RES4_XS(tt) - x_i, y_t = RES4_YS(tt);
      double dec[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) dec[k] = eps2 * rcp64(fma(R4.dec_a[k], dx, eps2));
      const double yL = (y_t + c.off[j] - y_i) + neps;
      const double yL2 = yL * yL;
      const double Ey = exp_lean(-yL2 * ieps2);
      double Av[3] = {0.0, 0.0, 0.0}, Bw[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int m = 0; m < 7; ++m) {
        const double zc = (double)(m - 3) * qd + neps, zm = zc + twoHH;
        const double tr = (1.0 - Ey * c.ezc[m]) * rcp64(yL2 + zc * zc);
        double tm = rcp64(yL2 + zm * zm);
        if (mcore) tm *= 1.0 - Ey * c.ezm7[m];
        const double pr = zc * tr, pm = zm * tm;
        if (m <= 2) {
          Av[m] += Gt * pr - Gb * pm;
          Bw[m] += Gt * tr - Gb * tm;
        }
        if (m >= 4) {
          Av[m - 4] += Gb * pr - Gt * pm;
          Bw[m - 4] += Gb * tr - Gt * tm;
        }
        if (m >= 2 && m <= 4) {
          Av[m - 2] += Gw * (pr - pm);
          Bw[m - 2] += Gw * (tr - tm);
        }
        if (RES_LV_SCHED_LIMIT == 2 ? (m == 3) : (RES_LV_SCHED_LIMIT && (m & 1))) __builtin_amdgcn_sched_barrier(0);  // (the reciprocal chains in two batches — 8, then 6 — instead of all 14 at once: the lane bookkeeping of a level needs a few registers)
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double w = -yL * Bw[k] * dec[k];
        cv[k] = Av[k] * dec[k];
        cw[k] = (w < 0.0) ? 0.0 : w;  // quirk (5) [A.6]
      }
    }
    // the running sums: every lane leaves its term in the wave's buffer; lane group k (k = 0, 1, 2) adds, for its target, the members'
    // terms of value k in member order (see RES_HAND_DOUBLES), takes the snapshots a member's chain needs on the way, and stores
    // (a pair the sequential solve does not visit — a target ahead of the member's tie group, a lane without a member — leaves
    // +0.0: adding it changes no bit of a sum that started at +0.0)
    RES4_FT(f2);
    const int mt = t - i0;  // the target as a member of this level (0 .. L - 1), if it is one
    double* hb = R4.hand[wave];
    const bool coll = ks < 3 && t < N;
    int slot[RES_LMAX];  // where member sm's term for this lane's target lies (the zero slot for members the level does not have)
#pragma unroll
    for (int sm = 0; sm < RES_LMAX; ++sm) slot[sm] = sm < L ? ks * 64 + sm * T + tl : 64 * RES_HAND_DOUBLES;
#pragma unroll
    for (int half = 0; half < 2; ++half) {  // V, then W through the same buffer
#pragma unroll
      for (int k = 0; k < 3; ++k) hb[k * 64 + lane] = half ? cw[k] : cv[k];
      RES_HAND_FENCE;
      if (coll) {
        const int q = (half ? 18 : 9) + j * 3 + ks;
        double p = RES4_ST(q, tt);
        double term[RES_LMAX];
#pragma unroll
        for (int sm = 0; sm < RES_LMAX; ++sm) term[sm] = hb[slot[sm]];
        if (part == 1) {  // (the chunks that hold the members: the snapshots their chains need — picked from the running sums
                          // by selects, not by a branch per member)
          double bef = p, own = p;
#pragma unroll
          for (int sm = 0; sm < RES_LMAX; ++sm) {
            bef = sm == mt ? p : bef;
            p = p + term[sm];
            own = sm == mt ? p : own;
          }
          if (mt >= 0 && mt < L) {
            if (!half) R4.lw.before[mt][j * 3 + ks] = bef;
            R4.lw.own[mt][(half ? 9 : 0) + j * 3 + ks] = own;
          }
        } else {
#pragma unroll
          for (int sm = 0; sm < RES_LMAX; ++sm) p = p + term[sm];
        }
        RES4_ST(q, tt) = p;
      }
      RES_HAND_FENCE;  // (the reads are through before the next terms overwrite the buffer)
    }
    RES4_FT(f3);
    RES4_FACC((part - 1) * 4 + 1, f1, f2); RES4_FACC((part - 1) * 4 + 2, f2, f3); RES4_FACC((part - 1) * 4 + 3, f0, f0 + 1);
    if (wp_s == wave) RES4_FACC((part - 1) * 4, f0, f1);
  }
  // part 1: this wave's passes over the members' chunks are in LDS (the fence above): counted for wave 3, whose chain waits for
  // all of them — no barrier: the other waves go straight on to the other chunks (part 2)
  if (part == 1 && n_mine > 0 && lane == 0) atomicAdd(&R4.mem_done, n_mine);
}

// wave 3, ahead of its chain: until every pass over the members' chunks is through.  (Bounded: a count that never arrives —
// it always does: the passes are a fixed share of waves that are running — sends the farm to the sequential solve, like a
// failed check, instead of hanging the block.)
RES_SRC_FN void res4_level_wait_members(Lvl4Shared& lv) {
  const int n1 = 3 * lv.n_own;
  int it = 0;
  while (__hip_atomic_load(&R4.mem_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < n1 && it < (1 << 20)) {
    __builtin_amdgcn_s_sleep(1);
    ++it;
  }
  if (it >= (1 << 20)) R4.lv_fail = 1;
  RES_HAND_FENCE;
}
// every wave, between its last transverse pass and its first deficit pass: until wave 3's chain has left the members' constants
// (as a rule it has, long ago — the transverse passes take longer than the chain; no barrier: whoever runs out of transverse
// passes starts on the deficit passes while the others finish theirs — the two kinds of pass touch different sums)
RES_SRC_FN void res4_level_wait_chain() {
  int it = 0;
  while (__hip_atomic_load(&R4.chain_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0 && it < (1 << 20)) {
    __builtin_amdgcn_s_sleep(1);
    ++it;
  }
  if (it >= (1 << 20)) R4.lv_fail = 1;
  RES_HAND_FENCE;
}

// ---- the part of a member's chain that needs nothing but its state (res4_level_begin): wave 3 derives it WHILE the other waves
// are on the transverse passes over the members' chunks — the roots and reciprocals that do not depend on the steering, and the
// Crespo-Hernandez prefactor (a pow): a quarter of the chain's instructions off the stage's critical path ----
RES_SRC_FN void res4_level_chain_pre(int tid, Lvl4Shared& lv, int i0, int L) {
  RES_PHASE_FENCE;
  const WfResolveConsts& c = R4.c;
  const int m = tid & 63;
  const bool act = m < L;
  const int mm = act ? m : 0, i = i0 + mm;
  const Src4Shared& s0 = lv.s[mm];
  const double cg = RES4_CG(i), ct = s0.ct, D = c.D;
  const double s_c = sqrt_nn(1.0 - ct);
  const double C0 = 1.0 - s_c;
  const double M0 = C0 * (2.0 - C0);
  const double i1sc = rcp64(1.0 + s_c);
  const double sM = sqrt_pos(M0);
  const double sz0v = D * 0.5 * sqrt_pos((ct * rcp64(2.0 * (1.0 - s_c))) * i1sc);
  const double ch_pref = c.ch_constant * POW_F64(s0.ai, c.ch_ai) * c.ch_amb_pow;
  if (act) {
    Fin4Shared& f = R4.lw.f[m];
    f.s_c = s_c; f.M0 = M0;
    f.E0 = C0 * C0 - c.e0c1 * C0 + c.e0c2;
    f.sM = sM; f.lnAB = (1.6 + sM) * rcp64(1.6 - sM);
    f.sz0v = sz0v; f.sy0v = sz0v * cg * c.cos_veer; f.snw = c.near_c * sqrt_pos(ct * 0.5); f.kdef = ct * cg * D * D * 0.125;
    f.ch_pref = ch_pref;
    f.cgv = cg;
    R4.lw.i1sc[m] = i1sc;
    R4.lw.iubar[m] = rcp64(s0.ubar);
  }
}

// ---- 2, 5 and the source-only part of 3 + 6 + 8 of every member (wave 3, member m in lane m): steering from the
// transverse velocities the member finds at its rotor, yaw-added recovery from the ones it leaves there ----
RES_SRC_FN void res4_level_chain(int tid, Lvl4Shared& lv, int i0, int L) {
  RES_PHASE_FENCE;
  const WfResolveConsts& c = R4.c;
  const int m = tid & 63;
  const bool act = m < L;
  const int mm = act ? m : 0, i = i0 + mm;
  Src4Shared& s0 = lv.s[mm];
  double TIs[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) TIs[j] = RES4_ST(27 + j, i);  // (final: no member ahead may raise them — checked in the turbulence pass)
  double vs = 0.0;
#pragma unroll
  for (int q = 0; q < 9; ++q) vs += R4.lw.before[mm][q];
  const double Vmean = vs * (1.0 / 9.0);
  const double cg = RES4_CG(i), sg = RES4_SG(i), ct = s0.ct, D = c.D, ubar = s0.ubar;
  double val = 2.0 * (Vmean - lv.vcore[mm]) * rcp64(lv.vtb[mm]);
  val = fmin(fmax(val, -1.0), 1.0);
  double asv = asin_small(val);
  if (__any(act && fabs(val) > 0.3)) {  // (per member the routine its sequential stage would take)
    const double aa = asin_any(val);
    asv = fabs(val) > 0.3 ? aa : asv;
  }
  const double g_off = c.sw_steer ? 0.5 * asv : 0.0;
  const double c2d = sqrt_nn(fmax(1.0 - val * val, 0.0)), cd = sqrt_pos(0.5 * (1.0 + c2d)), sd = 0.5 * val * rcp64(cd);
  const double cgd = c.sw_steer ? cg * cd - sg * sd : cg;
  const double gd_rad = -(RES4_GR(i) + g_off);
  const double s_cc = sqrt_nn(1.0 - ct * cgd);
  const double th0 = c.dm * (0.3 * gd_rad * rcp64(cgd)) * (1.0 - s_cc);
  double tan_th0 = tan_small(th0);
  if (__any(act && fabs(th0) > 0.5)) {
    const double ta = tan_any(th0);
    tan_th0 = fabs(th0) > 0.5 ? ta : tan_th0;
  }
  const double i1sc = R4.lw.i1sc[mm];  // (res4_level_chain_pre)
  const double sz0d = D * 0.5 * sqrt_pos((ct * cgd * rcp64(2.0 * (1.0 - s_cc))) * i1sc);
  const double sy0d = sz0d * cgd * c.cos_veer;
  // 5. yaw-added recovery [A.3-5]
  double vsum = 0.0, wsum = 0.0;
#pragma unroll
  for (int q = 0; q < 9; ++q) { vsum += R4.lw.own[mm][q]; wsum += R4.lw.own[mm][9 + q]; }
  const double I = TIs[0];
  const double k_tke = (ubar * I) * (ubar * I) * 1.5;
  const double vbar = vsum * (1.0 / 9.0), wbar = wsum * (1.0 / 9.0);
  const double I_tot = sqrt_nn((2.0 / 3.0) * 0.5 * (2.0 * k_tke + vbar * vbar + wbar * wbar)) * R4.lw.iubar[mm];
  const double dTI = c.sw_yar ? c.gch_gain * (I_tot - I) : 0.0;
  if (act) {
    Fin4Shared& f = R4.lw.f[m];
    f.cgd = cgd; f.s_cc = s_cc; f.th0 = th0; f.tan_th0 = tan_th0;
    f.sz0d = sz0d; f.sy0d = sy0d; f.is0d = rcp64(sy0d * sz0d);
    s0.Vmean = Vmean; s0.val = val;
    R4.lw.dTI[m] = dTI;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      s0.TIs[j] = TIs[j];
      RES4_ST(27 + j, i) = fmax(TIs[j] + dTI, c.amb);  // (stored: see res_source_finish)
    }
  }
  // ... and the constants of each member's three columns, a (member, column) pair per lane: the expressions of the sequential
  // stage's deficit pass (res4_deficit_pass) on the same values — every pair pass of the level would otherwise derive them again
  // (five reciprocals and a root of its ~ 30)
  RES_HAND_FENCE;
  {
    const int l = tid & 63, m2 = l / 3, j2 = l - 3 * m2;
    const bool act2 = l < 3 * L;
    const int mq = act2 ? m2 : 0;
    const Fin4Shared& f = R4.lw.f[mq];
    const double TIpre = lv.s[mq].TIs[j2], x_i = lv.s[mq].x_i;
    const double x0d_rel = c.D * f.cgd * (1.0 + f.s_cc) * rcp64(c.sqrt2 * (4.0 * c.defl_alpha * TIpre + 2.0 * c.defl_beta * (1.0 - f.s_c)));
    const double kyd = c.defl_ka * TIpre + c.defl_kb;
    const double TIq = TIpre + R4.lw.dTI[mq];
    const double x0v_rel = c.D * f.cgv * (1.0 + f.s_c) * rcp64(c.sqrt2 * (4.0 * c.alpha * TIq + 2.0 * c.beta * (1.0 - f.s_c)));
    Col4Shared cs;
    cs.x0d = x0d_rel + x_i;
    cs.ix0d_rel = rcp64(x0d_rel);
    cs.kyd = kyd;
    cs.d0 = f.tan_th0 * x0d_rel;
    cs.pfar = f.th0 * f.E0 * (1.0 / 5.2) * sqrt_pos(f.sy0d * f.sz0d * rcp64(kyd * kyd * f.M0));
    cs.x0v = x0v_rel + x_i;
    cs.ix0v_rel = rcp64(x0v_rel);
    cs.kyv = c.ka * TIq + c.kb;
    if (act2) R4.lw.col[mq][j2] = cs;
  }
}

// ---- 3 + 6 + 7: every (member, target behind it, column) pair; the squared deficits added in member order ----
RES_PASS_FN void res4_level_deficit(int tid, Lvl4Shared& lv, int i0, int L) {
  RES_PHASE_FENCE;
  const WfResolveConsts& c = R4.c;
  RES4_LV_LANES(L);
  const int N = R4.N;
  const Src4Shared& s = lv.s[km];
  const Fin4Shared& f = R4.lw.f[km];
  const bool veer_on = R4.veer_on != 0;
  const double x_i = s.x_i, y_i = s.y_i;
  const double q2 = c.off[2] * c.off[2];
  const int k_src = i0 + km;
  const int n_wp = 3 * lv.n_ch_df;
  for (;;) {
    int wp = 0;
    if (lane == 0) wp = atomicAdd(&R4.wp_df, 1);
    wp = __builtin_amdgcn_readfirstlane(wp);
    if (wp >= n_wp) break;
    const int j = wp % 3, t = i0 + 1 + (wp / 3) * T + tl;
    const bool valid_t = member && t < N;
    const bool active = valid_t && t > k_src;
    const int tt = t < N ? t : N - 1;
    double dU[3] = {0.0, 0.0, 0.0};
    int cnt = 0;
    if (active) {
      // source-side constants of this column [A.3-3, A.3-6] (res4_level_chain)
      const Col4Shared& cs = R4.lw.col[km][j];
      const double x0d = cs.x0d, ix0d_rel = cs.ix0d_rel, kyd = cs.kyd, d0 = cs.d0, pfar = cs.pfar;
      const double x0v = cs.x0v, ix0v_rel = cs.ix0v_rel, kyv = cs.kyv;
      const double x_t = RES4_XS(tt), y_t = RES4_YS(tt);
      const double dx = x_t - x_i;
      const double lin = c.ad + c.bd * dx;
      double d_near = (dx * ix0d_rel) * d0 + lin;
      if (!(x_t <= x0d)) d_near = 0.0;
      double d_far = 0.0;
      if (x_t > x0d) {
        const double sy = kyd * (x_t - x0d) + f.sy0d, sz = kyd * (x_t - x0d) + f.sz0d;
        const double sg_ = sqrt_pos(sy * sz * f.is0d);
        const double ln_arg = f.lnAB * (1.6 * sg_ - f.sM) * rcp64(1.6 * sg_ + f.sM);
        d_far = d0 + pfar * LOG_F64(ln_arg) + lin;
      }
      const double delta = d_near + d_far;
      double amp = 0.0, isy2 = 0.0, isz2 = 0.0, sy = 0.0, sz = 0.0;
      bool on = false;
      if (x_t > x_i + 0.1 && x_t < x0v) {
        const double up = dx * ix0v_rel, dn = (x0v - x_t) * ix0v_rel;
        sy = dn * f.snw + up * f.sy0v;
        sz = dn * f.snw + up * f.sz0v;
        on = true;
      } else if (x_t >= x0v) {
        sy = kyv * (x_t - x0v) + f.sy0v;
        sz = kyv * (x_t - x0v) + f.sz0v;
        on = true;
      }
      if (on) {
        const double isy = rcp64(sy), isz = rcp64(sz);
        double dd = 1.0 - f.kdef * isy * isz;
        dd = fmin(fmax(dd, 0.0), 1.0);
        amp = 1.0 - sqrt_nn(dd);
        isy2 = 0.5 * isy * isy;
        isz2 = 0.5 * isz * isz;
      }
      const double yy = (y_t + c.off[j]) - y_i - delta;
      double def[3];
      if (!veer_on) {
        const double e1 = amp * exp_lean(-(yy * yy) * isy2);
        const double e0 = e1 * exp_lean(-q2 * isz2);
        def[0] = e0; def[1] = e1; def[2] = e0;
      } else {
        const double ca = c.cos2_veer * isy2 + c.sin2_veer * isz2;
        const double cb = 0.5 * c.sin_2veer * (isz2 - isy2);
        const double cc = c.sin2_veer * isy2 + c.cos2_veer * isz2;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double zz = c.off[k];
          def[k] = amp * exp_lean(-(ca * yy * yy - 2.0 * cb * yy * zz + cc * zz * zz));
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        dU[k] = def[k] * R4.Uinit[k];
        if (dU[k] > c.overlap_thr) ++cnt;  // the comparison as FLORIS takes it [A.3-8]
      }
    }
    double* hb = R4.hand[tid >> 6];
#pragma unroll
    for (int k = 0; k < 3; ++k) hb[k * 64 + lane] = dU[k];  // (+0.0 for a pair the sequential solve does not visit: fma(0, 0, p) is p)
    RES_HAND_FENCE;
    if (valid_t) RES4_LCNT(ks, j, tt) = (unsigned char)cnt;
    if (ks < 3 && t < N) {  // lane group k: the squared deficits of grid row k, added in member order — 7. SOSFS [A.3-7]
      double p = RES4_ST(j * 3 + ks, tt);
      double d[RES_LMAX];
#pragma unroll
      for (int sm = 0; sm < RES_LMAX; ++sm) d[sm] = hb[sm < L ? ks * 64 + sm * T + tl : 64 * RES_HAND_DOUBLES];
#pragma unroll
      for (int sm = 0; sm < RES_LMAX; ++sm) p = fma(d[sm], d[sm], p);
      RES4_ST(j * 3 + ks, tt) = p;
    }
    RES_HAND_FENCE;
  }
}

// ---- 8. Crespo-Hernandez + overlap gating of every (member, target behind it) pair, the three columns in the lane
// (waves 0-2).  A target that is itself a member must not be touched by a member ahead of it: the check of the level ----
RES_PASS_FN void res4_level_turbulence(int tid, Lvl4Shared& lv, int i0, int L) {
  RES_PHASE_FENCE;
  const WfResolveConsts& c = R4.c;
  RES4_LV_LANES(L);
  const int N = R4.N;
  const Src4Shared& s = lv.s[km];
  const double x_i = s.x_i, y_i = s.y_i, D = c.D, ch_pref = R4.lw.f[km].ch_pref;
  const int k_src = i0 + km;
  const int n_ch = lv.n_ch_df;
  for (;;) {
    int ch = 0;
    if (lane == 0) ch = atomicAdd(&R4.wp_tb, 1);
    ch = __builtin_amdgcn_readfirstlane(ch);
    if (ch >= n_ch) break;
    const int t = i0 + 1 + ch * T + tl;
    const bool valid_t = member && t < N;
    const bool active = valid_t && t > k_src;
    const int tt = t < N ? t : N - 1;
    const double x_t = RES4_XS(tt), y_t = RES4_YS(tt);
    const bool reach = active && (x_t > x_i) && (x_t <= x_i + 15.0 * D);
    bool gate[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) gate[j] = reach && (fabs(y_i - (y_t + c.off[j])) < 2.0 * D);
    double cand = 0.0;
    if (gate[0] || gate[1] || gate[2]) {
      const double dx = x_t - x_i;
      const int cnt = (int)RES4_LCNT(ks, 0, tt) + (int)RES4_LCNT(ks, 1, tt) + (int)RES4_LCNT(ks, 2, tt);
      const double dxp = (dx <= 0.1) ? dx + 1.0 : dx;
      double ti = ch_pref * POW_F64(dxp * c.inv_D, c.ch_down);
      if (isnan(ti) || (isinf(ti) && ti > 0)) ti = 0.0;
      const double ti_added = ((double)cnt * (1.0 / 9.0)) * ti;
      cand = sqrt_pos(ti_added * ti_added + c.amb * c.amb);
    }
    const int mt = t - i0;
    if (active && mt < L) {  // a member behind this one: its column TIs were read by its own chain already
      const Src4Shared& st = lv.s[mt];
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (gate[j] && cand > st.TIs[j]) R4.lv_fail = 1;
    }
    double* hb = R4.hand[tid >> 6];
#pragma unroll
    for (int j = 0; j < 3; ++j) hb[j * 64 + lane] = gate[j] ? cand : 0.0;  // (0.0 never raises a TI: they start at the ambient one)
    RES_HAND_FENCE;
    if (ks < 3 && t < N && mt >= L) {  // lane group j: column j's TI of a target behind the level, raised member by member
      double p = RES4_ST(27 + ks, tt);
      double cd[RES_LMAX];
#pragma unroll
      for (int sm = 0; sm < RES_LMAX; ++sm) cd[sm] = hb[sm < L ? ks * 64 + sm * T + tl : 64 * RES_HAND_DOUBLES];
#pragma unroll
      for (int sm = 0; sm < RES_LMAX; ++sm) p = cd[sm] > p ? cd[sm] : p;
      RES4_ST(27 + ks, tt) = p;
    }
    RES_HAND_FENCE;
  }
}

// ---- the check of a level (wave 3, member m in lane m): the deficit sums of every member as they stand now hold what
// the members ahead of it added — the mean cube of its rotor speeds must be the one its state was derived from ----
RES_SRC_FN void res4_level_check(int tid, Lvl4Shared& lv, int i0, int L) {
  RES_PHASE_FENCE;
  const int m = tid & 63;
  const bool act = m >= 1 && m < L;
  const double m3_now = res4_rotor_m3(i0 + (act ? m : 0));
  if (act && __double_as_longlong(m3_now) != __double_as_longlong(lv.m3[m])) R4.lv_fail = 1;
}

// ---- which turbines may share a level (once per farm): from sorted turbine t on, as many as are pairwise either tied
// in x' or laterally far enough apart; fewer than three are not worth a level stage (two sequential stages cost the same) ----
RES_PASS_FN void res4_level_lengths(int tid, bool on) {
  RES_PHASE_FENCE;
  const int N = R4.N;
  const double la = R4.lvl_a, lb = R4.lvl_b;
  for (int t = tid < 256 ? tid : N; t < N; t += 256) {
    int L = 1;
    if (on) {
      while (L < RES_LMAX && t + L < N) {
        const int cnd = t + L;
        const double xc = RES4_XS(cnd), yc = RES4_YS(cnd);
        bool ok = true;
        for (int k = t; k < cnd; ++k) {
          const double dx = xc - RES4_XS(k), dy = fabs(yc - RES4_YS(k));
          ok = ok && (dx == 0.0 || dy >= la + lb * dx);
        }
        if (!ok) break;
        ++L;
      }
    }
    RES4_LVL(t) = L >= 3 ? L : 1;
  }
}

// ---- outputs [A.4] in the caller's turbine order; the farm's reward ----
RES_PASS_FN void res4_outputs(int tid, const WfResolveArgs& a, int b, size_t gofs) {
  RES_PHASE_FENCE;
  asm volatile("" : "+v"(tid));
  const WfResolveConsts& c = R4.c;
  const int N = R4.N;
  double psum = 0.0, lsum = 0.0;
  const int n_real = a.n_real ? a.n_real[b] : N;  // turbines the farm really has (padded layouts)
  for (int t = tid < 256 ? tid : N; t < N; t += 256) {  // (the four waves with roles: a helper wave has none here)
    const int o = a.gidx[gofs + t];
    [[maybe_unused]] const int ty = RES_TY(t);
    res_turbine_outputs(c, a, res_dyn + t * RES_TS, (size_t)b * N + o, o < n_real, R4.Uinit, R4.wd, RES4_CG(t), RES_TN(R4, ty),
                        R4.tws + RES_TOFS(ty), R4.tpw + RES_TOFS(ty), R4.tps + RES_TOFS(ty), RES_TC(R4, 1, ty, c.pP3),
                        RES_TC(R4, 2, ty, c.dens_cbrt), RES_TC(R4, 3, ty, c.rho_ref), psum, lsum);
  }
  if (a.reward) {  // reference simple_env.py:78-84 on the float64 values
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
      psum += __shfl_xor(psum, w);
      lsum += __shfl_xor(lsum, w);
    }
    if ((tid & 63) == 0 && tid < 256) { R4.red[tid >> 6][0] = psum; R4.red[tid >> 6][1] = lsum; }
    __syncthreads();
    if (tid == 0) {
      double ps = 0.0, ls = 0.0;
      for (int w = 0; w < 4; ++w) { ps += R4.red[w][0]; ls += R4.red[w][1]; }
      const double wr = a.ws_prev ? a.ws_prev[b] : R4.ws;
      a.reward[b] = (float)(ps / n_real / 1.0e6 * 1.0e3 / (wr * wr * wr) - (double)a.load_coef * ls / (4.0 * n_real));
    }
  }
}

#if !RES_MT
// [0] farms solved by the four-wave kernel, [1] of them solved a second time without levels (a level failed its check),
// [2] level stages, [3] sources inside them, [4] sequential stages, [5] farms solved with the helper waves at work — since the
// library was loaded (wfk_res_level_stats)
__device__ unsigned long long wf_res_lvl_stat[8];
#endif
#ifndef RES4_WAVES_ATTR
#define RES4_WAVES_ATTR
#endif
// HELPER WAVES (round 6).  A short list — a farm per CU or so: the 8-GPU share of a sweep, 39 farms — leaves every SIMD to ONE wave,
// and a float64 chain at one wave per SIMD leaves every other issue slot empty.  A launch of 512 threads gives the block four more
// waves (4-7) that take their share of a level stage's pair passes: the passes are drawn from counters, so more waves simply draw
// faster.  They have no part in a sequential stage (its three phases are one wave pass per column) beyond its barriers.  The sums
// are taken in member order whoever computes the terms: the same bits either way (tests/test_resolve_gpu.py).
// The list's length is known on the device only, and a long list is better served by 256-thread blocks, three or four to a CU, than
// by two wide ones (2 011 HornsRev2 farms: + 1.06 ms against + 1.25): the host picks the width from the length the PREVIOUS launch
// found (seen_host; the first launch is a narrow one).  Should a wide launch meet a list beyond two blocks per CU after all, waves
// 4-7 return at once (the registers of a wave that has ended are NOT handed to a new block while its block lives — measured: + 1.64
// — so this is damage control for one step, not a way to launch).
__global__ __launch_bounds__(64 * RES4_MAX_WAVES, WF_RES4_OCC) RES4_WAVES_ATTR void wf_resolve4_kernel(const WfResolveConsts c_arg, const WfResolveArgs a_arg, int n_pad, int max_count, int levels, int helpers_max) {
  const int tid = threadIdx.x;
  const int N = c_arg.N;
  const int n_list = *a_arg.count;
  // the list's length for the host, which picks the NEXT launch's width by it (wfk_launch_resolve4) — written when it changes only:
  // the store goes to pinned host memory
  // (on the path that returns and, below, behind the copies of the arguments: a store ahead of them makes the compiler copy the
  // whole argument block to scratch first)
  const bool tell = blockIdx.x == 0 && tid == 0 && a_arg.seen_host != nullptr;
  if (n_list == 0 || n_list > max_count) {  // nothing flagged (the common case: the launch costs its dispatch only) / the one-wave-per-farm kernel serves this count
    if (tell && *a_arg.seen_dev != n_list) {
      *a_arg.seen_dev = n_list;
      __hip_atomic_store(a_arg.seen_host, n_list, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  const bool helpers = blockDim.x > 256 && n_list <= helpers_max;
  if (tid >= 256 && !helpers) return;  // (ahead of the first barrier: a wave that has ended is not waited for)
  if (tid == 0) {
    R4.nw = helpers ? RES4_MAX_WAVES : 4;
    R4.c = c_arg;
    R4.a = a_arg;
    R4.N = N; R4.n_pad = n_pad; R4.veer_on = c_arg.sin2_veer != 0.0; R4.mcore = c_arg.mirror_core;
    // level rule (see Lvl4Shared): 8.6 standard deviations of a member's wake (exp(-37) of an amplitude <= 0.6 is below half an
    // ulp of the free stream) — sigma <= sigma_0 + k_y dx, sigma_0 = D / sqrt(8) at the rotor whatever the thrust (near wake:
    // near_wake_c D sqrt(Ct / 2)), the growth rate at a turbulence intensity of 0.3 — plus a quarter rotor of grid offsets and
    // the wake's deflection (<= 0.1 D + 0.15 dx at 40 deg of yaw)
    const double ky = fmax(c_arg.ka, c_arg.defl_ka) * 0.3 + fmax(c_arg.kb, c_arg.defl_kb);
    const double s0 = fmax(0.35355339059327379 * c_arg.D, 0.70710678118654752 * c_arg.near_c);
    R4.lvl_a = 0.35 * c_arg.D + 8.6 * s0;
    R4.lvl_b = 8.6 * ky + 0.15;
    R4.levels_on = levels && c_arg.sw_tv;
    if (tell && *R4.a.seen_dev != n_list) {
      *R4.a.seen_dev = n_list;
      __hip_atomic_store(R4.a.seen_host, n_list, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  res_stage_tables(R4, c_arg, a_arg, tid < 256 ? tid : 0x40000000, 256);
  for (int li = blockIdx.x; li < n_list; li += gridDim.x) {
    __syncthreads();  // the constants are in place / the previous farm's last readers are done
    RES_PHASE_FENCE;
    const WfResolveArgs& a = R4.a;
    if (tid == 0) {
      const int b = a_arg.list[li];
      const double ws = a.ws[(size_t)b * a.wind_stride];
      double wd = fmod(a.wd[(size_t)b * a.wind_stride], 360.0);  // reference interface.py:664 (Python's %)
      if (wd < 0.0) wd += 360.0;
      R4.ws = ws; R4.wd = wd; R4.Uinf = ws * R4.c.uinf1;  // inflow [A.2]
      for (int k = 0; k < 3; ++k) { R4.Uinit[k] = ws * R4.c.shearf[k]; R4.dec_a[k] = 4.0 * (R4.c.nu1[k] * ws) / R4.Uinf; }
    }
#if defined(WF_RES_STAMP) && !RES_MT
    unsigned long long st4[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long stl[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    [[maybe_unused]] unsigned n_lv = 0, n_lv_src = 0, n_seq = 0;
    bool use_levels = true;
    for (int attempt = 0; attempt < 2; ++attempt) {
    {  // (the farm's index and geometry offset are derived again where they are needed — here and at the outputs — instead of
       // being held across the solve: the stage loop has no register to spare)
    const int b = a_arg.list[li];
    const size_t gofs = a.farm_group ? (size_t)((a.farm_group[b] + a.shift) % a.mod) * N : (size_t)b * a.geom_stride;
    const float* yaw_b = (a.yaw_state ? a.yaw_state : a.yaw_in) + (size_t)b * N;
    int ti = tid;
    asm volatile("" : "+v"(ti));  // (what is derived from it is derived per attempt, not held across the solve)
    for (int t = ti < 256 ? ti : N; t < N; t += 256) {
      const double g = (double)yaw_b[a.gidx[gofs + t]];
      double sg, cg;
      if (__any(fabs(g) > 45.0)) sincos_any(g * kDeg, sg, cg);  // (never an admissible yaw command)
      else sincos_small(g * kDeg, sg, cg);
      RES4_XS(t) = a.gx[gofs + t]; RES4_YS(t) = a.gy[gofs + t]; RES4_CG(t) = cg; RES4_SG(t) = sg; RES4_GR(t) = g * kDeg;
#if RES_MT
      reinterpret_cast<int*>(res_dyn + t * RES_TS + 35)[0] = a.type_of[a.gidx[gofs + t]];
#endif
#pragma unroll 1
      for (int q = 0; q < 27; ++q) res_dyn[t * RES_TS + 5 + q] = 0.0;
      for (int j = 0; j < 3; ++j) res_dyn[t * RES_TS + 32 + j] = R4.c.amb;
    }
    }
    int ti = tid;
    asm volatile("" : "+v"(ti));
    if (ti == 0) { R4.lv_fail = 0; R4.wp_tv = 0; R4.mem_done = 0; R4.chain_done = 0; }
    if ((ti & 63) == 0) R4.hand[ti >> 6][64 * RES_HAND_DOUBLES] = 0.0;
    __syncthreads();
    for (int t = ti < 256 ? ti : N; t < N; t += 256) {  // start of the turbine's x' tie group (sorted order: ties are contiguous)
      int f = t;
      while (f > 0 && RES4_XS(f - 1) == RES4_XS(t)) --f;
      RES4_TIE(t) = f;
    }
    res4_level_lengths(ti, use_levels && R4.levels_on);
    __syncthreads();
    int lp = 0;  // parity of the next level stage's record (R4.lv[lp]): whoever derives a level's members ahead writes there
    if ((ti >> 6) == 3) {  // the first stage's source state(s)
      if (RES4_LVL(0) > 1) res4_level_begin(ti, R4.lv[0], 0, RES4_LVL(0));
      else res4_source_begin(ti, 0, res4_rotor_m3(0));
    }
    __syncthreads();
    for (int i = 0; i < N;) {
      // (the thread index is made opaque once per source: everything derived from it — wave, lane, a dozen LDS addresses per
      // phase — is recomputed where it is used instead of being hoisted out of this loop and held, or spilled, across it)
      int tq = tid;
      asm volatile("" : "+v"(tq));
      const int wq = __builtin_amdgcn_readfirstlane(tq >> 6);  // (the wave: uniform, and known to be — the branches on it are scalar ones)
      const int L = RES4_LVL(i);
      if (L > 1) {  // ---- a level stage: sources i .. i + L - 1 at once (their states were derived a stage ahead) ----
        Lvl4Shared& lv = R4.lv[lp];
        RES4_T(l0);
        // (the pass counters and flags are zeroed a barrier before they are used: wp_tv, mem_done and chain_done behind the
        // previous level stage's deficit passes or at the farm's start, wp_df and wp_tb here)
        if (tq == 0) { R4.wp_df = 0; R4.wp_tb = 0; }
        res4_level_transverse(tq, lv, i, L, 1);  // the chunks that hold the members themselves
        RES4_T(l1);
        if (wq == 3) {
          res4_level_chain_pre(tq, lv, i, L);
          res4_level_wait_members(lv);  // (no barrier: the waves without a member chunk are on the other chunks already)
        }
        RES4_T(l2);
        if (wq == 3) {
          res4_level_chain(tq, lv, i, L);
          RES_HAND_FENCE;  // (the chain's constants are in LDS ...)
          if ((tq & 63) == 0) __hip_atomic_store(&R4.chain_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (... before the flag is)
        }
        RES4_T(l3);
        res4_level_transverse(tq, lv, i, L, 2);  // every other target: whoever is free, wave 3 behind its chain
        RES4_T(l4);
        res4_level_wait_chain();
        RES4_T(l5);
        res4_level_deficit(tq, lv, i, L);
        RES4_T(l6);
        __syncthreads();
        RES4_T(l7);
        if (tq == 0) { R4.wp_tv = 0; R4.mem_done = 0; R4.chain_done = 0; }
        if (wq != 3) {
          if (wq == 0) res4_level_check(tq, lv, i, L);  // (ahead of its share of the drawn turbulence passes: wave 3 is busy with the next states)
          res4_level_turbulence(tq, lv, i, L);
        } else {  // the state(s) of the next stage's source(s)
          if (i + L < N) {
            const int Ln = RES4_LVL(i + L);
            if (Ln > 1) res4_level_begin(tq, R4.lv[lp ^ 1], i + L, Ln);
            else res4_source_begin(tq, i + L, res4_rotor_m3(i + L));
          }
        }
        RES4_T(l8);
        __syncthreads();
        RES4_T(l9);
        RES4_LACC(0, l0, l1); RES4_LACC(1, l1, l2); RES4_LACC(2, l2, l3); RES4_LACC(3, l3, l4); RES4_LACC(4, l4, l5);
        RES4_LACC(5, l5, l6); RES4_LACC(6, l6, l7); RES4_LACC(7, l7, l8); RES4_LACC(8, l8, l9);
        ++n_lv; n_lv_src += L;
        lp ^= 1;
        i += L;
        continue;
      }
      ++n_seq;
      if (wq >= 4) {  // a helper wave: the stage's barriers only
        __syncthreads();
        if (i + 1 < N) __syncthreads();
        __syncthreads();
        ++i;
        continue;
      }
      RES4_T(p0);
      if (wq < 3) {
        if (R4.c.sw_tv) res4_transverse_pass(tq, i, wq);
        else if ((tq & 63) < 3) {  // (no transverse velocities: the side buffer holds the unchanged — zero — state)
          const int q = wq * 3 + (tq & 63);
          R4.own[q] = RES4_ST(9 + q, i);
          R4.own[9 + q] = RES4_ST(18 + q, i);
        }
      } else {
        res4_source_chain(tq, i);
      }
      RES4_T(p1);
      __syncthreads();
      RES4_T(p2);
      double dTI = 0.0, m3_spec = 0.0;
      if (wq < 3) dTI = res4_recovery(tq, i);
      if (i + 1 < N) {
        const int Ln = RES4_LVL(i + 1);  // (> 1: a level follows — its members' states are derived beside the turbulence pass, from the final sums)
        if (wq < 3) {
          res4_deficit_pass(tq, i, wq, dTI);
        } else if (Ln == 1) {  // (see res4_source_begin: the next source's state, speculated beside the deficit pass ...)
          m3_spec = res4_rotor_m3(i + 1);
          res4_source_begin(tq, i + 1, m3_spec);
        }
        RES4_T(p3);
        __syncthreads();
        RES4_T(p4);
        RES4_ACC(2, p2, p3); RES4_ACC(3, p3, p4);
        if (wq < 3) {
          res4_turbulence_pass(tq, i, wq);
        } else if (Ln == 1) {  // (... and confirmed, or derived again, beside the turbulence pass)
          const double m3_now = res4_rotor_m3(i + 1);
          if (__double_as_longlong(m3_now) != __double_as_longlong(m3_spec)) res4_source_begin(tq, i + 1, m3_now);
        } else {
          res4_level_begin(tq, R4.lv[lp], i + 1, Ln);
        }
        RES4_T(p5);
        RES4_ACC(4, p4, p5);
      }
      RES4_T(p6);
      __syncthreads();
      RES4_T(p7);
      RES4_ACC(0, p0, p1); RES4_ACC(1, p1, p2); RES4_ACC(5, p6, p7);
      ++i;
    }
    if (!(use_levels && R4.lv_fail)) break;  // (block-uniform: every thread reads the flag behind the stage's last barrier)
    use_levels = false;  // a level failed its check: the farm once more, every source a stage of its own
    __syncthreads();
    }
#if !RES_MT
    if (tid == 0) {
      atomicAdd(&wf_res_lvl_stat[0], 1ull);
      if (!use_levels) atomicAdd(&wf_res_lvl_stat[1], 1ull);
      atomicAdd(&wf_res_lvl_stat[2], (unsigned long long)n_lv);
      atomicAdd(&wf_res_lvl_stat[3], (unsigned long long)n_lv_src);
      atomicAdd(&wf_res_lvl_stat[4], (unsigned long long)n_seq);
      if (R4.nw > 4) atomicAdd(&wf_res_lvl_stat[5], 1ull);
    }
#endif
#if defined(WF_RES_STAMP) && !RES_MT
    if ((tid & 63) == 0 && ((tid >> 6) == 0 || (tid >> 6) == 3)) {
      for (int k = 0; k < 6; ++k) atomicAdd(&wf_res4_stamp[((tid >> 6) ? 6 : 0) + k], st4[k]);
      for (int k = 0; k < 10; ++k) atomicAdd(&wf_res4_lstamp[((tid >> 6) ? 10 : 0) + k], stl[k]);
      if (tid == 0) atomicAdd(&wf_res4_stamp[12], 1ull);
    }
#endif
    RES_PHASE_FENCE;
    const int b = a_arg.list[li];
    const size_t gofs = a.farm_group ? (size_t)((a.farm_group[b] + a.shift) % a.mod) * N : (size_t)b * a.geom_stride;
    res4_outputs(tid, a, b, gofs);
    if (tid == 0) a.flags[b] = 0;
  }
}

#if !RES_MT
// level stages on (1, default; WF_RES_LEVELS seeds it) or off (0: every source a stage of its own — the A/B switch and the
// other side of the bit-identity test, tests/test_resolve_gpu.py); shared with the build for several turbine definitions
int g_res_levels = -1;
extern "C" void wfk_set_resolve_levels(int on) { g_res_levels = on ? 1 : 0; }
int g_res_helpers = -1;  // helper waves (WF_RES4_HELPERS / wfk_set_resolve_helpers): 0 never, 1 on lists expected to be short, 2 on every launch
extern "C" void wfk_set_resolve_helpers(int mode) { g_res_helpers = mode < 0 ? 0 : mode > 2 ? 2 : mode; }
// which kernels are enqueued (wfk_launch_resolve): 0 the four-wave kernel wherever it is the faster one (default), 1 "both" —
// rounds 3-5's rule, by the list's length (WF_RESOLVE_POLICY=both seeds it; tests and A/B runs switch it here)
int g_res_policy = -1;
extern "C" void wfk_set_resolve_policy(int both) { g_res_policy = both ? 1 : 0; }
extern "C" int wfk_res_level_stats(unsigned long long* out, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(wf_res_lvl_stat), sizeof(wf_res_lvl_stat));
  if (e == hipSuccess && reset) {
    unsigned long long z[8] = {};
    e = hipMemcpyToSymbol(HIP_SYMBOL(wf_res_lvl_stat), z, sizeof(z));
  }
  return (int)e;
}
#endif


#endif  // RES_PART == 2
#if defined(WF_RES_STAMP) && !RES_MT && RES_PART == 1
extern "C" int wfk_res_stamps(unsigned long long* out, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(wf_res_stamp), sizeof(wf_res_stamp));
  if (e == hipSuccess && reset) {
    unsigned long long z[8] = {};
    e = hipMemcpyToSymbol(HIP_SYMBOL(wf_res_stamp), z, sizeof(z));
  }
  return (int)e;
}
#endif

#if defined(WF_RES_STAMP) && !RES_MT && RES_PART == 2
extern "C" int wfk_res4_stamps(unsigned long long* out, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(wf_res4_stamp), sizeof(wf_res4_stamp));
  if (e == hipSuccess && reset) {
    unsigned long long z[16] = {};
    e = hipMemcpyToSymbol(HIP_SYMBOL(wf_res4_stamp), z, sizeof(z));
  }
  return (int)e;
}
extern "C" int wfk_res4_fn_stamps(unsigned long long* out, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(wf_res4_fstamp), sizeof(wf_res4_fstamp));
  if (e == hipSuccess && reset) {
    unsigned long long z[16] = {};
    e = hipMemcpyToSymbol(HIP_SYMBOL(wf_res4_fstamp), z, sizeof(z));
  }
  return e == hipSuccess ? 0 : -1;
}
extern "C" int wfk_res4_level_stamps(unsigned long long* out, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(wf_res4_lstamp), sizeof(wf_res4_lstamp));
  if (e == hipSuccess && reset) {
    unsigned long long z[20] = {};
    e = hipMemcpyToSymbol(HIP_SYMBOL(wf_res4_lstamp), z, sizeof(z));
  }
  return (int)e;
}
#endif

#if RES_PART == 2
#if RES_MT
extern int g_res_levels, g_res_helpers;
#endif
// the four-wave kernel's launch (called by wfk_launch_resolve of part 1): how many farms one residency holds -> *max4_out;
// launched when `launch & 1` (the caller decides: always for a device-side count, for `all` when B fits); `launch & 2`: the list
// is every farm of the batch (its length is known: B)
extern "C" hipError_t wfk_launch_resolve4(const WfResolveConsts* c, const WfResolveArgs* a, int B, int n_cu, int launch, int* max4_out,
                                          int any_count, hipStream_t s) {
  const int n_pad = (c->N + 1) & ~1;  // (keeps the int arrays behind the doubles aligned)
  const size_t dyn4 = (RES4_DYN_BYTES(n_pad) + 15) & ~(size_t)15;
  if (g_res_levels < 0) { const char* e = getenv("WF_RES_LEVELS"); g_res_levels = e ? atoi(e) : 1; }
  const int levels = g_res_levels;
  const size_t lds4 = dyn4 + sizeof(Res4Shared);
  int per_cu = (int)((160 * 1024) / lds4);
  {  // ... as the runtime counts it (LDS is handed out in granules: three blocks of 53 872 bytes do NOT fit 160 KB), asked once per size
    static int occ4[2048];  // (blocks of 256 threads per CU) + 1; the same value whoever writes it
    if (n_pad < 2048) {
      if (occ4[n_pad] == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, wf_resolve4_kernel, 256, dyn4) != hipSuccess) { (void)hipGetLastError(); nb = per_cu; }
        occ4[n_pad] = nb + 1;
      }
      if (occ4[n_pad] - 1 < per_cu) per_cu = occ4[n_pad] - 1;
    }
  }
  if (per_cu > WF_RES4_OCC) per_cu = WF_RES4_OCC;
  // One residency: as many blocks per CU as LDS and registers hold (HornsRev1 / 2: three; up to 51 turbines: four).  (Rounds 3-5:
  // at most two — the kernel had 256 VGPRs — and beyond them the one-wave kernel won: 680 farms 1.27 ms against 1.1.  At 128 VGPRs a
  // list of 600-800 farms is ONE round of this kernel with every SIMD shared by 2-3 farms' waves: issue-bound float64 work, the idle
  // slots of one farm's latency chain filled by another's; profiles/r06_res4_residency_ab.txt.)
  static const int per_cu_cap = [] { const char* e = getenv("WF_RES4_PER_CU"); return e ? atoi(e) : WF_RES4_OCC; }();
  if (per_cu > per_cu_cap) per_cu = per_cu_cap;
  // helper waves (see the kernel): a launch of 512 threads where the list is expected to be short; two such blocks fit a CU
  if (g_res_helpers < 0) { const char* e = getenv("WF_RES4_HELPERS"); g_res_helpers = e ? atoi(e) : 1; }
  static const int helpers_max_env = [] { const char* e = getenv("WF_RES4_HELPERS_MAX"); return e ? atoi(e) : -1; }();  // (experiments)
  const int helpers_max = !g_res_helpers ? 0 : helpers_max_env >= 0 ? helpers_max_env : 2 * n_cu;
  // 0 never, 1 by what is known of the list (every farm: B; flagged farms: the caller's hint from the previous launch), 2 always
  const bool wide = helpers_max > 0 && (g_res_helpers == 2 || ((launch & 2) ? B <= helpers_max : a->wide_hint != 0));
  if (wide && per_cu > 2) per_cu = 2;
  const int max4 = per_cu >= 1 ? n_cu * per_cu : 0;
  *max4_out = max4;
  if (!(launch & 1) || max4 <= 0) return hipSuccess;
  const int grid4 = B < max4 ? B : max4;
  // any_count: this launch is the only one behind the step — its persistent blocks walk a list of any length
  hipLaunchKernelGGL(wf_resolve4_kernel, dim3(grid4), dim3(wide ? 64 * RES4_MAX_WAVES : 256), dyn4, s, *c, *a, n_pad,
                     any_count ? 0x7fffffff : max4, levels, helpers_max);
  return hipGetLastError();
}
#endif  // RES_PART == 2

#if RES_PART == 1
#if RES_MT
#define wfk_launch_resolve4 wfk_launch_resolve4_mt
#endif
extern int g_res_policy;
extern "C" hipError_t wfk_launch_resolve4(const WfResolveConsts* c, const WfResolveArgs* a, int B, int n_cu, int launch, int* max4_out,
                                          int any_count, hipStream_t s);
extern "C" hipError_t wfk_launch_resolve(const WfResolveConsts* c, const WfResolveArgs* a, int B, int all, int* raw_flags,
                                         int n_cu, hipStream_t s) {
  hipError_t e = hipSuccess;
  if (all) {
    hipLaunchKernelGGL(wf_list_all_kernel, dim3((B + 255) / 256), dim3(256), 0, s, a->flags, B, a->list, a->count, raw_flags);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  int max4 = 0;
  if ((e = wfk_launch_resolve4(c, a, B, n_cu, all ? 2 : 0, &max4, 0, s)) != hipSuccess) return e;
  // ONE launch behind a step (round 6): the four-wave kernel serves a flagged list of ANY length — its persistent blocks walk the
  // list, up to max4 farms resident at once.  With level stages it is the faster kernel at every count (2 011 flagged HornsRev2
  // farms: + 1.07 ms against the one-wave kernel's + 1.62, profiles/r06_four_wave_always_ab.txt), and an empty list costs one
  // dispatch instead of two.  (Rounds 3-5 enqueued both kernels and let each read the count on the device; WF_RESOLVE_POLICY=both
  // restores that for A/B runs.)  The one-wave kernel remains for mode 2 at batches beyond one residency (`all`).
  if (g_res_policy < 0) { const char* e = getenv("WF_RESOLVE_POLICY"); g_res_policy = (e && std::string(e) == "both") ? 1 : 0; }
  const bool both = g_res_policy == 1;
  // mode 2 (every farm) at a batch beyond a residency: the four-wave kernel too where farms have levels to offer — 16 turbines
  // and more (HornsRev1 x 65536: 43.2 -> 25.9 ms, HornsRev2 25.4 -> 18.0, Ormonde 14.5 -> 11.5; a 7-turbine row in line with the
  // wind has none and is 25 % faster on the one-wave kernel: profiles/r06_mode2_ab.txt)
  const bool all4 = all && c->N >= 16 && !both;
  const bool only4 = (!all || all4) && max4 > 0 && !both;
  if (max4 > 0 && (!all || B <= max4 || all4)) {
    if ((e = wfk_launch_resolve4(c, a, B, n_cu, all ? 3 : 1, &max4, only4 ? 1 : 0, s)) != hipSuccess) return e;
  }
  if (only4) return hipSuccess;
  if (!all || B > max4) {
    // persistent one-wave blocks over the compacted list: enough to fill the chip several times over, never more than farms
    // (two waves per SIMD hold 8 farms per CU; more blocks than that only cost launch time when the list is empty — 20 us
    // for 8192 blocks that return at once, 7 us for 2048)
    // (every farm — B is known here — runs 27 % faster from a grid of 32 blocks per CU than from 8 persistent ones: 53.5 against
    // 73.7 ms at 65536 HornsRev1 farms, tools/gridab.sh; the flagged list is short and an empty launch should be cheap)
    const int n_pad = (c->N + 1) & ~1;
    const int per_cu_grid = all ? 4 * WF_RES_GRID_PER_CU : WF_RES_GRID_PER_CU;
    const int grid = B < n_cu * per_cu_grid ? B : n_cu * per_cu_grid;
    const size_t dyn = sizeof(double) * RES_TS * (size_t)n_pad + sizeof(int) * (size_t)n_pad;
    hipLaunchKernelGGL(wf_resolve_kernel, dim3(grid), dim3(64), dyn, s, *c, *a, n_pad, max4 + 1);
  }
  return hipGetLastError();
}
#endif  // RES_PART == 1
