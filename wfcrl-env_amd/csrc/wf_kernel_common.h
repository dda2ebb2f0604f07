// wf_kernel_common.h — device helpers shared by the step kernels (wf_kernels.hip: the register-slot kernel;
// wf_kernels_ll.hip: the one-block-at-a-time kernel with the per-farm source log).  Anonymous namespace: every
// translation unit that includes this gets its own copy.
#pragma once
#include <hip/hip_runtime.h>

#include "wf_device.h"

namespace {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kDeg2Rad = kPi / 180.0f;
constexpr float kRad2Deg = 180.0f / kPi;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kGs = 0.8493218002880191f;  // sqrt(log2(e) / 2)

__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float flog2(float x) { return __builtin_amdgcn_logf(x); }

// LDS-DMA by hand (round 5): one wave-instruction moves 64 x 16 bytes from global memory (lane's address g) to 1 KiB of LDS at
// l (wave-uniform).  Issued through inline assembly the transfer is INVISIBLE to the compiler's wait-count bookkeeping — which
// is the point: the builtin (__builtin_amdgcn_global_load_lds) made it guard the first LDS read after ANY such transfer with
// s_waitcnt vmcnt(0), so the "prefetch" of the next chunk was waited for at the top of the chunk that issued it, and every
// register prefetch of a log record at the top of the next loop iteration (rounds 2-4: DESIGN_HISTORY "what the replay loop was
// actually waiting for").  Now the kernel waits where the data is needed: wf_dma_wait() in front of the chunk's closing barrier.
// (Compiler-issued loads that are waited for in between may wait for an older transfer with them — in-order counter — never
// for too little.)  agent: sc1, served by the L2 past the vector L1.
// (the LDS base reaches M0 through an "{m0}"-constrained operand — ADVICE r5: the compiler owns M0 too (movrel / s_set_gpr_idx,
// its own LDS-DMA and GWS builtins), so it writes the register itself and knows what it holds; clobber lists may not name M0)
__device__ __forceinline__ void lds_dma16(const void* g, void* l, bool agent) {
  const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)l);
  if (agent) asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1" : : "v"(g), "{m0}"(la) : "memory");
  else asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "{m0}"(la) : "memory");
}
__device__ __forceinline__ void wf_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// cube root, sign-aware like np.cbrt (unphysically tight layouts can drive grid velocities negative, and the
// reference keeps computing): exp2(log2|x|/3) polished by one Newton step (rel. err ~1e-7); cbrt(0) = 0
__device__ __forceinline__ float fcbrt_pos(float x) {
  const float ax = fabsf(x);
  float y = fexp2(flog2(ax) * (1.0f / 3.0f));
  const float y2 = y * y;
  y = y - (y2 * y - ax) * frcp(3.0f * y2);
  y = (ax > 1.0e-30f) ? y : 0.0f;
  return copysignf(y, x);
}

// atan(r) and asin(x) for the small arguments this model produces (|V/U| ~ 1e-2, |val| ~ 3e-2): 7-term odd series,
// abs. err < 1e-10 for |r| <= 0.25 resp. |x| <= 0.3; callers fall back to libm (wave-uniform) beyond that.
__device__ __forceinline__ float atan_small(float r) {
  const float r2 = r * r;
  return r * fmaf(r2, fmaf(r2, fmaf(r2, fmaf(r2, fmaf(r2, fmaf(r2, 0.0769230769f, -0.0909090909f), 0.1111111111f),
                                              -0.1428571429f), 0.2f), -0.3333333333f), 1.0f);
}
__device__ __forceinline__ float asin_small(float x) {
  const float x2 = x * x;
  return x * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 0.0173527644f, 0.0223721591f), 0.0303819444f),
                                              0.0446428571f), 0.075f), 0.1666666667f), 1.0f);
}

struct TableLds {
  float knot[WF_TABLE_PAD], ct[WF_TABLE_PAD], cts[WF_TABLE_PAD], pw[WF_TABLE_PAD], pws[WF_TABLE_PAD];
  unsigned char bucket[WF_BUCKETS];
};

// index of the last knot <= v (v inside the table range)
__device__ __forceinline__ int table_segment(const WfConsts& c, const TableLds& T, float v) {
  int b = (int)((v - c.bucket_x0) * c.bucket_h_inv);
  b = min(max(b, 0), WF_BUCKETS - 1);
  int j = T.bucket[b];
  for (int p = 0; p < c.max_probe; ++p) {  // usually 0-1 rounds: leave as soon as no lane moves
    const bool mv = T.knot[j + 1] <= v;
    if (!__any(mv)) break;
    j += mv ? 1 : 0;
  }
  return min(j, c.n_table - 2);
}

// scipy interp1d(linear, fill_value=(lo,hi)) on the LDS copy of the table
// risk: WF_RISK_THRUST_RAMP when the turbine sits on a segment of the thrust table with v |dCt/dv| > ct_kappa;
//       WF_RISK_THRUST_UNITY when its thrust coefficient is within 0.005 of 1 (a user table; nrel_5MW peaks at 0.99): the wake
//       amplitude Ct / (1 + sqrt(1 - Ct)) then loses 1e-5 to the cancellation in 1 - Ct, and behind such a turbine the velocity
//       u = Uinit (1 - deficit) is a tenth or less of the deficit — the error reaches 1e-4 of u and there is no float32 bound
//       (layout fuzzer, round 4: a table clipped at 0.9999 over 3-11 m/s; TI off by 0.7 on a dense farm).  Farms with this
//       flag are ALWAYS solved again in float64, whatever wf_set_risk_resolve says (wf_dispatch.hip: launch_step).
__device__ __forceinline__ float table_ct(const WfConsts& c, const TableLds& T, float v, unsigned& risk) {
  int j = table_segment(c, T, v);
  float r = fmaf(T.cts[j], v - T.knot[j], T.ct[j]);
  const bool in_range = v >= T.knot[0] && v <= T.knot[c.n_table - 1];
  const bool inside = in_range && r > 0.0001f && r < 0.9999f;
  risk = ((inside && fabsf(T.cts[j]) * v > c.ct_kappa) ? (unsigned)WF_RISK_THRUST_RAMP : 0u) |
         ((in_range && r > 0.995f) ? (unsigned)WF_RISK_THRUST_UNITY : 0u);
  r = (v < T.knot[0]) ? 0.0001f : r;
  r = (v > T.knot[c.n_table - 1]) ? 0.9999f : r;
  return fminf(fmaxf(r, 0.0001f), 0.9999f);
}
__device__ __forceinline__ float table_pw(const WfConsts& c, const TableLds& T, float v, float& slope) {
  int j = table_segment(c, T, v);
  float r = fmaf(T.pws[j], v - T.knot[j], T.pw[j]);
  const bool outside = v < T.knot[0] || v > T.knot[c.n_table - 1];
  slope = outside ? 0.0f : T.pws[j];
  return outside ? 0.0f : r;
}

// Register-resident per-turbine state of one lane: S target slots x (9 + 9 + 9 + 3) floats.
template <int S, int NE = 6>
struct Slots {
  // SOSFS [A.3-7]: FLORIS sums (deficit * Uinit(z))^2; Uinit(z_k) is the same for every source, so the sum of
  // deficit^2 is kept instead, and the Gaussian deficit is even in z - HH: rows k = 0 and k = 2 always hold the
  // same value.  esq[j*2] = rows 0 and 2 of column j, esq[j*2 + 1] = row 1:  u(j,k) = Uinit_k * (1 - sqrt(esq)).
  float esq[S][NE];  // NE = 9 with wind veer: [3 j + k], the rotated Gaussian is not even in z - HH
  float V[S][9], W[S][9];
  float TI[S][3];   // per grid column j (independent of k)            [A.3-8]
};

// Per-wave sorted geometry + commanded yaw, staged in LDS once (lane-private reads for targets,
// group-broadcast reads for the source).
template <int EPW, int NP, bool WITH_XY>
struct GeoLds {
  double x[WITH_XY ? EPW : 1][WITH_XY ? NP : 1];  // sorted x' (float64: the sign of dx decides every mask)
  double yd[WITH_XY ? EPW : 1][WITH_XY ? NP : 1]; // sorted y' (float64: the 2 D lateral gate is decided on it and the lateral offsets are taken from it; table mode takes dx, dy, gates from the pair table)
  float yaw[EPW][NP]; // commanded yaw in sorted order, degrees
  float cg[EPW][NP], sg[EPW][NP];  // cos / sin of the commanded yaw (evaluated once per turbine)
};

// Source-side constants of the deflection + deficit models for one grid column j [A.3-3, A.3-6]
struct ColConsts {
  float x0d, kyd, d0, pj;   // deflection: near-wake length, expansion rate, delta0, far-wake log prefactor
  float x0v, ix0v, kyv;     // deficit:    near-wake length, its reciprocal, expansion rate
};
struct SrcConsts {
  float sy0d, sz0d, inv_s0d, lnA, lnB, sM, tan_th0;  // deflection
  float sy0v, snw, kdef;                             // deficit
};

// deflection + deficit of one target column: returns (e1, e0) = deficit at k = 1 and at k = 0, 2
__device__ __forceinline__ void column_deficit(const WfConsts& c, const SrcConsts& sc, const ColConsts& cc, float dx,
                                               float ylat, float lin, float amp_on, float& e1, float& e0) {
  // deflection (TI before mixing) [A.3-3]
  const float xs = fmaxf(dx - cc.x0d, 0.0f);
  const float syd = fmaf(cc.kyd, xs, sc.sy0d), szd = fmaf(cc.kyd, xs, sc.sz0d);
  const float s = fsqrt(syd * szd * sc.inv_s0d);
  const float arg = sc.lnA * fmaf(1.6f, s, -sc.sM) * frcp(sc.lnB * fmaf(1.6f, s, sc.sM));
  const float d_far = fmaf(cc.pj, flog2(arg), cc.d0);
  const float delta = ((dx > cc.x0d) ? d_far : dx * sc.tan_th0) + lin;
  // deficit (TI after mixing) [A.3-6]
  const bool far = dx >= cc.x0v;
  const float up = dx * cc.ix0v;
  const float xf = dx - cc.x0v;
  const float sy = far ? fmaf(cc.kyv, xf, sc.sy0v) : fmaf(up, sc.sy0v - sc.snw, sc.snw);
  const float sz = far ? fmaf(cc.kyv, xf, c.sz0v) : fmaf(up, c.sz0v - sc.snw, sc.snw);
  const float isy = frcp(sy), isz = frcp(sz);
  const float xarg = sc.kdef * isy * isz;
  const float C = (xarg >= 1.0f) ? 1.0f : xarg * frcp(1.0f + fsqrt(fmaxf(1.0f - xarg, 0.0f)));
  const float yy = (ylat - delta) * isy;
  const float zz = c.off[2] * isz;
  const float amp = amp_on * C;
  e1 = amp * fexp2(-0.5f * kLog2e * yy * yy);
  e0 = e1 * fexp2(-0.5f * kLog2e * zz * zz);
}

// The part of column_deficit that depends on the column's constants only (not on its lateral offset): one evaluation
// serves every grid column that shares the constants (split-TI sources: usually two of the three columns agree).
struct ColWake { float delta, isy, amp, ez; };
__device__ __forceinline__ ColWake column_wake(const WfConsts& c, const SrcConsts& sc, const ColConsts& cc, float dx, float lin,
                                               float amp_on) {
  const float xs = fmaxf(dx - cc.x0d, 0.0f);
  const float syd = fmaf(cc.kyd, xs, sc.sy0d), szd = fmaf(cc.kyd, xs, sc.sz0d);
  const float s = fsqrt(syd * szd * sc.inv_s0d);
  const float arg = sc.lnA * fmaf(1.6f, s, -sc.sM) * frcp(sc.lnB * fmaf(1.6f, s, sc.sM));
  const float d_far = fmaf(cc.pj, flog2(arg), cc.d0);
  ColWake w;
  w.delta = ((dx > cc.x0d) ? d_far : dx * sc.tan_th0) + lin;
  const bool far = dx >= cc.x0v;
  const float up = dx * cc.ix0v;
  const float xf = dx - cc.x0v;
  const float sy = far ? fmaf(cc.kyv, xf, sc.sy0v) : fmaf(up, sc.sy0v - sc.snw, sc.snw);
  const float sz = far ? fmaf(cc.kyv, xf, c.sz0v) : fmaf(up, c.sz0v - sc.snw, sc.snw);
  const float isz = frcp(sz);
  w.isy = frcp(sy);
  const float xarg = sc.kdef * w.isy * isz;
  const float C = (xarg >= 1.0f) ? 1.0f : xarg * frcp(1.0f + fsqrt(fmaxf(1.0f - xarg, 0.0f)));
  const float zz = c.off[2] * isz;
  w.amp = amp_on * C;
  w.ez = fexp2(-0.5f * kLog2e * zz * zz);
  return w;
}
// ... and the column's own part: (e1, e0) = deficit at k = 1 and at k = 0, 2 (the arithmetic of column_deficit)
__device__ __forceinline__ void column_rows(const ColWake& w, float ylat, float& e1, float& e0) {
  const float yy = (ylat - w.delta) * w.isy;
  e1 = w.amp * fexp2(-0.5f * kLog2e * yy * yy);
  e0 = e1 * w.ez;
}

// The same with wind veer [FLORIS gauss.py rCalt]: r = a yy^2 - 2 b yy zz + c zz^2 with
//   a = cos^2/(2 sy^2) + sin^2/(2 sz^2),  b = sin(2 phi)/4 (1/sz^2 - 1/sy^2),  c = sin^2/(2 sy^2) + cos^2/(2 sz^2):
// the rows zz = -+ D/4 carry the cross term -+ 2 b yy D/4.  Returns the deficits of rows k = 0, 1, 2.
__device__ __forceinline__ void column_deficit_veer(const WfConsts& c, const SrcConsts& sc, const ColConsts& cc, float dx,
                                                    float ylat, float lin, float amp_on, float& ea, float& e1, float& eb) {
  const float xs = fmaxf(dx - cc.x0d, 0.0f);
  const float syd = fmaf(cc.kyd, xs, sc.sy0d), szd = fmaf(cc.kyd, xs, sc.sz0d);
  const float s = fsqrt(syd * szd * sc.inv_s0d);
  const float arg = sc.lnA * fmaf(1.6f, s, -sc.sM) * frcp(sc.lnB * fmaf(1.6f, s, sc.sM));
  const float d_far = fmaf(cc.pj, flog2(arg), cc.d0);
  const float delta = ((dx > cc.x0d) ? d_far : dx * sc.tan_th0) + lin;
  const bool far = dx >= cc.x0v;
  const float up = dx * cc.ix0v;
  const float xf = dx - cc.x0v;
  const float sy = far ? fmaf(cc.kyv, xf, sc.sy0v) : fmaf(up, sc.sy0v - sc.snw, sc.snw);
  const float sz = far ? fmaf(cc.kyv, xf, c.sz0v) : fmaf(up, c.sz0v - sc.snw, sc.snw);
  const float isy = frcp(sy), isz = frcp(sz);
  const float xarg = sc.kdef * isy * isz;
  const float C = (xarg >= 1.0f) ? 1.0f : xarg * frcp(1.0f + fsqrt(fmaxf(1.0f - xarg, 0.0f)));
  const float iy2 = kGs * kGs * isy * isy, iz2 = kGs * kGs * isz * isz;  // log2(e) / (2 sigma^2)
  const float yy = ylat - delta;
  const float q = c.off[2];
  const float A = fmaf(c.veer_c2, iy2, c.veer_s2 * iz2), Cz = fmaf(c.veer_s2, iy2, c.veer_c2 * iz2);
  const float Bq = c.veer_bq * (iz2 - iy2) * yy;  // 2 b yy q (in log2 units)
  const float r1 = A * yy * yy, rq = fmaf(Cz, q * q, r1);
  // one exponential per row on the whole (positive definite) quadratic form: the factors exp(-+ 2 b yy q) overflow on
  // their own where the Gaussian itself is zero (0 x inf behind a turbine below cut-in, whose near-wake sigma is < 1 m)
  const float amp = amp_on * C;
  e1 = amp * fexp2(-r1);
  ea = amp * fexp2(-fmaxf(rq + Bq, 0.0f));  // zz = -q:  r = a yy^2 + 2 b yy q + c q^2
  eb = amp * fexp2(-fmaxf(rq - Bq, 0.0f));  // zz = +q
}

// The pair-coefficient record of (source i, target t) — sorted indices of one wind direction — at `o` (WF_PAIR_STRIDE
// floats, layout in wf_device.h).  Everything of the transverse-velocity pass [A.3-4] that does not depend on the farm's
// state, plus the pair's float64 decisions; float64 arithmetic, rounded once.
__device__ inline void wf_pair_record(const WfPairConsts& pc, const double* __restrict__ gx, const double* __restrict__ gy,
                                      int i, int t, float* __restrict__ o) {
  if (t >= pc.N) {  // padding target of the kernel variant: permanently "upstream"
    for (int q = 0; q < WF_PAIR_STRIDE; ++q) o[q] = 0.0f;
    o[WF_PAIR_DX] = -1.0f;
    return;
  }
  const double dx = gx[t] - gx[i];
  const double dy = gy[t] - gy[i];
  if (dx < 0.0) {  // upstream target: nothing reaches it; only the sign of dx is ever looked at
    for (int q = 0; q < WF_PAIR_STRIDE; ++q) o[q] = 0.0f;
    o[WF_PAIR_DX] = (float)dx;
    o[WF_PAIR_DY] = (float)dy;
    return;
  }
  const double R = pc.D / 2.0;
  const double hs[3] = {pc.HH + R, pc.HH - R, pc.HH};  // top tip vortex, bottom tip vortex, wake rotation
  for (int j = 0; j < 3; ++j) {
    const double yL = dy + pc.off[j] + pc.num_eps;
    for (int k = 0; k < 3; ++k) {
      const double z = pc.HH + pc.off[k];
      const double dec = 1.0 / (pc.decay_a[k] * dx + 1.0);
      double cv[3], cw[3];
      for (int v = 0; v < 3; ++v) {
        const double zc = z - hs[v] + pc.num_eps, zm = z + hs[v] + pc.num_eps;
        const double r = yL * yL + zc * zc, rm = yL * yL + zm * zm;
        const double T = (1.0 - exp(-r / pc.eps2)) / r, Tm = (1.0 - exp(-rm / pc.eps2)) / rm;
        cv[v] = dec * (zc * T - zm * Tm);
        cw[v] = -yL * dec * (T - Tm);
      }
      float* q = o + (3 * j + k) * 4;  // {aV, bV, aW, bW}: Gt = gam_top*Gy and Gb = -gam_bot*Gy folded
      q[0] = (float)(pc.gam_top * cv[0] - pc.gam_bot * cv[1]);
      q[1] = (float)cv[2];
      q[2] = (float)(pc.gam_top * cw[0] - pc.gam_bot * cw[1]);
      q[3] = (float)cw[2];
    }
  }
  const double dxp = (dx > 0.1) ? dx : dx + 1.0;  // Crespo-Hernandez distance with FLORIS' masks [A.3-8]
  for (int q = 39; q < WF_PAIR_STRIDE; ++q) o[q] = 0.0f;
  // reach of the wake-added TI exactly as FLORIS tests it:  x_t <= x_i + 15 D  in float64 (on regular grids whole
  // multiples of D sit on this threshold and the rounding of the rotation decides)
  o[WF_PAIR_TIPOW] = (gx[t] <= gx[i] + pc.fifteenD) ? (float)pow(dxp / pc.D, pc.ch_down) : 0.0f;
  o[WF_PAIR_DX] = (float)dx;
  o[WF_PAIR_DY] = (float)dy;
  // the other two discontinuities of the pair, decided on the float64 coordinates in FLORIS' own form [A.3-6, A.3-8]:
  // bit j: grid column j inside the lateral gate |y_i - (y_t + off_j)| < 2 D; bit 3: X > x_i + 0.1 (the deficit is on)
  int bits = 0;
  for (int j = 0; j < 3; ++j) bits |= (fabs(gy[i] - (gy[t] + pc.off[j])) < pc.twoD) ? (1 << j) : 0;
  bits |= (gx[t] > gx[i] + 0.1) ? 8 : 0;
  o[WF_PAIR_BITS] = __int_as_float(bits);
}

}  // namespace
