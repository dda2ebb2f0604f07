// wf_resolve.hip — float64 farm solve on the device for the farms the float32 kernels flag (include/wfstep.h:
// wf_set_risk_resolve), and for every farm of a model the float32 kernels do not implement (wind_veer != 0).
//
// The reference evaluates the whole path in float64 (reference wfcrl/interface.py:564 `fi.calculate_wake`, FLORIS 3.5
// sequential solver; SURVEY.md Appendix A, tags [A.x] below).  The float32 step kernels cannot reproduce a float64
// decision inside their own rounding band (the overlap count "deficit * Uinit > 0.05" [A.3-8]) nor the result on an
// ill-conditioned piece of the turbine tables; they raise a per-farm risk flag there.  This file makes the 1e-4 contract
// unconditional: the flagged farms are compacted on the device (no host round trip) and solved again in float64, the
// comparison taken exactly as FLORIS takes it, and their outputs overwritten.
//
// Mapping: one farm per workgroup, one thread per (sorted) target turbine, the turbine's state — 9 wake deficits, 9 V,
// 9 W, 3 column TIs, all float64 — in registers for the whole solve; sources i = 0 .. N-1 in sorted order, the source's
// rotor means broadcast through LDS (two barriers per source: its state, then its transverse velocities including its
// own contribution, which the yaw-added recovery [A.3-5] needs before the deficit [A.3-6]).  The per-source constants
// are derived redundantly by every thread.  Persistent blocks walk the compacted farm list.
// Only exactness-preserving algebra is used (vortex core 1 - exp(-(y^2+z^2)/eps^2) with the z factor a constant);
// sums are taken in a different order than NumPy takes them: results agree with the CPU oracle to ~1e-13 relative.
#include <hip/hip_runtime.h>

#include "wf_device.h"
#include "wf_resolve.h"

namespace {

constexpr double kDeg = 3.14159265358979323846 / 180.0;
constexpr double kTwoPi = 2.0 * 3.14159265358979323846;

__device__ __forceinline__ double cosd(double a) { return cos(a * kDeg); }
__device__ __forceinline__ double sind(double a) { return sin(a * kDeg); }

// scipy interp1d(linear, bounds_error=False, fill_value=(lo, hi)) on the LDS copy of a table column
__device__ inline double interp_fill(double xq, int n, const double* xs, const double* ys, double lo, double hi) {
  if (xq < xs[0]) return lo;
  if (xq > xs[n - 1]) return hi;
  if (xq == xs[n - 1]) return ys[n - 1];
  int j = 0;  // last knot <= xq, at most n - 2: bisection (the dependent LDS probes of a linear scan cost 5 k cycles per call)
  for (int step = 32; step >= 1; step >>= 1) {
    const int k = j + step;
    if (k <= n - 2 && xq >= xs[k]) j = k;
  }
  const double slope = (ys[j + 1] - ys[j]) / (xs[j + 1] - xs[j]);
  return slope * (xq - xs[j]) + ys[j];
}

}  // namespace

// farms with a nonzero risk flag -> list (any order), count; raw = copy of the flags as the float32 kernels raised them
__global__ void wf_compact_flagged_kernel(const int* __restrict__ flags, int B, int all, int* __restrict__ list,
                                          int* __restrict__ count, int* __restrict__ raw) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int f = flags[b];
  raw[b] = f;
  if (f != 0 || all) list[atomicAdd(count, 1)] = b;
}

// 1 / x to float64 accuracy from the hardware estimate and two Newton steps (6 instructions; the compiler's IEEE division
// sequence is 11): the quotients here are all between normal, finite, nonzero operands
__device__ __forceinline__ double rcp64(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}

// Source-side constants of the deflection + deficit models for one grid column (TI of that column) [A.3-3, A.3-6]
struct ColD {
  double x0d, ix0d_rel, kyd, d0, pfar;  // deflection: near-wake end (absolute x), 1 / (x0 - x_i), expansion rate, delta0, far-wake log prefactor
  double x0v, ix0v_rel, kyv;            // deficit: near-wake end (absolute x), 1 / (x0 - x_i), expansion rate
};

#ifndef WF_RES_OCC
#define WF_RES_OCC 1  // waves per SIMD the register allocator is asked to make room for (HornsRev1, 1394 flagged farms: 1 -> 1.95 ms, 2 -> 2.3, 3 -> 2.6, 4 -> 3.7 with 300 spilled registers: tools/res_occ_sweep.sh)
#endif
template <int THREADS>
__global__ __launch_bounds__(THREADS, WF_RES_OCC) void wf_resolve_kernel(const WfResolveConsts c_arg, const WfResolveArgs a) {
  __shared__ double tws[WF_TABLE_PAD], tct[WF_TABLE_PAD], tpw[WF_TABLE_PAD];
  __shared__ double xsL[THREADS], ysL[THREADS], yawL[THREADS];
  __shared__ double bc[5], bc2[2];
  __shared__ double red[2][THREADS / 64];
  // the model constants are read from LDS where they are used (by value in the kernel arguments the compiler keeps all
  // ~130 of them in SGPRs across the source loop and spills: 1700 v_readlane / v_writelane in the first version)
  __shared__ WfResolveConsts c;
  const int t = threadIdx.x;
  if (t == 0) c = c_arg;
  const int N = c_arg.N;
  for (int k = t; k < c_arg.n_table; k += THREADS) {
    tws[k] = a.tab64[k];
    tct[k] = a.tab64[WF_TABLE_PAD + k];
    tpw[k] = a.tab64[2 * WF_TABLE_PAD + k];
  }
  const int n_list = *a.count;
  const bool live = t < N;
  const bool veer_on = c_arg.sin2_veer != 0.0;
  const bool mcore = c_arg.mirror_core != 0;

  for (int li = blockIdx.x; li < n_list; li += gridDim.x) {
    const int b = a.list[li];
    size_t gofs = 0;
    if (a.farm_group) gofs = (size_t)((a.farm_group[b] + a.shift) % a.mod) * N;
    else gofs = (size_t)b * a.geom_stride;
    const double ws = a.ws[(size_t)b * a.wind_stride];
    double wd = fmod(a.wd[(size_t)b * a.wind_stride], 360.0);  // reference interface.py:664 (Python's %)
    if (wd < 0.0) wd += 360.0;
    const float* yaw_b = (a.yaw_state ? a.yaw_state : a.yaw_in) + (size_t)b * N;
    int o = 0;
    double x_t = 0.0, y_t = 0.0, yaw_t = 0.0;
    __syncthreads();  // the previous farm's last readers of xsL / bc are done; c is in place
    if (live) {
      o = a.gidx[gofs + t];
      x_t = a.gx[gofs + t];
      y_t = a.gy[gofs + t];
      yaw_t = (double)yaw_b[o];
      xsL[t] = x_t; ysL[t] = y_t; yawL[t] = yaw_t;
    }
    const double D = c.D, eps2 = c.eps2, ieps2 = 1.0 / c.eps2;
    // inflow [A.2]
    double Uinit[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) Uinit[k] = ws * c.shearf[k];
    const double Uinf = ws * c.uinf1;
    // SOSFS [A.3-7]: FLORIS chains hypot(wake, deficit * Uinit); the sum of squares is kept and the root taken where the
    // velocity is needed (the source's own rotor mean, the outputs)
    double wake2[9], V[9], W[9], TI[3];
#pragma unroll
    for (int q = 0; q < 9; ++q) { wake2[q] = 0.0; V[q] = 0.0; W[q] = 0.0; }
#pragma unroll
    for (int j = 0; j < 3; ++j) TI[j] = c.amb;

    for (int i = 0; i < N; ++i) {
      asm volatile("" ::: "memory");  // the constants in LDS are re-read per source, not hoisted into registers for all of them
      if (t == i) {  // the source's state [A.3-1, A.3-2]
        double m3 = 0.0, vs = 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          const double u = Uinit[q % 3] - sqrt(wake2[q]);
          m3 += u * u * u;
          vs += V[q];
        }
        bc[0] = m3 / 9.0; bc[1] = vs / 9.0; bc[2] = TI[0]; bc[3] = TI[1]; bc[4] = TI[2];
      }
      __syncthreads();
      const double x_i = xsL[i], y_i = ysL[i], g = yawL[i];
      const double dx = x_t - x_i;
      const bool down = live && dx >= 0.0;  // ties (dx = 0) count as downstream for the transverse velocities [A.3-4]
      const bool wave_on = __any(down);     // (the source's own wave always is)
      double ct = 0.0, ai = 0.0, G_wr = 0.0, gam_top = 0.0, gam_bot = 0.0, cg = 1.0, sg = 0.0, ubar = 1.0;
      if (wave_on) {
        ubar = cbrt(bc[0]);
        sincos(g * kDeg, &sg, &cg);
        double ct_tab = interp_fill(ubar, c.n_table, tws, tct, 0.0001, 0.9999);
        ct_tab = fmin(fmax(ct_tab, 0.0001), 0.9999);
        ct = ct_tab * cg;
        ai = 0.5 / cg * (1.0 - sqrt(1.0 - ct * cg));
        G_wr = 0.25 * kTwoPi * D * (ai - ai * ai) * ubar / c.TSR;
        gam_top = (kTwoPi / 16.0) * D * c.vel_top * Uinf * ct;
        gam_bot = (kTwoPi / 16.0) * D * c.vel_bot * Uinf * ct;
      }

      // 4. transverse velocities (commanded yaw) on this thread's turbine: per grid column the 7 + 7 distinct vertical
      // offsets of the three vortices and their ground mirrors on the 3 x 3 grid (wf_device.h: zc / zm classes)
      if (c.sw_tv && wave_on && down) {
        const double sc = sg * cg;
        const double qd = c_arg.off[2], neps = c_arg.num_eps, twoHH = 2.0 * c_arg.HH;
        const double Gt = sc * gam_top / kTwoPi, Gb = -sc * gam_bot / kTwoPi, Gw = G_wr / kTwoPi;
        double dec[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) dec[k] = eps2 * rcp64(4.0 * (c.nu1[k] * ws) * dx / Uinf + eps2);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const double yL = (y_t + c.off[j] - y_i) + c.num_eps;
          const double yL2 = yL * yL;
          const double Ey = exp(-yL2 * ieps2);
          double Av[3] = {0.0, 0.0, 0.0}, Bw[3] = {0.0, 0.0, 0.0};
#pragma unroll
          for (int m = 0; m < 7; ++m) {
            // (offsets from the class index and three scalars; only the 7 core factors of the real vortices are tabulated:
            // 42 tabulated constants cost 84 registers across this loop)
            const double zc = (double)(m - 3) * qd + neps, zm = zc + twoHH;
            const double tr = (1.0 - Ey * c_arg.ezc[m]) * rcp64(yL2 + zc * zc);   // core / r of a real vortex at offset zc
            double tm = rcp64(yL2 + zm * zm);                                       // ... of a mirror vortex at zm
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
          }
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            V[j * 3 + k] += Av[k] * dec[k];
            const double w = -yL * Bw[k] * dec[k];
            W[j * 3 + k] += (w < 0.0) ? 0.0 : w;  // quirk (5) [A.6]
          }
        }
      }
      if (t == i) {
        double vs = 0.0, wsum = 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q) { vs += V[q]; wsum += W[q]; }
        bc2[0] = vs / 9.0; bc2[1] = wsum / 9.0;
      }
      __syncthreads();
      if (!wave_on) continue;
      const double TIs[3] = {bc[2], bc[3], bc[4]};
      // 2. secondary steering [A.3-2]: the three means on the source's own grid are geometry constants
      const double v_top = gam_top * c.k_top, v_bot = -gam_bot * c.k_bot, v_core = G_wr * c.k_core;
      double val = 2.0 * (bc[1] - v_core) / (v_top + v_bot);
      val = fmin(fmax(val, -1.0), 1.0);
      const double g_eff = c.sw_steer ? g + (0.5 * asin(val)) / kDeg : g;
      // 5. yaw-added recovery [A.3-5]
      double dTI = 0.0;
      {
        const double I = TIs[0];
        const double k_tke = (ubar * I) * (ubar * I) / (2.0 / 3.0);
        const double vbar = bc2[0], wbar = bc2[1];
        const double I_tot = sqrt((2.0 / 3.0) * 0.5 * (2.0 * k_tke + vbar * vbar + wbar * wbar)) / ubar;
        if (c.sw_yar) dTI = c.gch_gain * (I_tot - I);
      }
      if (t == i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) TI[j] += dTI;
      }
      // 3 + 6. deflection (TI before mixing, effective yaw) and deficit (TI after mixing, commanded yaw) [A.3-3, A.3-6]:
      // everything that depends on the source (and the column's TI) only
      // cosd(-g_eff) = cos(g + d), d = asin(val) / 2: half-angle formulas instead of a second cosine
      const double c2d = sqrt(fmax(1.0 - val * val, 0.0)), cd = sqrt(0.5 * (1.0 + c2d)), sd = 0.5 * val / cd;
      const double cgd = c.sw_steer ? cg * cd - sg * sd : cg;
      const double gd = -g_eff;
      const double s_cc = sqrt(1.0 - ct * cgd), s_c = sqrt(1.0 - ct);
      const double th0 = c.dm * (0.3 * (gd * kDeg) / cgd) * (1.0 - s_cc);
      const double tan_th0 = tan(th0);
      const double cgv = cg;  // cosd(-g)
      const double C0 = 1.0 - s_c;
      const double M0 = C0 * (2.0 - C0);
      const double E0 = C0 * C0 - c.e0c1 * C0 + c.e0c2;
      const double sM = sqrt(M0);
      const double sz0d = D * 0.5 * sqrt((ct * cgd / (2.0 * (1.0 - s_cc))) / (1.0 + s_c));
      const double sy0d = sz0d * cgd * c.cos_veer;
      const double is0d = 1.0 / (sy0d * sz0d);
      const double lnAB = (1.6 + sM) / (1.6 - sM);
      const double sz0v = D * 0.5 * sqrt((ct / (2.0 * (1.0 - s_c))) / (1.0 + s_c));
      const double sy0v = sz0v * cgv * c.cos_veer;
      const double snw = c.near_c * sqrt(ct / 2.0);
      const double kdef = ct * cgv * D * D / 8.0;
      const double ch_pref = c.ch_constant * exp(c.ch_ai * log(ai)) * c.ch_amb_pow;
      auto column = [&](double TIpre) {
        ColD k;
        const double x0r = D * cgd * (1.0 + s_cc) / (c.sqrt2 * (4.0 * c.defl_alpha * TIpre + 2.0 * c.defl_beta * (1.0 - s_c)));
        k.x0d = x0r + x_i;
        k.ix0d_rel = 1.0 / (k.x0d - x_i);
        k.kyd = c.defl_ka * TIpre + c.defl_kb;
        k.d0 = tan_th0 * (k.x0d - x_i);
        k.pfar = th0 * E0 / 5.2 * sqrt(sy0d * sz0d / (k.kyd * k.kyd * M0));
        const double TIq = TIpre + dTI;
        const double x0v = D * cgv * (1.0 + s_c) / (c.sqrt2 * (4.0 * c.alpha * TIq + 2.0 * c.beta * (1.0 - s_c)));
        k.x0v = x0v + x_i;
        k.ix0v_rel = 1.0 / (k.x0v - x_i);
        k.kyv = c.ka * TIq + c.kb;
        return k;
      };
      const bool same = (TIs[0] == TIs[1]) && (TIs[1] == TIs[2]);  // (uniform: a property of the source)
      if (!down) continue;
      const double lin = c.ad + c.bd * dx;
      // one target column: deflection -> delta; deficit -> amplitude and the Gaussian's 1 / (2 sigma^2)
      struct ColT { double delta, amp, isy2, isz2; };
      auto target_col = [&](const ColD& k) {
        ColT r;
        double d_near = (dx * k.ix0d_rel) * k.d0 + lin;
        if (!(x_t <= k.x0d)) d_near = 0.0;  // [x >= x_i] holds here
        double d_far = 0.0;
        if (x_t > k.x0d) {
          const double sy = k.kyd * (x_t - k.x0d) + sy0d, sz = k.kyd * (x_t - k.x0d) + sz0d;
          const double s = sqrt(sy * sz * is0d);
          const double ln_arg = lnAB * (1.6 * s - sM) * rcp64(1.6 * s + sM);
          d_far = k.d0 + k.pfar * log(ln_arg) + lin;
        }
        r.delta = d_near + d_far;
        r.amp = 0.0; r.isy2 = 0.0; r.isz2 = 0.0;
        double sy = 0.0, sz = 0.0;
        bool on = false;
        if (x_t > x_i + 0.1 && x_t < k.x0v) {  // the masks as FLORIS takes them on the coordinates
          const double up = dx * k.ix0v_rel, dn = (k.x0v - x_t) * k.ix0v_rel;
          sy = dn * snw + up * sy0v;
          sz = dn * snw + up * sz0v;
          on = true;
        } else if (x_t >= k.x0v) {
          sy = k.kyv * (x_t - k.x0v) + sy0v;
          sz = k.kyv * (x_t - k.x0v) + sz0v;
          on = true;
        }
        if (on) {
          const double isy = rcp64(sy), isz = rcp64(sz);
          double dd = 1.0 - kdef * isy * isz;
          dd = fmin(fmax(dd, 0.0), 1.0);
          r.amp = 1.0 - sqrt(dd);
          r.isy2 = 0.5 * isy * isy;
          r.isz2 = 0.5 * isz * isz;
        }
        return r;
      };
      int cnt = 0;
      const double q2 = c.off[2] * c.off[2];
      ColT Tc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (j == 0 || !same) Tc = target_col(column(TIs[j]));  // (one column serves all three when the source's TIs agree)
        const double yy = (y_t + c.off[j]) - y_i - Tc.delta;
        double def[3];
        if (!veer_on) {  // r = yy^2 / (2 sy^2) + zz^2 / (2 sz^2), zz = -q, 0, +q
          const double e1 = Tc.amp * exp(-(yy * yy) * Tc.isy2);
          const double e0 = e1 * exp(-q2 * Tc.isz2);
          def[0] = e0; def[1] = e1; def[2] = e0;
        } else {  // FLORIS rCalt [gauss.py]: the Gaussian rotated by the veer angle
          const double ca = c.cos2_veer * Tc.isy2 + c.sin2_veer * Tc.isz2;
          const double cb = 0.5 * c.sin_2veer * (Tc.isz2 - Tc.isy2);
          const double cc = c.sin2_veer * Tc.isy2 + c.cos2_veer * Tc.isz2;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const double zz = c.off[k];
            def[k] = Tc.amp * exp(-(ca * yy * yy - 2.0 * cb * yy * zz + cc * zz * zz));
          }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double dU = def[k] * Uinit[k];
          if (dU > c.overlap_thr) ++cnt;  // the comparison as FLORIS takes it [A.3-8]
          wake2[j * 3 + k] = fma(dU, dU, wake2[j * 3 + k]);
        }
      }
      // 8. Crespo-Hernandez + overlap gating [A.3-8]
      const bool reach = (x_t > x_i) && (x_t <= x_i + 15.0 * D);
      bool gate[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) gate[j] = reach && (fabs(y_i - (y_t + c.off[j])) < 2.0 * D);
      if (gate[0] || gate[1] || gate[2]) {
        const double dxp = (dx <= 0.1) ? dx + 1.0 : dx;  // dx > -0.1 holds for every downstream turbine
        double ti = ch_pref * exp(c.ch_down * log(dxp / D));
        if (isnan(ti) || (isinf(ti) && ti > 0)) ti = 0.0;
        const double ti_added = ((double)cnt / 9.0) * ti;
        const double cand = sqrt(ti_added * ti_added + c.amb * c.amb);
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (gate[j] && cand > TI[j]) TI[j] = cand;
      }
    }  // sources

    // ---- outputs [A.4] in the caller's turbine order; reward partial sums ----
    double pw = 0.0, lsum = 0.0;
    if (live) {
      double m3 = 0.0, mu = 0.0, mv = 0.0, mw = 0.0, dir = 0.0, U[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        U[q] = Uinit[q % 3] - sqrt(wake2[q]);
        m3 += U[q] * U[q] * U[q];
        mu += U[q]; mv += V[q]; mw += W[q];
        dir += wd - atan2(V[q], U[q]) / kDeg;
      }
      mu /= 9.0; mv /= 9.0; mw /= 9.0;
      double su = 0.0, sv = 0.0, sw = 0.0;
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        su += (U[q] - mu) * (U[q] - mu);
        sv += (V[q] - mv) * (V[q] - mv);
        sw += (W[q] - mw) * (W[q] - mw);
      }
      const double wsp = cbrt(m3 / 9.0);
      const double veff = c.dens_cbrt * (wsp * pow(cosd(yaw_t), c.pP3));
      pw = c.rho_ref * interp_fill(veff, c.n_table, tws, tpw, 0.0, 0.0);
      const double l0 = (TI[0] + TI[1] + TI[2]) / 3.0, l1 = sqrt(su / 9.0), l2 = sqrt(sv / 9.0), l3 = sqrt(sw / 9.0);
      lsum = fabs(l0) + fabs(l1) + fabs(l2) + fabs(l3);
      const size_t oo = (size_t)b * N + o;
      if (a.o_power) a.o_power[oo] = (float)pw;
      if (a.o_ws) a.o_ws[oo] = (float)wsp;
      if (a.o_wd) a.o_wd[oo] = (float)(dir / 9.0);
      if (a.o_load) reinterpret_cast<float4*>(a.o_load)[oo] = make_float4((float)l0, (float)l1, (float)l2, (float)l3);
    }
    if (a.reward) {  // reference simple_env.py:78-84 on the float64 values
#pragma unroll
      for (int w = 32; w >= 1; w >>= 1) {
        pw += __shfl_xor(pw, w);
        lsum += __shfl_xor(lsum, w);
      }
      if ((t & 63) == 0) { red[0][t >> 6] = pw; red[1][t >> 6] = lsum; }
      __syncthreads();
      if (t == 0) {
        double ps = 0.0, ls = 0.0;
        for (int w = 0; w < THREADS / 64; ++w) { ps += red[0][w]; ls += red[1][w]; }
        const double wr = a.ws_prev ? a.ws_prev[b] : ws;
        a.reward[b] = (float)(ps / N / 1.0e6 * 1.0e3 / (wr * wr * wr) - (double)a.load_coef * ls / (4.0 * N));
      }
    }
    if (t == 0) a.flags[b] = 0;
  }
}

extern "C" hipError_t wfk_launch_resolve(const WfResolveConsts* c, const WfResolveArgs* a, int B, int all, int* raw_flags,
                                         hipStream_t s) {
  hipError_t e = hipMemsetAsync(a->count, 0, sizeof(int), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(wf_compact_flagged_kernel, dim3((B + 255) / 256), dim3(256), 0, s, a->flags, B, all, a->list, a->count, raw_flags);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  // persistent blocks over the compacted list: enough to fill the chip several times over, never more than farms
  int grid = B < 2048 ? B : 2048;
  const int N = c->N;
  if (N <= 64) hipLaunchKernelGGL(wf_resolve_kernel<64>, dim3(grid), dim3(64), 0, s, *c, *a);
  else if (N <= 128) hipLaunchKernelGGL(wf_resolve_kernel<128>, dim3(grid), dim3(128), 0, s, *c, *a);
  else if (N <= 192) hipLaunchKernelGGL(wf_resolve_kernel<192>, dim3(grid), dim3(192), 0, s, *c, *a);
  else hipLaunchKernelGGL(wf_resolve_kernel<256>, dim3(grid), dim3(256), 0, s, *c, *a);
  return hipGetLastError();
}
