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
  int j = 0;
  while (j < n - 2 && xq >= xs[j + 1]) ++j;
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

template <int THREADS>
__global__ __launch_bounds__(THREADS) void wf_resolve_kernel(const WfResolveConsts c, const WfResolveArgs a) {
  __shared__ double tws[WF_TABLE_PAD], tct[WF_TABLE_PAD], tpw[WF_TABLE_PAD];
  __shared__ double xsL[THREADS], ysL[THREADS], yawL[THREADS];
  __shared__ double bc[5], bc2[2];
  __shared__ double red[2][THREADS / 64];
  const int t = threadIdx.x;
  const int N = c.N;
  for (int k = t; k < c.n_table; k += THREADS) {
    tws[k] = a.tab64[k];
    tct[k] = a.tab64[WF_TABLE_PAD + k];
    tpw[k] = a.tab64[2 * WF_TABLE_PAD + k];
  }
  const int n_list = *a.count;
  const bool live = t < N;
  const double D = c.D, eps2 = c.eps2;

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
    __syncthreads();  // the previous farm's last readers of xsL / bc are done
    if (live) {
      o = a.gidx[gofs + t];
      x_t = a.gx[gofs + t];
      y_t = a.gy[gofs + t];
      yaw_t = (double)yaw_b[o];
      xsL[t] = x_t; ysL[t] = y_t; yawL[t] = yaw_t;
    }
    // inflow [A.2]
    double Uinit[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) Uinit[k] = ws * c.shearf[k];
    const double Uinf = ws * c.uinf1;
    double wake[9], V[9], W[9], TI[3];
#pragma unroll
    for (int q = 0; q < 9; ++q) { wake[q] = 0.0; V[q] = 0.0; W[q] = 0.0; }
#pragma unroll
    for (int j = 0; j < 3; ++j) TI[j] = c.amb;

    for (int i = 0; i < N; ++i) {
      if (t == i) {  // the source's state [A.3-1, A.3-2]
        double m3 = 0.0, vs = 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          const double u = Uinit[q % 3] - wake[q];
          m3 += u * u * u;
          vs += V[q];
        }
        bc[0] = m3 / 9.0; bc[1] = vs / 9.0; bc[2] = TI[0]; bc[3] = TI[1]; bc[4] = TI[2];
      }
      __syncthreads();
      const double x_i = xsL[i], y_i = ysL[i], g = yawL[i];
      const double ubar = cbrt(bc[0]), Vmean = bc[1];
      const double TIs[3] = {bc[2], bc[3], bc[4]};
      const double cg = cosd(g);
      double ct_tab = interp_fill(ubar, c.n_table, tws, tct, 0.0001, 0.9999);
      ct_tab = fmin(fmax(ct_tab, 0.0001), 0.9999);
      const double ct = ct_tab * cg;
      const double ai = 0.5 / cg * (1.0 - sqrt(1.0 - ct * cg));
      const double G_wr = 0.25 * kTwoPi * D * (ai - ai * ai) * ubar / c.TSR;
      const double gam_top = (kTwoPi / 16.0) * D * c.vel_top * Uinf * ct;
      const double gam_bot = (kTwoPi / 16.0) * D * c.vel_bot * Uinf * ct;
      const double dx = x_t - x_i;
      const bool down = live && dx >= 0.0;  // ties (dx = 0) count as downstream for the transverse velocities [A.3-4]

      // 4. transverse velocities (commanded yaw) on this thread's turbine
      if (c.sw_tv && __any(down)) {
        if (down) {
          const double sc = sind(g) * cg;
          const double Gs[3] = {sc * gam_top, -sc * gam_bot, G_wr};
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const double yL = (y_t + c.off[j] - y_i) + c.num_eps;
            const double yL2 = yL * yL;
            const double Ey = exp(-yL2 / eps2);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const double decay = eps2 / (4.0 * (c.nu1[k] * ws) * dx / Uinf + eps2);
              double v = 0.0, w = 0.0;
#pragma unroll
              for (int vv = 0; vv < 3; ++vv) {
                {
                  const double zc = c.zr[k][vv];
                  const double r = yL2 + zc * zc;
                  const double kk = Gs[vv] / (kTwoPi * r) * (1.0 - Ey * c.ezr[k][vv]) * decay;
                  v += kk * zc; w -= kk * yL;
                }
                {
                  const double zc = c.zm[k][vv];  // ground mirror
                  const double r = yL2 + zc * zc;
                  const double kk = Gs[vv] / (kTwoPi * r) * (1.0 - Ey * c.ezm[k][vv]) * decay;
                  v -= kk * zc; w += kk * yL;
                }
              }
              if (w < 0.0) w = 0.0;  // quirk (5) [A.6]
              V[j * 3 + k] += v;
              W[j * 3 + k] += w;
            }
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
      // 2. secondary steering [A.3-2]: the three means on the source's own grid are geometry constants
      const double v_top = gam_top * c.k_top, v_bot = -gam_bot * c.k_bot, v_core = G_wr * c.k_core;
      double val = 2.0 * (Vmean - v_core) / (v_top + v_bot);
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
      if (!__any(down)) continue;
      if (!down) continue;
      // 3 + 6. deflection (TI before mixing, effective yaw) and deficit (TI after mixing, commanded yaw) [A.3-3, A.3-6]
      const double gd = -g_eff, cgd = cosd(gd);
      const double s_cc = sqrt(1.0 - ct * cgd), s_c = sqrt(1.0 - ct);
      const double th0 = c.dm * (0.3 * (gd * kDeg) / cgd) * (1.0 - s_cc);
      const double tan_th0 = tan(th0);
      const double gv = -g, cgv = cosd(gv);
      const double C0 = 1.0 - s_c;
      const double M0 = C0 * (2.0 - C0);
      const double E0 = C0 * C0 - c.e0c1 * C0 + c.e0c2;
      const double sM = sqrt(M0);
      const double sz0d = D * 0.5 * sqrt((ct * cgd / (2.0 * (1.0 - s_cc))) / (1.0 + s_c));
      const double sy0d = sz0d * cgd * c.cos_veer;
      const double sz0v = D * 0.5 * sqrt((ct / (2.0 * (1.0 - s_c))) / (1.0 + s_c));
      const double sy0v = sz0v * cgv * c.cos_veer;
      const double snw = c.near_c * sqrt(ct / 2.0);
      const double lin = c.ad + c.bd * dx;
      double defU[9];
      int cnt = 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double delta;
        {
          const double TIq = TIs[j];
          const double x0 = D * cgd * (1.0 + s_cc) / (c.sqrt2 * (4.0 * c.defl_alpha * TIq + 2.0 * c.defl_beta * (1.0 - s_c))) + x_i;
          const double ky = c.defl_ka * TIq + c.defl_kb;
          const double d0 = tan_th0 * (x0 - x_i);
          double d_near = (dx / (x0 - x_i)) * d0 + lin;
          if (!(x_t <= x0)) d_near = 0.0;  // [x >= x_i] holds here
          double d_far = 0.0;
          if (x_t > x0) {
            const double sy = ky * (x_t - x0) + sy0d, sz = ky * (x_t - x0) + sz0d;
            const double s = sqrt(sy * sz / (sy0d * sz0d));
            const double ln_arg = ((1.6 + sM) * (1.6 * s - sM)) / ((1.6 - sM) * (1.6 * s + sM));
            d_far = d0 + th0 * E0 / 5.2 * sqrt(sy0d * sz0d / (ky * ky * M0)) * log(ln_arg) + lin;
          }
          delta = d_near + d_far;
        }
        double amp = 0.0, ca = 0.0, cb = 0.0, cc = 0.0;
        bool on = false;
        {
          const double TIq = TIs[j] + dTI;
          const double x0 = D * cgv * (1.0 + s_c) / (c.sqrt2 * (4.0 * c.alpha * TIq + 2.0 * c.beta * (1.0 - s_c))) + x_i;
          double sy = 0.0, sz = 0.0;
          if (x_t > x_i + 0.1 && x_t < x0) {  // the masks as FLORIS takes them on the coordinates
            const double up = dx / (x0 - x_i), dn = (x0 - x_t) / (x0 - x_i);
            sy = dn * snw + up * sy0v;
            sz = dn * snw + up * sz0v;
            on = true;
          } else if (x_t >= x0) {
            const double ky = c.ka * TIq + c.kb;
            sy = ky * (x_t - x0) + sy0v;
            sz = ky * (x_t - x0) + sz0v;
            on = true;
          }
          if (on) {
            double dd = 1.0 - ct * cgv / (8.0 * sy * sz / (D * D));
            dd = fmin(fmax(dd, 0.0), 1.0);
            amp = 1.0 - sqrt(dd);
            // FLORIS rCalt with veer [gauss.py]: a, b, c of the rotated Gaussian (veer = 0: 1/(2 sy^2), 0, 1/(2 sz^2))
            const double isy2 = 1.0 / (2.0 * sy * sy), isz2 = 1.0 / (2.0 * sz * sz);
            ca = c.cos2_veer * isy2 + c.sin2_veer * isz2;
            cb = 0.5 * c.sin_2veer * (isz2 - isy2);
            cc = c.sin2_veer * isy2 + c.cos2_veer * isz2;
          }
        }
        const double yy = (y_t + c.off[j]) - y_i - delta;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double zz = c.off[k];
          const double r = ca * yy * yy - 2.0 * cb * yy * zz + cc * zz * zz;
          const double deficit = on ? amp * exp(-r) : 0.0;
          defU[j * 3 + k] = deficit * Uinit[k];
          if (defU[j * 3 + k] > c.overlap_thr) ++cnt;
        }
      }
      // 7. SOSFS [A.3-7]
#pragma unroll
      for (int q = 0; q < 9; ++q) wake[q] = hypot(wake[q], defU[q]);
      // 8. Crespo-Hernandez + overlap gating [A.3-8]
      {
        const double upm = (dx <= 0.1) ? 1.0 : 0.0;  // dx > -0.1 holds for every downstream turbine
        const double dxp = dx + upm;
        const double ch_pref = c.ch_constant * pow(ai, c.ch_ai) * c.ch_amb_pow;
        double ti = ch_pref * pow(dxp / D, c.ch_down);
        if (isnan(ti) || (isinf(ti) && ti > 0)) ti = 0.0;
        const double overlap = (double)cnt / 9.0;
        const bool reach = (x_t > x_i) && (x_t <= x_i + 15.0 * D);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const bool gate = reach && (fabs(y_i - (y_t + c.off[j])) < 2.0 * D);
          const double ti_added = gate ? overlap * ti : 0.0;
          const double cand = sqrt(ti_added * ti_added + c.amb * c.amb);
          if (cand > TI[j]) TI[j] = cand;
        }
      }
    }  // sources

    // ---- outputs [A.4] in the caller's turbine order; reward partial sums ----
    double pw = 0.0, lsum = 0.0;
    if (live) {
      double m3 = 0.0, mu = 0.0, mv = 0.0, mw = 0.0, dir = 0.0, U[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        U[q] = Uinit[q % 3] - wake[q];
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
