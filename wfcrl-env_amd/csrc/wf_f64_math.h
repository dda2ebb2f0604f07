// wf_f64_math.h — lean float64 arithmetic of the float64 re-solve (wf_resolve.hip): square root, reciprocal, exp, log, pow and
// the trigonometric functions of the per-source chain without the device library's range scaling and special cases, plus
// inline general-range versions for the rare arguments outside the model's normal ranges.  Checked on the device against the
// library by tools/ubench/lean_f64_check.hip (profiles/r05_lean_f64_check.txt).
#pragma once
#include <hip/hip_runtime.h>

namespace {

constexpr double kDeg = 3.14159265358979323846 / 180.0;
constexpr double kTwoPi = 2.0 * 3.14159265358979323846;

// ---- lean float64 arithmetic (round 5) --------------------------------------------------------------------------------
// The re-solve is LATENCY-bound: one farm's N-stage dependent chain, a wave (or four) per farm.  The device library's sqrt /
// exp / log / division are general-range IEEE routines — 17-25, ~50, ~60 and 11+ instructions with range scaling, special-case
// selects and (exp, log) table lookups; round 4's kernels issued ~3 000 instructions per source stage, three quarters of them
// inside those routines.  Every operand of this model is a positive, finite, NORMAL double of moderate size, so the routines
// below drop the scaling and the special cases and keep the accuracy (<= ~2 ulp): the stage is ~800 instructions now.
// 1 / x from the hardware estimate and two Newton steps (6 instructions; the compiler's IEEE division sequence is 11)
__device__ __forceinline__ double rcp64(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}
// sqrt(x), 1e-280 < x < 1e280: Goldschmidt from the hardware reciprocal-root estimate with a final correction — the sequence
// LLVM emits for sqrt(double) without its range scaling (10 instructions against 17 + the library's wrapper)
__device__ __forceinline__ double sqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  double d = fma(-g, g, x);
  g = fma(d, h, g);
  d = fma(-g, g, x);
  return fma(d, h, g);
}
// ... for an operand that may be exactly 0 (a sum of squared deficits before any wake arrives, a variance)
__device__ __forceinline__ double sqrt_nn(double x) { return x > 1.0e-280 ? sqrt_pos(x) : 0.0; }
// exp(x), x <= 0 (down to total underflow) or moderately positive: x = k ln 2 + r, |r| <= 0.347, 2^k by v_ldexp_f64; the
// series to r^13 (remainder 0.347^14 / 14! = 4e-18) as two interleaved chains in r^2 (the wave is alone on its SIMD: a
// dependent fma costs its full latency)
__device__ __forceinline__ double exp_lean(double x_in) {
  const double x = fmin(fmax(x_in, -1000.0), 1000.0);  // (+-inf: total underflow / overflow through v_ldexp_f64 below)
  const double k = rint(x * 1.4426950408889634);
  double r = fma(k, -6.93147180369123816490e-01, x);
  r = fma(k, -1.90821492927058770002e-10, r);
  const double r2 = r * r;
  double pe = 1.0 / 479001600.0, po = 1.0 / 6227020800.0;  // 1/12!, 1/13!
  pe = fma(pe, r2, 1.0 / 3628800.0);  po = fma(po, r2, 1.0 / 39916800.0);
  pe = fma(pe, r2, 1.0 / 40320.0);    po = fma(po, r2, 1.0 / 362880.0);
  pe = fma(pe, r2, 1.0 / 720.0);      po = fma(po, r2, 1.0 / 5040.0);
  pe = fma(pe, r2, 1.0 / 24.0);       po = fma(po, r2, 1.0 / 120.0);
  pe = fma(pe, r2, 0.5);              po = fma(po, r2, 1.0 / 6.0);
  pe = fma(pe, r2, 1.0);              po = fma(po, r2, 1.0);
  const double res = ldexp(fma(po, r, pe), (int)k);
  return x_in != x_in ? x_in : res;  // (fmax / fmin drop a NaN)
}
// log(x), x > 0 normal: x = 2^e m, m in [sqrt(1/2), sqrt 2); log m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.1716: the
// series to s^21 (remainder 0.0295^11 / 23 = 6e-19)
__device__ __forceinline__ double log_lean(double x) {
  int e = __builtin_amdgcn_frexp_exp(x);
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  const bool lo = m < 0.70710678118654752;
  m = lo ? 2.0 * m : m;
  e = lo ? e - 1 : e;
  const double s = (m - 1.0) * rcp64(m + 1.0);
  const double s2 = s * s, s4 = s2 * s2;
  double pa = 1.0 / 19.0, pb = 1.0 / 21.0;  // two interleaved chains in s^4
  pa = fma(pa, s4, 1.0 / 15.0);  pb = fma(pb, s4, 1.0 / 17.0);
  pa = fma(pa, s4, 1.0 / 11.0);  pb = fma(pb, s4, 1.0 / 13.0);
  pa = fma(pa, s4, 1.0 / 7.0);   pb = fma(pb, s4, 1.0 / 9.0);
  pa = fma(pa, s4, 1.0 / 3.0);   pb = fma(pb, s4, 1.0 / 5.0);
  const double p = fma(fma(pb, s2, pa), s2, 1.0);  // 1 + s^2/3 + s^4/5 + ...
  const double ed = (double)e;
  return fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, 2.0 * s * p));
}
__device__ __forceinline__ double pow_lean(double x, double y) { return exp_lean(y * log_lean(x)); }
// log(x) over the whole domain: the routine above serves every positive finite double (v_frexp_* normalise denormals)
__device__ __forceinline__ double log_any(double x) {
  double r = log_lean(x);
  r = (x == 0.0) ? -__builtin_huge_val() : r;
  r = (x < 0.0 || x != x) ? __builtin_nan("") : r;
  return (x == __builtin_huge_val()) ? x : r;
}
__device__ __forceinline__ double pow_any(double x, double y) { return exp_lean(y * log_any(x)); }

// The transcendental functions of the per-source chain on the argument ranges this model produces, as short branch-free
// polynomials; general-range versions built on them follow (chosen wave-uniformly where an argument leaves the range).
// sin(x) and cos(x), |x| <= 0.8 (every admissible yaw, 45 deg = 0.785): Taylor series to x^21 / x^20 (next terms
// 0.8^23 / 23! = 2e-25, 0.8^22 / 22! = 7e-24)
__device__ __forceinline__ void sincos_small(double x, double& sn_out, double& cs_out) {
  const double x2 = x * x;
  double sn = 1.0 / 51090942171709440000.0, cs = 1.0 / 2432902008176640000.0;  // 1/21!, 1/20!
  const double fs[10] = {1.0 / 121645100408832000.0, 1.0 / 355687428096000.0, 1.0 / 1307674368000.0, 1.0 / 6227020800.0,
                         1.0 / 39916800.0, 1.0 / 362880.0, 1.0 / 5040.0, 1.0 / 120.0, 1.0 / 6.0, 1.0};
  const double fc[10] = {1.0 / 6402373705728000.0, 1.0 / 20922789888000.0, 1.0 / 87178291200.0, 1.0 / 479001600.0,
                         1.0 / 3628800.0, 1.0 / 40320.0, 1.0 / 720.0, 1.0 / 24.0, 1.0 / 2.0, 1.0};
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    sn = fma(-x2, sn, fs[k]);
    cs = fma(-x2, cs, fc[k]);
  }
  sn_out = x * sn;
  cs_out = cs;
}
// tan(x), |x| <= 0.5: the same series, one division
__device__ __forceinline__ double tan_small(double x) {
  double sn, cs;
  sincos_small(x, sn, cs);
  return sn * rcp64(cs);
}
// asin(x), |x| <= 0.3: odd series, coefficients (2k)! / (4^k (k!)^2 (2k + 1)), 17 terms (0.3^36 / 37 = 4e-21)
__device__ __forceinline__ double asin_small(double x) {
  const double x2 = x * x;
  double cf[18];
  cf[0] = 1.0;
  double b = 1.0;
#pragma unroll
  for (int k = 1; k < 18; ++k) {
    b *= (2.0 * k - 1.0) / (2.0 * k);
    cf[k] = b / (2.0 * k + 1.0);
  }
  const double x4 = x2 * x2;
  double pe = cf[16], po = cf[17];  // even / odd powers of x^2: two interleaved chains
#pragma unroll
  for (int k = 14; k >= 0; k -= 2) {
    pe = fma(pe, x4, cf[k]);
    po = fma(po, x4, cf[k + 1]);
  }
  return x * fma(po, x2, pe);
}
// atan(r), |r| <= 0.1: odd series, 10 terms (0.1^20 / 21 = 5e-22)
__device__ __forceinline__ double atan_small(double r) {
  const double r2 = r * r;
  double p = -1.0 / 19.0;
#pragma unroll
  for (int k = 8; k >= 0; --k) p = fma(p, r2, ((k & 1) ? -1.0 : 1.0) / (2.0 * k + 1.0));
  return r * p;
}
// cbrt(x), x > 0: float estimate, two Newton steps in float64
__device__ __forceinline__ double cbrt_pos(double x) {
  double y = (double)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)x) * (1.0f / 3.0f));
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const double y2 = y * y;
    y = y - (y2 * y - x) * rcp64(3.0 * y2);
  }
  return y;
}

// ---- general-range versions (inline; rarely taken) ----
// cbrt of any finite double: x = m 2^(3 q + r) -> cbrt(m 2^r) 2^q, sign carried over (np.cbrt semantics)
__device__ __forceinline__ double cbrt_any(double x) {
  const double ax = fabs(x);
  const int e = __builtin_amdgcn_frexp_exp(ax);
  const double m = __builtin_amdgcn_frexp_mant(ax);  // [0.5, 1)
  const int q = (e >= 0 ? e : e - 2) / 3, r = e - 3 * q;  // floor division: r in {0, 1, 2}
  const double y = ldexp(cbrt_pos(ldexp(m, r)), q);
  return (ax > 0.0 && ax < __builtin_huge_val()) ? copysign(y, x) : x;  // +-0, +-inf and NaN pass through
}
// sin / cos of any moderate angle: x = k pi/2 + r (three-part pi/2), |r| <= pi/4, then the series above
__device__ __forceinline__ void sincos_any(double x, double& s_out, double& c_out) {
  const double k = rint(x * 0.63661977236758134);
  double r = fma(k, -1.57079632673412561417e+00, x);
  r = fma(k, -6.07710050650619224932e-11, r);
  r = fma(k, -2.02226624879595063154e-21, r);
  double sr, cr;
  sincos_small(r, sr, cr);
  const int n = (int)k & 3;
  const double s1 = (n & 1) ? cr : sr, c1 = (n & 1) ? sr : cr;
  s_out = (n & 2) ? -s1 : s1;
  c_out = ((n + 1) & 2) ? -c1 : c1;
}
__device__ __forceinline__ double tan_any(double x) {
  double sn, cs;
  sincos_any(x, sn, cs);
  return sn * rcp64(cs);
}
// atan(r), 0 <= r <= 1: r -> (r - c) / (1 + r c) about the nearest c = j / 8, then the series for |arg| <= 0.0625
__device__ __forceinline__ double atan_01(double r) {
  const double j = rint(r * 8.0), cpt = j * 0.125;
  const double arg = (r - cpt) * rcp64(fma(r, cpt, 1.0));
  const double at[9] = {0.0, 0.12435499454676144, 0.24497866312686414, 0.35877067027057225, 0.46364760900080615,
                        0.5585993153435624, 0.6435011087932844, 0.7188299996216245, 0.7853981633974483};
  double base = 0.0;
#pragma unroll
  for (int k = 1; k < 9; ++k) base = (j == (double)k) ? at[k] : base;
  return base + atan_small(arg);
}
__device__ __forceinline__ double atan2_any(double y, double x) {
  const double ay = fabs(y), ax = fabs(x);
  const bool swap = ay > ax;
  const double num = swap ? ax : ay, den = swap ? ay : ax;
  double a = den > 0.0 ? atan_01(num * rcp64(den)) : 0.0;
  a = swap ? 1.5707963267948966 - a : a;
  a = x < 0.0 ? 3.141592653589793 - a : a;
  return copysign(a, y);
}
// asin(x), |x| <= 1
__device__ __forceinline__ double asin_any(double x) { return atan2_any(x, sqrt_nn(fmax(1.0 - x * x, 0.0))); }

}  // namespace
