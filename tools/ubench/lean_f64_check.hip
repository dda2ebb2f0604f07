// Device check of csrc/wf_f64_math.h against the device library: max relative / absolute deviation over sampled arguments.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off -I wfcrl-env_amd/csrc -o lean_f64_check tools/ubench/lean_f64_check.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include "wf_f64_math.h"
__device__ double atomicMaxD(double* addr, double v) {
  unsigned long long* a = (unsigned long long*)addr, old = *a, assumed;
  do { assumed = old; if (__longlong_as_double(assumed) >= v) break; old = atomicCAS(a, assumed, __double_as_longlong(v)); } while (assumed != old);
  return __longlong_as_double(old);
}
__global__ void check(double* worst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double u = (i + 0.5) / n, v = ((i * 2654435761u) % 1000003u) / 1000003.0;
  auto rel = [](double a, double b) { return fabs(a - b) / fmax(fabs(b), 1e-300); };
  const double x = exp(log(10.0) * (-6.0 + 12.0 * u));
  atomicMaxD(&worst[0], rel(sqrt_pos(x), sqrt(x)));
  atomicMaxD(&worst[1], rel(rcp64(x), 1.0 / x));
  atomicMaxD(&worst[2], rel(exp_lean(-700.0 * u), exp(-700.0 * u)));
  atomicMaxD(&worst[2], rel(exp_lean(20.0 * (v - 0.5)), exp(20.0 * (v - 0.5))));
  atomicMaxD(&worst[3], fabs(log_lean(x) - log(x)) / fmax(fabs(log(x)), 1.0));
  atomicMaxD(&worst[4], rel(pow_any(x, -0.32), pow(x, -0.32)));
  double sn, cs; const double a = 1.6 * (u - 0.5);
  sincos_small(a, sn, cs); atomicMaxD(&worst[5], fmax(fabs(sn - sin(a)), fabs(cs - cos(a))));
  atomicMaxD(&worst[6], rel(tan_small(u - 0.5), tan(u - 0.5)));
  atomicMaxD(&worst[7], fabs(asin_small(0.6 * (u - 0.5)) - asin(0.6 * (u - 0.5))));
  const double big = 200.0 * (u - 0.5);
  sincos_any(big, sn, cs); atomicMaxD(&worst[8], fmax(fabs(sn - sin(big)), fabs(cs - cos(big))));
  atomicMaxD(&worst[9], fabs(atan2_any(4.0 * (u - 0.5), 4.0 * (v - 0.5)) - atan2(4.0 * (u - 0.5), 4.0 * (v - 0.5))));
  const double cb = (v < 0.5 ? -1.0 : 1.0) * x;
  atomicMaxD(&worst[10], rel(cbrt_any(cb), cbrt(cb)));
  atomicMaxD(&worst[11], fabs(asin_any(2.0 * (u - 0.5)) - asin(2.0 * (u - 0.5))));
  atomicMaxD(&worst[12], rel(cbrt_pos(x), cbrt(x)));
  atomicMaxD(&worst[13], fabs(atan_small(0.2 * (u - 0.5)) - atan(0.2 * (u - 0.5))));
  if (i == 0) {
    worst[16] = exp_lean(-__builtin_huge_val()); worst[17] = exp_lean(__builtin_huge_val()); worst[18] = exp_lean(__builtin_nan(""));
    worst[19] = log_any(0.0); worst[20] = log_any(-1.0); worst[21] = pow_any(0.0, -0.32); worst[22] = cbrt_any(0.0); worst[23] = cbrt_any(-8.0);
    worst[24] = exp_lean(-1.0); worst[25] = sqrt_pos(2.0); worst[26] = log_lean(2.0); worst[27] = __builtin_amdgcn_rsq(4.0);
  }
}
int main() {
  double* d; hipMalloc(&d, 32 * sizeof(double)); hipMemset(d, 0, 32 * sizeof(double));
  const int n = 1 << 22;
  hipLaunchKernelGGL(check, dim3(n / 256), dim3(256), 0, 0, d, n);
  double h[32]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[14] = {"sqrt_pos", "rcp64", "exp_lean", "log_lean", "pow_any", "sincos_small", "tan_small", "asin_small", "sincos_any", "atan2_any", "cbrt_any", "asin_any", "cbrt_pos", "atan_small"};
  for (int k = 0; k < 14; ++k) printf("%-14s max deviation from the device library %.3e\n", nm[k], h[k]);
  printf("exp(-inf)=%g exp(+inf)=%g exp(nan)=%g log(0)=%g log(-1)=%g pow(0,-0.32)=%g cbrt(0)=%g cbrt(-8)=%g | exp(-1)=%.17g sqrt(2)=%.17g log(2)=%.17g rsq(4)=%.17g\n",
         h[16], h[17], h[18], h[19], h[20], h[21], h[22], h[23], h[24], h[25], h[26], h[27]);
  return 0;
}
