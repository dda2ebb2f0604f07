// Does alternating kernels WITH and WITHOUT a private segment (scratch) on one stream cost more than the launches themselves?
// (round 5: the float64 re-solve kernels carried 660-820 bytes of scratch per lane; the step kernels carry none)
//   hipcc --offload-arch=gfx950 -O3 -o scratch_switch scratch_switch.hip && ./scratch_switch
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_plain(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
__device__ __noinline__ float callee(volatile float* a, int i) { return a[i & 63] * 2.0f; }
__global__ void k_scratch(float* p, int n) {  // a stack array indexed dynamically: a real private segment
  volatile float a[64];
  for (int i = 0; i < 64; ++i) a[i] = p[0] + i;
  float s = 0.0f;
  for (int i = 0; i < n; ++i) s += callee(a, i + threadIdx.x);
  if (threadIdx.x == 0 && blockIdx.x == 0) p[1] = s;
}
int main() {
  float* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, int mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, s);
      for (int i = 0; i < 200; ++i) {
        if (mode == 0) { hipLaunchKernelGGL(k_plain, dim3(512), dim3(256), 0, s, d); hipLaunchKernelGGL(k_plain, dim3(512), dim3(256), 0, s, d); }
        if (mode == 1) { hipLaunchKernelGGL(k_plain, dim3(512), dim3(256), 0, s, d); hipLaunchKernelGGL(k_scratch, dim3(512), dim3(256), 0, s, d, 1); }
        if (mode == 2) { hipLaunchKernelGGL(k_scratch, dim3(512), dim3(256), 0, s, d, 1); hipLaunchKernelGGL(k_scratch, dim3(512), dim3(256), 0, s, d, 1); }
      }
      hipEventRecord(e1, s); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%-40s %.2f us per pair of launches\n", name, ms * 1000.0f / 200);
    }
  };
  run("plain + plain", 0);
  run("plain + scratch (alternating)", 1);
  run("scratch + scratch", 2);
  return 0;
}
