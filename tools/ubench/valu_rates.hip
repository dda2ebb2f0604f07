// Microbenchmark (GPU box): issue rate of fp32 VALU instruction classes on gfx950 as a function of waves
// per SIMD.  Establishes the VALU roof that DESIGN.md prices the step kernel against.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void k(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0000001f, c = 1e-7f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {  // 8 independent v_fma_f32 chains, 64 instr / iter
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if (KIND == 1) {  // 4 independent v_pk_fma_f32 chains x2 = 64 instr / iter (128 fma)
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                        "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
    } else if (KIND == 2) {  // v_exp_f32
      REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                        "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 3) {  // v_rcp_f32
      REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                        "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 4) {  // 6 fma + 1 rcp + 1 exp per 8
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_rcp_f32 %3, %3\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_exp_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if (KIND == 5) {  // v_mul_f32 + v_add_f32 alternating
      REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                        "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if (KIND == 6) {  // single dependent chain v_fma_f32
      REP64(asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(a0) : "v"(b), "v"(c));)
    } else if (KIND == 7) {  // v_pk_mul_f32 / v_pk_add_f32
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
                        "v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
    } else if (KIND == 8) {  // v_sqrt_f32
      REP8(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                        "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 9) {  // v_cndmask + v_cmp pairs
      REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                        "v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int KIND>
void run(const char* name, float* d, int flops_per_instr) {
  const int iters = 2000;
  for (int wps : {1, 2, 4, 8}) {
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)iters * 64 * wps;            // wave-instructions issued per SIMD
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-26s waves/SIMD=%d  %.3f ms  %.2f cyc/wave-instr/SIMD (at 2.4GHz)  %.1f T lane-ops/s\n", name, wps, ms,
           cyc / instr_per_simd, instr_per_simd * 1024 * 64 * flops_per_instr / (ms * 1e-3) / 1e12);
  }
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_fma_f32 x8 indep", d, 1);
  run<1>("v_pk_fma_f32 x4 indep", d, 2);
  run<5>("v_mul/v_add", d, 1);
  run<7>("v_pk_mul/v_pk_add", d, 2);
  run<2>("v_exp_f32", d, 1);
  run<3>("v_rcp_f32", d, 1);
  run<8>("v_sqrt_f32", d, 1);
  run<4>("6 fma + rcp + exp", d, 1);
  run<9>("v_cmp + v_cndmask", d, 1);
  run<6>("v_fma_f32 dependent", d, 1);
  return 0;
}
