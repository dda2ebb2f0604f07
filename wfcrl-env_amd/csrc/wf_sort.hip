// wf_sort.hip — farm order of a launch with a wind PER FARM (reference wfcrl/mdp.py:237-258: every env draws its own
// direction at reset): launch slots sorted by wind direction, so that the 64 / G farms a wave of wf_step_ll_kernel solves
// lie within a fraction of a degree of each other.  Their (source, target) geometry then nearly agrees, and the kernel's
// wave-uniform far-source / far-pair skip (wf_kernels_ll.hip) takes as it does on the pair-table path.  The result of a
// farm does not depend on its slot: the order is purely a matter of speed.
// Device-side (hipcub radix sort on float keys, stable): nothing is read back, wf_set_wind stays asynchronous.
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

__global__ void wf_dir_keys_kernel(int B, const double* __restrict__ wd, float* __restrict__ key, int* __restrict__ val) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double d = fmod(wd[b], 360.0);
  if (d < 0.0) d += 360.0;
  key[b] = (float)d;
  val[b] = b;
}

__global__ void wf_fill_int_kernel(int n, int* __restrict__ a, int v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}

// bytes of temporary storage the sort of B keys needs
extern "C" hipError_t wfk_sort_tmp_bytes(int B, size_t* bytes) {
  *bytes = 0;
  return hipcub::DeviceRadixSort::SortPairs(nullptr, *bytes, (const float*)nullptr, (float*)nullptr, (const int*)nullptr,
                                            (int*)nullptr, B, 0, 32, (hipStream_t)0);
}

// perm[0 .. B) = farm indices in ascending wind direction (mod 360), perm[B .. n_slots) = -1 (padding of the last block);
// keys: 2 B floats, vals: B ints, tmp: wfk_sort_tmp_bytes(B)
extern "C" hipError_t wfk_sort_by_direction(int B, int n_slots, const double* wd, float* keys, int* vals, void* tmp, size_t tmp_bytes,
                                            int* perm, hipStream_t s) {
  hipLaunchKernelGGL(wf_dir_keys_kernel, dim3((B + 255) / 256), dim3(256), 0, s, B, wd, keys, vals);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, (const float*)keys, keys + B, (const int*)vals, perm, B, 0, 32, s);
  if (e != hipSuccess) return e;
  if (n_slots > B) {
    hipLaunchKernelGGL(wf_fill_int_kernel, dim3((n_slots - B + 255) / 256), dim3(256), 0, s, n_slots - B, perm + B, -1);
    e = hipGetLastError();
  }
  return e;
}
