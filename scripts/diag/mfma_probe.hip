// Probe: issue rate of v_mfma_f32_32x32x2_f32 by accumulator pattern (one wave per SIMD, 256-thread workgroups, one per CU).
//   mode 0: 16 independent accumulators round-robin (the weight-gradient kernel's pattern)
//   mode 1: 16 accumulators, four consecutive MFMAs per accumulator (the chain kernels' pattern)
//   mode 2: 4 accumulators round-robin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float *out, unsigned long long *cyc, int iters) {
  floatx16 acc[16];
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = threadIdx.x * 0.002f - i; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) acc[4 * ta + tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], b[tb], acc[4 * ta + tb], 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t + 4 * (it & 3)] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[t], acc[t + 4 * (it & 3)], 0, 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k & 3], b[(k >> 2) & 3], acc[k & 3], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float *out; unsigned long long *cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 4096;
  unsigned long long h[256];
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
      if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
      if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
    printf("mode %d: %.1f shader cycles per MFMA (16 MFMAs x %d iterations per wave, all 256 CUs)\n", mode, avg / (16.0 * iters), iters);
  }
  return 0;
}
