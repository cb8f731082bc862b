// Probe: cost of DEPENDENT back-to-back v_mfma_f32_32x32x2_f32 (same accumulator) against independent ones.
//   mode 0: 8 accumulators, 4 consecutive MFMAs per accumulator (the register-resident chain kernel's order)
//   mode 1: 8 accumulators, 2 consecutive MFMAs per accumulator, two passes
//   mode 2: 8 accumulators round-robin (no MFMA depends on the one in front of it)
//   mode 3: pairs interleaved: a0 b0 a1 b1 a2 b2 a3 b3 on accumulators (t, t+1)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define MF(c, x, y) c = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c, 0, 0, 0)
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float *out, unsigned long long *cyc, int iters) {
  floatx16 acc[8];
  for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = threadIdx.x * 0.002f - i; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int t = 0; t < 8; ++t) { MF(acc[t], a[0], b[0]); MF(acc[t], a[1], b[1]); MF(acc[t], a[2], b[2]); MF(acc[t], a[3], b[3]); }
    } else if (MODE == 1) {
#pragma unroll
      for (int t = 0; t < 8; ++t) { MF(acc[t], a[0], b[0]); MF(acc[t], a[1], b[1]); }
#pragma unroll
      for (int t = 0; t < 8; ++t) { MF(acc[t], a[2], b[2]); MF(acc[t], a[3], b[3]); }
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 8; ++t) MF(acc[t], a[j], b[j]);
    } else {
#pragma unroll
      for (int t = 0; t < 8; t += 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { MF(acc[t], a[j], b[j]); MF(acc[t + 1], a[j], b[j]); }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float *out; unsigned long long *cyc;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
  const int iters = 4096;
  unsigned long long h[256];
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
      if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
      if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
      if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
    printf("mode %d: %.2f shader cycles per MFMA\n", mode, avg / (32.0 * iters));
  }
  return 0;
}
