// Probe: which fp32 MFMA shape sustains more FLOP/s under the chip's clock management (MI355X_MICROARCH.md, DVFS give-back
// item 7)?  One wave per SIMD, 256 accumulator registers per wave (a 128 x 128 output tile), operands re-read from LDS
// (random data) every k-step, all 256 CUs, back-to-back launches for ~2 s per shape.
//   shape 0: v_mfma_f32_32x32x2_f32, 16 tiles, 8 ds_read_b32 per 16 MFMAs (2 samples)
//   shape 1: v_mfma_f32_16x16x4_f32, 64 tiles, 16 ds_read_b32 per 64 MFMAs (4 samples)
//   shape 2: shape 0 with operands held in registers (no LDS traffic)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rnd(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return (float)(x & 0xffff) / 32768.f - 1.f;
}

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void probe(float *out, unsigned long long *stamp, int iters) {
  __shared__ float lds[16 * 512];
  for (int i = threadIdx.x; i < 16 * 512; i += 256) lds[i] = rnd(i * 2654435761u + blockIdx.x);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  unsigned long long t0 = 0, r0 = 0;
  if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  float s = 0.f;
  if (SHAPE == 0 || SHAPE == 2) {
    floatx16 acc[16];
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int l31 = lane & 31, h = lane >> 5;
    float a[4], b[4];
    for (int t = 0; t < 4; ++t) { a[t] = lds[h * 512 + 32 * t + l31]; b[t] = lds[h * 512 + 256 + 32 * t + l31]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int sl = 0; sl < 8; ++sl) {
        if (SHAPE == 0) {
#pragma unroll
          for (int t = 0; t < 4; ++t) { a[t] = lds[(2 * sl + h) * 512 + 32 * t + l31]; b[t] = lds[(2 * sl + h) * 512 + 256 + 32 * t + l31]; }
        }
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) acc[4 * ta + tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], b[tb], acc[4 * ta + tb], 0, 0, 0);
      }
    }
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  } else {
    floatx4 acc[64];
    for (int t = 0; t < 64; ++t) for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
    const int l15 = lane & 15, k = lane >> 4;
    float a[8], b[8];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
#pragma unroll
        for (int t = 0; t < 8; ++t) { a[t] = lds[(4 * sl + k) * 512 + 16 * t + l15]; b[t] = lds[(4 * sl + k) * 512 + 256 + 16 * t + l15]; }
#pragma unroll
        for (int ta = 0; ta < 8; ++ta)
#pragma unroll
          for (int tb = 0; tb < 8; ++tb) acc[8 * ta + tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], b[tb], acc[8 * ta + tb], 0, 0, 0);
      }
    }
    for (int t = 0; t < 64; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    stamp[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime() - t0;
    stamp[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

int main() {
  float *out; unsigned long long *st;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&st, 256 * 32);
  const int iters = 2000;            // 16 samples per iteration per wave
  unsigned long long h[1024];
  for (int shape = 0; shape < 3; ++shape) {
    double wall = 0; int n = 0;
    auto T0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count() < 2.5) {
      auto a = std::chrono::steady_clock::now();
      for (int r = 0; r < 10; ++r) {
        if (shape == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, out, st, iters);
        if (shape == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, out, st, iters);
        if (shape == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 0, 0, out, st, iters);
      }
      hipDeviceSynchronize();
      wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count() / 10; ++n;
    }
    hipMemcpy(h, st, 256 * 32, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < 256; ++i) { cyc += (double)h[4 * i]; rt += (double)h[4 * i + 1]; }
    const double flop = 256.0 * 4 * iters * 16.0 * 2 * 128 * 128;     // CUs x waves x iterations x samples x 2 x tile
    printf("shape %d: last batch %.3f ms/launch -> %.1f TFLOP/s; in-kernel clock %.3f GHz; %.1f shader cycles per 4096 flop\n", shape,
           wall * 1e3, flop / wall / 1e12, cyc / rt * 0.1, cyc / 256 / (iters * 16.0 * 2 * 128 * 128 / 4096));
  }
  return 0;
}
