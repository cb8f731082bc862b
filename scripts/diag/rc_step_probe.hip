// Probe: the step pattern of the register-resident chain kernel (mlp_rc.hip) in isolation -- which ingredient costs the
// ~15 % over 64 cycles per MFMA seen in its chunk phase?  One wave per SIMD, 8 accumulator tiles, B operands in registers.
//   mode 0: steps of 4 dependent MFMAs, A operand constant (no LDS)
//   mode 1: + one ds_read_b128 per step (hand-issued, lgkmcnt(1)), operands rotate through 3 registers
//   mode 2: + 8 LDS-DMA pieces (global_load_lds_dwordx4) per 32 steps, vmcnt(0) + s_barrier once per 32 steps
//   mode 3: mode 2 without the barrier
//   mode 10: mode 3 with the ds_read between MFMA 1 and 2 of a step and the piece between MFMA 2 and 3;  mode 11: mode 10
//            without any pieces
//   mode 12: a plain global_load_dwordx4 (to registers, never used) in place of each LDS-DMA piece of mode 3;  mode 13: a
//            global_store_dwordx4 in their place
//   mode 4 / 5: 4 / 16 pieces per 32 steps (no barrier);  mode 6: 8 pieces, never waited for
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int OFF> __device__ __forceinline__ void rd(floatx4 &d, unsigned a) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float *out, const float *img, unsigned long long *cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float ring[3 * 8192 + 1024];
  for (int i = threadIdx.x; i < 3 * 8192 + 1024; i += 256) ring[i] = (float)(i & 255) * 0.001f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  floatx16 acc[8], B;
  for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int r = 0; r < 16; ++r) B[r] = lane * 0.01f + r;
  floatx4 A[3];
  floatx4 stage[8];
  for (int k = 0; k < 8; ++k) stage[k] = floatx4{0.f, 0.f, 0.f, 0.f};
  const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)ring + lane * 16;
  rd<0>(A[0], base); rd<4096>(A[1], base);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  A[2] = A[0];
  const float *src = img + (size_t)blockIdx.x * 8192 + lane * 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int slot = 0;
  for (int it = 0; it < iters; ++it) {       // one "chunk": 32 steps
    const unsigned sa = base + slot * 32768;
    float *dst = ring + ((slot + 2) % 3) * 8192;
    slot = (slot + 1) % 3;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int q = i / 8, t = i % 8;
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 18) {
        // register staging with EXACT waits: LDS operations return in order, so the operand read of this step has landed once
        // at most (1 + the ds_writes issued behind it) operations are outstanding -- a flat lgkmcnt(1) also waits for the writes
        const int w2 = (it > 0 && i - 2 >= 0 && i - 2 < 16 && ((i - 2) & 1) == 0) ? 1 : 0;
        const int w1 = (it > 0 && i - 1 >= 0 && i - 1 < 16 && ((i - 1) & 1) == 0) ? 1 : 0;
        if (w1 + w2 == 0) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        else if (w1 + w2 == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
      } else if (MODE >= 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
      const floatx4 a = A[(MODE >= 1) ? (i + 2 * 0) % 3 : 0];
      __builtin_amdgcn_sched_barrier(0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, B[4 * q + 0], acc[t], 0, 0, 0);
      if (MODE == 10 || MODE == 11) {
        __builtin_amdgcn_sched_barrier(0);
        switch ((i + 2) % 8) {
          case 0: rd<0 * 4096>(A[(i + 2) % 3], sa); break;
          case 1: rd<1 * 4096>(A[(i + 2) % 3], sa); break;
          case 2: rd<2 * 4096>(A[(i + 2) % 3], sa); break;
          case 3: rd<3 * 4096>(A[(i + 2) % 3], sa); break;
          case 4: rd<4 * 4096>(A[(i + 2) % 3], sa); break;
          case 5: rd<5 * 4096>(A[(i + 2) % 3], sa); break;
          case 6: rd<6 * 4096>(A[(i + 2) % 3], sa); break;
          default: rd<7 * 4096>(A[(i + 2) % 3], sa); break;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, B[4 * q + 1], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE >= 1 && MODE != 10 && MODE != 11) {
        switch ((i + 2) % 8) {
          case 0: rd<0 * 4096>(A[(i + 2) % 3], sa); break;
          case 1: rd<1 * 4096>(A[(i + 2) % 3], sa); break;
          case 2: rd<2 * 4096>(A[(i + 2) % 3], sa); break;
          case 3: rd<3 * 4096>(A[(i + 2) % 3], sa); break;
          case 4: rd<4 * 4096>(A[(i + 2) % 3], sa); break;
          case 5: rd<5 * 4096>(A[(i + 2) % 3], sa); break;
          case 6: rd<6 * 4096>(A[(i + 2) % 3], sa); break;
          default: rd<7 * 4096>(A[(i + 2) % 3], sa); break;
        }
      }
      {
        // pieces issued at this step: modes 2/3/6: one at every even step of the second half; 4: every 4th; 5: every step;
        // 7: two at every 4th step; 8: eight at step 16; 9: four at steps 16 and 24
        int n_here = 0, first = 0;
        if (MODE == 2 || MODE == 3 || MODE == 6 || MODE == 10) { n_here = (i >= 16 && (i & 1) == 0) ? 1 : 0; first = (i - 16) >> 1; }
        if (MODE == 4) { n_here = (i >= 16 && (i & 3) == 0) ? 1 : 0; first = (i - 16) >> 2; }
        if (MODE == 5) { n_here = i >= 16 ? 1 : 0; first = i - 16; }
        if (MODE == 7) { n_here = (i >= 16 && (i & 3) == 0) ? 2 : 0; first = ((i - 16) >> 2) * 2; }
        if (MODE == 8) { n_here = i == 16 ? 8 : 0; first = 0; }
        if (MODE == 9) { n_here = (i == 16 || i == 24) ? 4 : 0; first = i == 16 ? 0 : 4; }
        if (MODE == 12 || MODE == 13) { n_here = 0; }
        // 14: mode 3 with the waves' pieces on DIFFERENT steps (wave parity picks even / odd steps): do the four lock-stepped
        // waves of a CU queue up behind one another at the texture-address unit?   15: mode 3 launched with ONE wave per CU
        if (MODE == 14) { n_here = (i >= 16 && ((i + wave) & 1) == 0 && i - (wave & 1) < 32) ? 1 : 0; first = (i - 16) >> 1; }
        if (MODE == 15) { n_here = (i >= 16 && (i & 1) == 0) ? 1 : 0; first = (i - 16) >> 1; }
        // 16: pieces spread over all four quarter positions: wave w issues at steps 16 + 2 k + (w & 1), waves 2, 3 one MFMA later
        if (MODE == 16) { n_here = (i >= 16 && ((i + wave) & 1) == 0) ? 1 : 0; first = (i - 16) >> 1; }
        // 17: register staging -- a plain global_load_dwordx4 per piece at the even steps of the second half (as 12) and, one
        // chunk later, a ds_write_b128 of what arrived at the even steps of the FIRST half (vmcnt waited before the first)
        if (MODE == 17 || MODE == 18) {
          if (i >= 16 && (i & 1) == 0) {
            const int k = (i - 16) >> 1;
            const int p = (wave + 4 * k) & 31;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stage[k]) : "v"(src + p * 256) : "memory");
          }
          if (i < 16 && (i & 1) == 0 && it > 0) {
            const int k = i >> 1;
            const int p = (wave + 4 * k) & 31;
            if (k == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned da = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)(dst + p * 256) + lane * 16;
            asm volatile("ds_write_b128 %0, %1" ::"v"(da), "v"(stage[k]) : "memory");
          }
        }
        if ((MODE == 12 || MODE == 13) && i >= 16 && (i & 1) == 0) {
          const int p = (wave + 4 * ((i - 16) >> 1)) & 31;
          const float *gp = src + p * 256;
          if (MODE == 12) {
            floatx4 tmp;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(tmp) : "v"(gp) : "memory");
          } else {
            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(gp), "v"(A[0]) : "memory");
          }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k < n_here) {
            const int p = (wave + 4 * (first + k)) & 31;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * 256),
                                             (__attribute__((address_space(3))) void *)(dst + p * 256), 16, 0, 0);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, B[4 * q + 2], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, B[4 * q + 3], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE >= 2 && MODE != 11 && i == 15) {
        if (MODE == 12) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE != 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE == 2) __builtin_amdgcn_s_barrier();
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float *out, *img; unsigned long long *cyc;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8); (void)hipMalloc(&img, 256 * 8192 * 4 + 65536);
  (void)hipMemset(img, 0, 256 * 8192 * 4 + 65536);
  float *wbuf; (void)hipMalloc(&wbuf, 256 * 8192 * 4 + 65536);
  const int iters = 540;
  unsigned long long h[256];
  for (int mode = 0; mode < 19; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, out, img, cyc, iters);
      if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, out, img, cyc, iters);
      if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 0, 0, out, img, cyc, iters);
      if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(256), 0, 0, out, img, cyc, iters);
      if (mode == 4) hipLaunchKernelGGL(probe<4>, dim3(256), dim3(256), 0, 0, out, img, cyc, iters);
      if (mode == 5) hipLaunchKernelGGL(probe<5>, dim3(256), dim3(256), 0, 0, out, img, cyc, iters);
      if (mode == 6) hipLaunchKernelGGL(probe<6>, dim3(256), dim3(256), 0, 0, out, img, cyc, iters);
      if (mode == 7) hipLaunchKernelGGL(probe<7>, dim3(256), dim3(256), 0, 0, out, img, cyc, iters);
      if (mode == 8) hipLaunchKernelGGL(probe<8>, dim3(256), dim3(256), 0, 0, out, img, cyc, iters);
      if (mode == 9) hipLaunchKernelGGL(probe<9>, dim3(256), dim3(256), 0, 0, out, img, cyc, iters);
      if (mode == 10) hipLaunchKernelGGL(probe<10>, dim3(256), dim3(256), 0, 0, out, img, cyc, iters);
      if (mode == 11) hipLaunchKernelGGL(probe<11>, dim3(256), dim3(256), 0, 0, out, img, cyc, iters);
      if (mode == 12) hipLaunchKernelGGL(probe<12>, dim3(256), dim3(256), 0, 0, out, img, cyc, iters);
      if (mode == 13) hipLaunchKernelGGL(probe<13>, dim3(256), dim3(256), 0, 0, out, (const float *)wbuf, cyc, iters);
      if (mode == 18) hipLaunchKernelGGL(probe<18>, dim3(256), dim3(256), 0, 0, out, img, cyc, iters);
      if (mode == 17) hipLaunchKernelGGL(probe<17>, dim3(256), dim3(256), 0, 0, out, img, cyc, iters);
      if (mode == 14) hipLaunchKernelGGL(probe<14>, dim3(256), dim3(256), 0, 0, out, img, cyc, iters);
      if (mode == 15) hipLaunchKernelGGL(probe<15>, dim3(256), dim3(64), 0, 0, out, img, cyc, iters);
      if (mode == 16) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(128), 0, 0, out, img, cyc, iters);
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
    printf("mode %d: %.2f shader cycles per MFMA (%.0f per chunk of 128)\n", mode, avg / (128.0 * iters), avg / iters);
  }
  return 0;
}
