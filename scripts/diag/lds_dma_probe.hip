// Probe: does global_load_lds_dwordx4 address LDS above 64 KB on gfx950, and what is the per-lane destination stride?
// Each wave DMA-copies 1 KB pieces of a global pattern to chosen LDS byte offsets; the workgroup reads LDS back with ds_read.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(256, 1) void probe(const float *src, float *out, int n_floats) {
  __shared__ __attribute__((aligned(16))) float lds[3 * 10240];   // 120 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 3 * 10240; i += 256) lds[i] = -1.f;
  __syncthreads();
  // pieces: piece p (1 KB = 256 floats) of src -> LDS float offset p * 256, for p = wave, wave + 4, ... < 120
  for (int p = wave; p < 120; p += 4)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * 256 + lane * 4),
                                     (__attribute__((address_space(3))) void *)(lds + p * 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = tid; i < n_floats; i += 256) out[i] = lds[i];
}
int main() {
  const int n = 3 * 10240;
  std::vector<float> h(n), o(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *dout;
  hipMalloc(&d, n * 4); hipMalloc(&dout, n * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d, dout, n);
  hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
  int bad = 0, first = -1;
  for (int i = 0; i < n; ++i) if (o[i] != h[i]) { if (first < 0) first = i; ++bad; }
  printf("mismatches %d of %d, first at float %d (byte %d): got %g expected %g\n", bad, n, first, first * 4, first >= 0 ? o[first] : 0.f, first >= 0 ? h[first] : 0.f);
  for (int p = 0; p < 120; p += 8) printf("piece %3d (byte %6d): lds[0]=%g lds[1]=%g lds[4]=%g lds[255]=%g\n", p, p * 1024, o[p * 256], o[p * 256 + 1], o[p * 256 + 4], o[p * 256 + 255]);
  return 0;
}
