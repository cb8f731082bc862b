// pad_cols_probe.hip -- the index expression of k_pad_cols_multi before and after commit b4fb5c1, on a destination that is a
// COLUMN SLICE of a wider matrix (what the compact-dX0 change of that commit started to pass: W0c[:, :12] and W0c[:, 12:52] of
// a [256, 52] tensor), with canary values around the destination.  Reports how many canary elements each expression overwrites
// and how far behind the destination tensor's last element the furthest write lands.  Everything stays inside one allocation of
// this program: nothing faults.
//     hipcc --offload-arch=gfx950 -O2 -o scripts/diag/pad_cols_probe scripts/diag/pad_cols_probe.hip && scripts/diag/pad_cols_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Args { const float *src; float *dst; int rows, cols, width; long ld_src, ld_dst; };

template <bool OLD>
__global__ void k_copy(Args a) {
  const long n = OLD ? (long)a.rows * a.ld_dst : (long)a.rows * a.width;   // launch size of the respective host code
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  if (OLD) {      // a5aee36 and before: the destination is ASSUMED to be a whole [rows, ld_dst] matrix
    const long r = e / a.ld_dst, c = e - r * a.ld_dst;
    a.dst[e] = c < a.cols ? a.src[r * a.ld_src + c] : 0.f;
  } else {        // b4fb5c1: width columns per row at pitch ld_dst
    const long r = e / a.width, c = e - r * a.width;
    a.dst[r * a.ld_dst + c] = c < a.cols ? a.src[r * a.ld_src + c] : 0.f;
  }
}

int main() {
  const int rows = 256, src_cols = 106, pitch = 52, col0 = 12, width = 40;   // W0[:, 66:106] -> W0c[:, 12:52], W0c = [256, 52]
  const long w0c = (long)rows * pitch, guard = 4096;                           // W0c followed by `guard` canary floats
  std::vector<float> h_src((size_t)rows * src_cols, 1.f), h(w0c + guard);
  float *d_src, *d;
  hipMalloc(&d_src, h_src.size() * 4);
  hipMalloc(&d, h.size() * 4);
  hipMemcpy(d_src, h_src.data(), h_src.size() * 4, hipMemcpyHostToDevice);
  for (int old = 1; old >= 0; --old) {
    for (auto &v : h) v = 7.f;
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    Args a{d_src + 66, d + col0, rows, width, width, src_cols, pitch};
    const long n = old ? (long)rows * pitch : (long)rows * width;
    if (old) hipLaunchKernelGGL(k_copy<true>, dim3((n + 255) / 256), dim3(256), 0, 0, a);
    else hipLaunchKernelGGL(k_copy<false>, dim3((n + 255) / 256), dim3(256), 0, 0, a);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    long outside = 0, beyond = 0, furthest = -1;
    for (long i = 0; i < (long)h.size(); ++i) {
      const bool inside = i < w0c && (i % pitch) >= col0 && (i % pitch) < col0 + width;
      if (!inside && h[i] != 7.f) { ++outside; if (i >= w0c) { ++beyond; furthest = i - w0c; } }
    }
    printf("%s index expression: %ld elements written outside the [256, 40] destination slice, %ld of them BEHIND the end of the "
           "[256, 52] tensor it belongs to (furthest: %ld floats = %ld bytes past the end)\n",
           old ? "old (dst[e], e < rows * ld_dst)   " : "new (dst[r * ld_dst + c], c < width)", outside, beyond,
           furthest >= 0 ? furthest + 1 : 0L, (furthest >= 0 ? furthest + 1 : 0L) * 4);
  }
  return 0;
}
