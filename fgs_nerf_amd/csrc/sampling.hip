// sampling.hip -- ray/AABB entry, per-ray sample counts, packed (ray_id, step_id) emission,
// NDC / background samplers and the nearest-voxel mask lookup.
// Reference operators: model/cuda/render_utils_kernel.cu:11-424 (cited per kernel).
#include "fgs_common.h"

namespace {

struct RaySeg {
  float t_min, t_max;
};

// render_utils_kernel.cu:11-35
__device__ __forceinline__ RaySeg ray_aabb(const float *o, const float *d, const float *lo, const float *hi,
                                           float near, float far) {
  float a[3], b[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = (d[c] == 0.f) ? (float)1e-6 : d[c];
    a[c] = (hi[c] - o[c]) / v;
    b[c] = (lo[c] - o[c]) / v;
  }
  const float en = fmaxf(fmaxf(fminf(a[0], b[0]), fminf(a[1], b[1])), fminf(a[2], b[2]));
  const float ex = fminf(fminf(fmaxf(a[0], b[0]), fmaxf(a[1], b[1])), fmaxf(a[2], b[2]));
  RaySeg s;
  s.t_min = fmaxf(fminf(en, far), near);
  s.t_max = fmaxf(fminf(ex, far), near);
  return s;
}

// render_utils_kernel.cu:37-55
__device__ __forceinline__ int64_t ray_n_samples(const float *d, float t_min, float t_max, float stepdist) {
  const float rn = fgs_rnorm3(d[0], d[1], d[2]);
  const float c = ceilf((t_max - t_min) * rn / stepdist);
  return (int64_t)fmax((double)c, 1.);
}

__global__ void k_t_minmax(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                           const float *__restrict__ xyz_min, const float *__restrict__ xyz_max, float near, float far,
                           int64_t n_rays, float *__restrict__ t_min, float *__restrict__ t_max) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
  const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
  const float lo[3] = {xyz_min[0], xyz_min[1], xyz_min[2]}, hi[3] = {xyz_max[0], xyz_max[1], xyz_max[2]};
  const RaySeg s = ray_aabb(o, d, lo, hi, near, far);
  t_min[r] = s.t_min;
  t_max[r] = s.t_max;
}

__global__ void k_n_samples(const float *__restrict__ rays_d, const float *__restrict__ t_min,
                            const float *__restrict__ t_max, float stepdist, int64_t n_rays,
                            int64_t *__restrict__ n_samples) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
  n_samples[r] = ray_n_samples(d, t_min[r], t_max[r], stepdist);
}

// render_utils_kernel.cu:57-79
__global__ void k_start_dir(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                            const float *__restrict__ t_min, int64_t n_rays, float *__restrict__ rays_start,
                            float *__restrict__ rays_dir) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
  const float rn = fgs_rnorm3(d[0], d[1], d[2]);
  const float t = t_min[r];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    rays_start[3 * r + c] = fmaf(d[c], t, rays_o[3 * r + c]);
    rays_dir[3 * r + c] = d[c] / rn;
  }
}

// t_min/t_max/n_steps in one pass (first half of sample_pts_on_rays_cuda, :203-210)
__global__ void k_count(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                        const float *__restrict__ xyz_min, const float *__restrict__ xyz_max, float near, float far,
                        float stepdist, int64_t n_rays, int64_t *__restrict__ n_steps, float *__restrict__ t_min,
                        float *__restrict__ t_max) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
  const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
  const float lo[3] = {xyz_min[0], xyz_min[1], xyz_min[2]}, hi[3] = {xyz_max[0], xyz_max[1], xyz_max[2]};
  const RaySeg s = ray_aabb(o, d, lo, hi, near, far);
  t_min[r] = s.t_min;
  t_max[r] = s.t_max;
  n_steps[r] = ray_n_samples(d, s.t_min, s.t_max, stepdist);
}

// Exclusive prefix sum of int64 counts into out[n+1] (out[n] = total).  One 1024-thread workgroup;
// each thread owns a contiguous slice, slices are combined with a wave-shuffle + LDS scan.
// Replaces N_steps.cumsum(0) and N_steps.sum().item() of render_utils_kernel.cu:211-212.
// GUARD (fgs_exclusive_scan_guard_i64): the capacity guard of a sync-free step (fgs_count_guard, gridopt.hip) applied while
// the offsets are written -- every offset cut at `capacity`, flags / total updated from the uncut total -- one launch less.
constexpr int SCAN_THREADS = 1024;
template <bool GUARD>
__global__ __launch_bounds__(SCAN_THREADS) void k_exclusive_scan_i64(const int64_t *__restrict__ in, int64_t n,
                                                                      int64_t *__restrict__ out, int64_t capacity,
                                                                      int *__restrict__ flags, int64_t *__restrict__ total) {
  __shared__ int64_t wave_tot[SCAN_THREADS / FGS_WAVE];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t per = (n + SCAN_THREADS - 1) / SCAN_THREADS;
  const int64_t lo = (int64_t)tid * per, hi = (lo + per < n) ? lo + per : n;
  int64_t sum = 0;
  // (up to 8 elements per thread -- 8192 rays -- are read ONCE, all loads in flight together, and kept in registers for the second
  // pass: the two dependent passes over memory were 11 us for 8192 elements)
  constexpr int KEEP = 8;
  int64_t v[KEEP];
  const bool kept = per <= KEEP;
  if (kept) {
#pragma unroll
    for (int u = 0; u < KEEP; ++u) v[u] = (lo + u < hi) ? in[lo + u] : 0;
#pragma unroll
    for (int u = 0; u < KEEP; ++u) sum += v[u];
  } else {
    for (int64_t i = lo; i < hi; ++i) sum += in[i];
  }
  // inclusive scan across the wave
  int64_t inc = sum;
#pragma unroll
  for (int off = 1; off < FGS_WAVE; off <<= 1) {
    const int64_t up = __shfl_up(inc, off, FGS_WAVE);
    if (lane >= off) inc += up;
  }
  if (lane == FGS_WAVE - 1) wave_tot[wv] = inc;
  __syncthreads();
  int64_t wave_base = 0;
  for (int w = 0; w < wv; ++w) wave_base += wave_tot[w];
  int64_t run = wave_base + inc - sum;
  if (kept) {
#pragma unroll
    for (int u = 0; u < KEEP; ++u) {
      if (lo + u < hi) out[lo + u] = (GUARD && run > capacity) ? capacity : run;
      run += v[u];
    }
  } else {
    for (int64_t i = lo; i < hi; ++i) {
      out[i] = (GUARD && run > capacity) ? capacity : run;
      run += in[i];
    }
  }
  if (tid == SCAN_THREADS - 1) {
    int64_t tot = 0;
    for (int w = 0; w < SCAN_THREADS / FGS_WAVE; ++w) tot += wave_tot[w];
    if (GUARD) {
      const int over = tot > capacity ? 1 : 0;
      flags[1] = over;
      if (over) flags[0] = 1;
      if (total) *total += over ? capacity : tot;
      if (over) tot = capacity;
    }
    out[n] = tot;
  }
}

// Second half of sample_pts_on_rays_cuda (:213-241): sample idx -> (ray, step) by binary search in the
// exclusive prefix sum (replaces the scatter-1 + cumsum + __set_step_id trio, :144-164), then the
// point and its out-of-bbox flag (:166-194).
__global__ void k_emit(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                       const float *__restrict__ xyz_min, const float *__restrict__ xyz_max, float stepdist,
                       int64_t n_rays, const float *__restrict__ t_min, const int64_t *__restrict__ cumsum,
                       int64_t capacity, float *__restrict__ rays_pts, uint8_t *__restrict__ mask_outbbox,
                       int64_t *__restrict__ ray_id, int64_t *__restrict__ step_id) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = cumsum[n_rays];
  if (idx >= total || idx >= capacity) return;
  // largest r with cumsum[r] <= idx  (n_steps >= 1 for every ray, so it is unique)
  int64_t lo = 0, hi = n_rays;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (cumsum[mid] <= idx) lo = mid; else hi = mid;
  }
  const int64_t r = lo;
  const int64_t s = idx - cumsum[r];
  const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
  const float rn = fgs_rnorm3(d[0], d[1], d[2]);
  const float t = t_min[r];
  const float dist = stepdist * (float)(int)s;
  float p[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float start = fmaf(d[c], t, rays_o[3 * r + c]);
    const float dir = d[c] / rn;
    p[c] = fmaf(dir, dist, start);
    rays_pts[3 * idx + c] = p[c];
  }
  mask_outbbox[idx] = (uint8_t)((xyz_min[0] > p[0]) | (xyz_min[1] > p[1]) | (xyz_min[2] > p[2]) |
                                (xyz_max[0] < p[0]) | (xyz_max[1] < p[1]) | (xyz_max[2] < p[2]));
  ray_id[idx] = r;
  step_id[idx] = s;
}

// render_utils_kernel.cu:244-270
__global__ void k_sample_ndc(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                             const float *__restrict__ xyz_min, const float *__restrict__ xyz_max, int64_t n_samples,
                             int64_t n_rays, float *__restrict__ rays_pts, uint8_t *__restrict__ mask_outbbox) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_samples * n_rays) return;
  const int64_t r = idx / n_samples, s = idx % n_samples;
  const float dist = ((float)(int)s) / (float)(int)(n_samples - 1);
  float p[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    p[c] = fmaf(rays_d[3 * r + c], dist, rays_o[3 * r + c]);
    rays_pts[3 * idx + c] = p[c];
  }
  mask_outbbox[idx] = (uint8_t)((xyz_min[0] > p[0]) | (xyz_min[1] > p[1]) | (xyz_min[2] > p[2]) |
                                (xyz_max[0] < p[0]) | (xyz_max[1] < p[1]) | (xyz_max[2] < p[2]));
}

// render_utils_kernel.cu:300-340 (double literals of the reference keep their promotions)
__global__ void k_sample_bg(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                            const float *__restrict__ t_max, float bg_preserve, int64_t n_samples, int64_t n_rays,
                            float *__restrict__ rays_pts) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_samples * n_rays) return;
  const int64_t r = idx / n_samples, s = idx % n_samples;
  const float t_inner = t_max[r];
  const float frac = ((float)(int)s) / (float)(int)n_samples;
  const float ori_t_outer = (float)((double)t_inner - 1. + 1. / (1. - (double)frac));
  float q[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) q[c] = fmaf(rays_d[3 * r + c], ori_t_outer, rays_o[3 * r + c]);
  const float t_outer = sqrtf(fmaf(q[2], q[2], fmaf(q[1], q[1], q[0] * q[0])));
  const float m = fmaxf(fabsf(q[0]), fmaxf(fabsf(q[1]), fabsf(q[2])));
  const float R = t_outer / m;
  const float o2i = (float)((double)(R * R / (t_outer * t_outer)) * (1. - (double)bg_preserve) +
                            (double)(R / t_outer * bg_preserve));
#pragma unroll
  for (int c = 0; c < 3; ++c) rays_pts[3 * idx + c] = q[c] * o2i;
}

// render_utils_kernel.cu:373-392
__global__ void k_maskcache(const uint8_t *__restrict__ world, const float *__restrict__ xyz,
                            const float *__restrict__ scale, const float *__restrict__ shift, int sz_i, int sz_j,
                            int sz_k, int64_t n_pts, uint8_t *__restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pts) return;
  const int i = (int)roundf(fmaf(xyz[3 * p + 0], scale[0], shift[0]));
  const int j = (int)roundf(fmaf(xyz[3 * p + 1], scale[1], shift[1]));
  const int k = (int)roundf(fmaf(xyz[3 * p + 2], scale[2], shift[2]));
  uint8_t v = 0;
  if (fgs_in(i, sz_i) && fgs_in(j, sz_j) && fgs_in(k, sz_k)) v = world[(int64_t)i * sz_j * sz_k + (int64_t)j * sz_k + k];
  out[p] = v;
}

}  // namespace

FGS_API int fgs_infer_t_minmax(const float *rays_o, const float *rays_d, const float *xyz_min, const float *xyz_max,
                               float near, float far, int64_t n_rays, float *t_min, float *t_max, fgs_stream_t stream) {
  FGS_REQUIRE(n_rays >= 0 && n_rays < FGS_MAX_ELEMS, FGS_E_RANGE, "fgs_infer_t_minmax: n_rays=%lld", (long long)n_rays);
  if (n_rays == 0) return 0;
  FGS_REQUIRE(rays_o && rays_d && xyz_min && xyz_max && t_min && t_max, FGS_E_INVALID, "fgs_infer_t_minmax: null pointer");
  hipLaunchKernelGGL(k_t_minmax, dim3(fgs_blocks(n_rays)), dim3(FGS_BLOCK), 0, fgs_s(stream), rays_o, rays_d, xyz_min,
                     xyz_max, near, far, n_rays, t_min, t_max);
  FGS_LAUNCH_OK("fgs_infer_t_minmax");
  return 0;
}

FGS_API int fgs_infer_n_samples(const float *rays_d, const float *t_min, const float *t_max, float stepdist,
                                int64_t n_rays, int64_t *n_samples, fgs_stream_t stream) {
  FGS_REQUIRE(n_rays >= 0 && n_rays < FGS_MAX_ELEMS, FGS_E_RANGE, "fgs_infer_n_samples: n_rays=%lld", (long long)n_rays);
  if (n_rays == 0) return 0;
  FGS_REQUIRE(rays_d && t_min && t_max && n_samples, FGS_E_INVALID, "fgs_infer_n_samples: null pointer");
  hipLaunchKernelGGL(k_n_samples, dim3(fgs_blocks(n_rays)), dim3(FGS_BLOCK), 0, fgs_s(stream), rays_d, t_min, t_max,
                     stepdist, n_rays, n_samples);
  FGS_LAUNCH_OK("fgs_infer_n_samples");
  return 0;
}

FGS_API int fgs_infer_ray_start_dir(const float *rays_o, const float *rays_d, const float *t_min, int64_t n_rays,
                                    float *rays_start, float *rays_dir, fgs_stream_t stream) {
  FGS_REQUIRE(n_rays >= 0 && n_rays < FGS_MAX_ELEMS, FGS_E_RANGE, "fgs_infer_ray_start_dir: n_rays=%lld", (long long)n_rays);
  if (n_rays == 0) return 0;
  FGS_REQUIRE(rays_o && rays_d && t_min && rays_start && rays_dir, FGS_E_INVALID, "fgs_infer_ray_start_dir: null pointer");
  hipLaunchKernelGGL(k_start_dir, dim3(fgs_blocks(n_rays)), dim3(FGS_BLOCK), 0, fgs_s(stream), rays_o, rays_d, t_min,
                     n_rays, rays_start, rays_dir);
  FGS_LAUNCH_OK("fgs_infer_ray_start_dir");
  return 0;
}

FGS_API int fgs_sample_count(const float *rays_o, const float *rays_d, const float *xyz_min, const float *xyz_max,
                             float near, float far, float stepdist, int64_t n_rays, int64_t *n_steps, float *t_min,
                             float *t_max, int64_t *steps_cumsum, fgs_stream_t stream) {
  FGS_REQUIRE(n_rays > 0 && n_rays < FGS_MAX_ELEMS, FGS_E_RANGE, "fgs_sample_count: n_rays=%lld", (long long)n_rays);
  FGS_REQUIRE(rays_o && rays_d && xyz_min && xyz_max && n_steps && t_min && t_max && steps_cumsum, FGS_E_INVALID,
              "fgs_sample_count: null pointer");
  FGS_REQUIRE(stepdist > 0.f, FGS_E_INVALID, "fgs_sample_count: stepdist must be > 0");
  hipLaunchKernelGGL(k_count, dim3(fgs_blocks(n_rays)), dim3(FGS_BLOCK), 0, fgs_s(stream), rays_o, rays_d, xyz_min,
                     xyz_max, near, far, stepdist, n_rays, n_steps, t_min, t_max);
  FGS_LAUNCH_OK("fgs_sample_count/count");
  hipLaunchKernelGGL(k_exclusive_scan_i64<false>, dim3(1), dim3(SCAN_THREADS), 0, fgs_s(stream), (const int64_t *)n_steps,
                     n_rays, steps_cumsum, (int64_t)0, (int *)nullptr, (int64_t *)nullptr);
  FGS_LAUNCH_OK("fgs_sample_count/scan");
  return 0;
}

FGS_API int fgs_sample_emit(const float *rays_o, const float *rays_d, const float *xyz_min, const float *xyz_max,
                            float stepdist, int64_t n_rays, const float *t_min, const int64_t *steps_cumsum,
                            int64_t capacity, float *rays_pts, uint8_t *mask_outbbox, int64_t *ray_id, int64_t *step_id,
                            fgs_stream_t stream) {
  FGS_REQUIRE(n_rays > 0 && n_rays < FGS_MAX_ELEMS, FGS_E_RANGE, "fgs_sample_emit: n_rays=%lld", (long long)n_rays);
  FGS_REQUIRE(capacity >= 0 && capacity < FGS_MAX_ELEMS, FGS_E_RANGE, "fgs_sample_emit: capacity=%lld", (long long)capacity);
  if (capacity == 0) return 0;
  FGS_REQUIRE(rays_o && rays_d && xyz_min && xyz_max && t_min && steps_cumsum && rays_pts && mask_outbbox && ray_id && step_id,
              FGS_E_INVALID, "fgs_sample_emit: null pointer");
  hipLaunchKernelGGL(k_emit, dim3(fgs_blocks(capacity)), dim3(FGS_BLOCK), 0, fgs_s(stream), rays_o, rays_d, xyz_min,
                     xyz_max, stepdist, n_rays, t_min, steps_cumsum, capacity, rays_pts, mask_outbbox, ray_id, step_id);
  FGS_LAUNCH_OK("fgs_sample_emit");
  return 0;
}

FGS_API int fgs_sample_ndc_pts(const float *rays_o, const float *rays_d, const float *xyz_min, const float *xyz_max,
                               int64_t n_samples, int64_t n_rays, float *rays_pts, uint8_t *mask_outbbox,
                               fgs_stream_t stream) {
  FGS_REQUIRE(n_rays >= 0 && n_samples >= 0 && n_rays * n_samples < FGS_MAX_ELEMS, FGS_E_RANGE, "fgs_sample_ndc_pts: size");
  if (n_rays * n_samples == 0) return 0;
  FGS_REQUIRE(rays_o && rays_d && xyz_min && xyz_max && rays_pts && mask_outbbox, FGS_E_INVALID, "fgs_sample_ndc_pts: null pointer");
  hipLaunchKernelGGL(k_sample_ndc, dim3(fgs_blocks(n_rays * n_samples)), dim3(FGS_BLOCK), 0, fgs_s(stream), rays_o, rays_d,
                     xyz_min, xyz_max, n_samples, n_rays, rays_pts, mask_outbbox);
  FGS_LAUNCH_OK("fgs_sample_ndc_pts");
  return 0;
}

FGS_API int fgs_sample_bg_pts(const float *rays_o, const float *rays_d, const float *t_max, float bg_preserve,
                              int64_t n_samples, int64_t n_rays, float *rays_pts, fgs_stream_t stream) {
  FGS_REQUIRE(n_rays >= 0 && n_samples >= 0 && n_rays * n_samples < FGS_MAX_ELEMS, FGS_E_RANGE, "fgs_sample_bg_pts: size");
  if (n_rays * n_samples == 0) return 0;
  FGS_REQUIRE(rays_o && rays_d && t_max && rays_pts, FGS_E_INVALID, "fgs_sample_bg_pts: null pointer");
  hipLaunchKernelGGL(k_sample_bg, dim3(fgs_blocks(n_rays * n_samples)), dim3(FGS_BLOCK), 0, fgs_s(stream), rays_o, rays_d,
                     t_max, bg_preserve, n_samples, n_rays, rays_pts);
  FGS_LAUNCH_OK("fgs_sample_bg_pts");
  return 0;
}

FGS_API int fgs_maskcache_lookup(const uint8_t *world, const float *xyz, const float *xyz2ijk_scale,
                                 const float *xyz2ijk_shift, int sz_i, int sz_j, int sz_k, int64_t n_pts, uint8_t *out,
                                 fgs_stream_t stream) {
  FGS_REQUIRE(n_pts >= 0 && n_pts < FGS_MAX_ELEMS, FGS_E_RANGE, "fgs_maskcache_lookup: n_pts=%lld", (long long)n_pts);
  if (n_pts == 0) return 0;
  FGS_REQUIRE(world && xyz && xyz2ijk_scale && xyz2ijk_shift && out, FGS_E_INVALID, "fgs_maskcache_lookup: null pointer");
  FGS_REQUIRE(sz_i > 0 && sz_j > 0 && sz_k > 0, FGS_E_INVALID, "fgs_maskcache_lookup: empty mask volume");
  hipLaunchKernelGGL(k_maskcache, dim3(fgs_blocks(n_pts)), dim3(FGS_BLOCK), 0, fgs_s(stream), world, xyz, xyz2ijk_scale,
                     xyz2ijk_shift, sz_i, sz_j, sz_k, n_pts, out);
  FGS_LAUNCH_OK("fgs_maskcache_lookup");
  return 0;
}

// Exclusive prefix sum of int64 counts: out[0..n] (out[n] = total).  Used between the fused march kernel and the
// survivor kernels (per-ray survivor counts -> segment offsets), the same scan sample_count uses.
namespace {
// out[a][i][0..2] = src_a[sel[i]][0..2]: the four row gathers of a training batch (model/nerf_training.py:256-261: target,
// rays_o, rays_d, viewdirs by one index vector) in one launch, written where the captured step reads its inputs
struct BatchSrc { const float *p[4]; };
__global__ __launch_bounds__(FGS_BLOCK) void k_gather_batch(const int64_t *__restrict__ sel, int64_t n, int64_t n_src, BatchSrc src,
                                                            float *__restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 4 * n) return;
  const int64_t a = t / n, i = t - a * n;
  int64_t r = sel[i];
  r = r < 0 ? 0 : (r >= n_src ? n_src - 1 : r);        // (an index out of range must not become a fault)
  const float *s = src.p[a] + 3 * r;
  float *o = out + 3 * t;
  o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
}
}  // namespace

namespace {
__global__ __launch_bounds__(FGS_BLOCK) void k_copy_f32x4(const float4 *__restrict__ src, float4 *__restrict__ dst, int64_t n4,
                                                          const float *__restrict__ src_tail, float *__restrict__ dst_tail, int tail) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) dst[i] = src[i];
  if (i < tail) dst_tail[i] = src_tail[i];
}
}  // namespace

// dst[0..n) = src[0..n) (float32, both 16-byte aligned): the staged batch of a captured step into its static inputs as a kernel
// launch -- the runtime's blit path (hipMemcpyAsync device-to-device) measured 5 us + a 5 us gap per step in front of the graph.
FGS_API int fgs_copy_f32(const float *src, float *dst, int64_t n, fgs_stream_t stream) {
  FGS_REQUIRE(n >= 0 && n < FGS_MAX_ELEMS, FGS_E_RANGE, "fgs_copy_f32: n=%lld", (long long)n);
  if (n == 0) return 0;
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  FGS_REQUIRE(src && dst && (aligned || n < 4) && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3) == 0,
              FGS_E_INVALID, "fgs_copy_f32: null or unaligned pointer (16 bytes; 4 for fewer than four floats)");
  const int64_t n4 = n / 4;
  const int tail = (int)(n - 4 * n4);
  hipLaunchKernelGGL(k_copy_f32x4, dim3(fgs_blocks(n4 > 0 ? n4 : 1)), dim3(FGS_BLOCK), 0, fgs_s(stream),
                     reinterpret_cast<const float4 *>(src), reinterpret_cast<float4 *>(dst), n4, src + 4 * n4, dst + 4 * n4, tail);
  FGS_LAUNCH_OK("fgs_copy_f32");
  return 0;
}

FGS_API int fgs_gather_batch(const int64_t *sel, int64_t n, int64_t n_src, const float *src0, const float *src1, const float *src2,
                             const float *src3, float *out, fgs_stream_t stream) {
  FGS_REQUIRE(n >= 0 && n_src > 0 && n < FGS_MAX_ELEMS / 16, FGS_E_RANGE, "fgs_gather_batch: n=%lld of %lld", (long long)n,
              (long long)n_src);
  if (n == 0) return 0;
  FGS_REQUIRE(sel && src0 && src1 && src2 && src3 && out, FGS_E_INVALID, "fgs_gather_batch: null pointer");
  const BatchSrc src{{src0, src1, src2, src3}};
  hipLaunchKernelGGL(k_gather_batch, dim3(fgs_blocks(4 * n)), dim3(FGS_BLOCK), 0, fgs_s(stream), sel, n, n_src, src, out);
  FGS_LAUNCH_OK("fgs_gather_batch");
  return 0;
}

FGS_API int fgs_exclusive_scan_i64(const int64_t *in, int64_t n, int64_t *out, fgs_stream_t stream) {
  FGS_REQUIRE(n >= 0 && n < FGS_MAX_ELEMS, FGS_E_RANGE, "fgs_exclusive_scan_i64: n=%lld", (long long)n);
  FGS_REQUIRE(out && (n == 0 || in), FGS_E_INVALID, "fgs_exclusive_scan_i64: null pointer");
  hipLaunchKernelGGL(k_exclusive_scan_i64<false>, dim3(1), dim3(SCAN_THREADS), 0, fgs_s(stream), in, n, out, (int64_t)0,
                     (int *)nullptr, (int64_t *)nullptr);
  FGS_LAUNCH_OK("fgs_exclusive_scan_i64");
  return 0;
}

// fgs_exclusive_scan_i64 followed by fgs_count_guard(out, n + 1, capacity, flags, total) in one launch.
FGS_API int fgs_exclusive_scan_guard_i64(const int64_t *in, int64_t n, int64_t *out, int64_t capacity, int *flags,
                                         int64_t *total, fgs_stream_t stream) {
  FGS_REQUIRE(n >= 0 && n < FGS_MAX_ELEMS && capacity >= 0, FGS_E_RANGE, "fgs_exclusive_scan_guard_i64: n=%lld capacity=%lld",
              (long long)n, (long long)capacity);
  FGS_REQUIRE(out && flags && (n == 0 || in), FGS_E_INVALID, "fgs_exclusive_scan_guard_i64: null pointer");
  hipLaunchKernelGGL(k_exclusive_scan_i64<true>, dim3(1), dim3(SCAN_THREADS), 0, fgs_s(stream), in, n, out, capacity, flags,
                     total);
  FGS_LAUNCH_OK("fgs_exclusive_scan_guard_i64");
  return 0;
}
