// losses.hip -- the ray-dependent loss terms of one training iteration (model/nerf_training.py:308-327) and their
// gradients as one small kernel each way, instead of ~40 elementwise / reduction launches of the autograd graph.
// (training-loop host overhead.)
//
//   main      w_main   * mean((rgb_marched - target)^2)                                    :308
//   rgbper    w_rgbper * sum_i w_i * sum_c (raw_rgb_ic - target[ray_i]_c)^2 / N            :311-314  (weights detached)
//   entropy   w_ent    * H(clamp(alphainv_cum[N-1], 1e-6, 1-1e-6))                          :316-319  (`[..., -1]` on a 1-D
//                                                                                                      tensor: ONE ray, kept)
//   orient    w_ori    * sum_i w_i * min(0, normal_i . (-viewdir_i))^2                      :321-323, nerf.py:469-478
//   sigmoid   w_sig    * mean((sigmoid_rgb - target)^2)                                     :325-327
#include "fgs_common.h"

namespace {

struct LossArgs {
  int64_t N, M;
  const int64_t *m_dev;   // fgs_dyn_t.row_count: M is then the capacity
  const float *rgb_marched, *sigmoid_rgb, *target, *alphainv_cum;  // per ray
  const float *weights, *normal, *raw_rgb;                         // per survivor
  const int64_t *ray_id;
  const float *viewdirs;  // per ray
  float w_main, w_rgbper, w_ent, w_ori, w_sig;
};

__device__ __forceinline__ float block_sum_to(float v, float *dst) {
  __shared__ float part[FGS_BLOCK / FGS_WAVE];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < FGS_BLOCK / FGS_WAVE; ++w) s += part[w];
    atomicAdd(dst, s);
  }
  return v;
}

// ray terms of element i (of N*3)
__device__ __forceinline__ float loss_rays_term(const LossArgs &L, int64_t i) {
  float acc = 0.f;
  if (i < L.N * 3) {
    const float t = L.target[i];
    const float a = L.rgb_marched[i] - t, b = L.sigmoid_rgb[i] - t;
    const float inv = 1.f / (float)(L.N * 3);
    acc = L.w_main * (a * a) * inv + L.w_sig * (b * b) * inv;
  }
  if (i == 0 && L.w_ent > 0.f) {
    const float p = fminf(fmaxf(L.alphainv_cum[L.N - 1], 1e-6f), 1.f - 1e-6f);
    acc += L.w_ent * (-(p * logf(p) + (1.f - p) * logf(1.f - p)));
  }
  return acc;
}

// survivor terms of survivor m
__device__ __forceinline__ float loss_surv_term(const LossArgs &L, int64_t m) {
  float acc = 0.f;
  if (m < L.M) {
    const float w = L.weights[m];
    const int64_t r = L.ray_id[m];
    if (L.w_ori > 0.f) {
      const float d = -((L.normal[3 * m] * L.viewdirs[3 * r] + L.normal[3 * m + 1] * L.viewdirs[3 * r + 1]) +
                        L.normal[3 * m + 2] * L.viewdirs[3 * r + 2]);
      const float q = fminf(0.f, d);
      acc += L.w_ori * w * q * q;
    }
    if (L.w_rgbper > 0.f) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float e = L.raw_rgb[3 * m + c] - L.target[3 * r + c];
        s += e * e;
      }
      acc += L.w_rgbper * s * w / (float)L.N;
    }
  }
  return acc;
}

// One launch for both index ranges (thread i: element i of the N*3 ray values AND survivor i): each launch of a captured
// step costs ~5 us whatever it does.
// `partials` != NULL: every block leaves its sum in partials[block] and the LAST block to arrive (a counter word that it
// resets) adds them up in a fixed order and writes the scalar -- no zero fill in front of the launch (a node of its own in a
// captured step), and the value no longer depends on the order in which the blocks' atomics land.
__global__ __launch_bounds__(FGS_BLOCK) void k_loss_fwd(LossArgs L, float *loss, int with_surv, float *partials,
                                                        unsigned *counter) {
  L.M = fgs_rows(L.M, L.m_dev);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float acc = loss_rays_term(L, i);
  if (with_surv) acc += loss_surv_term(L, i);
  if (!partials) {
    block_sum_to(acc, loss);
    return;
  }
  __shared__ float part[FGS_BLOCK / FGS_WAVE];
  __shared__ int is_last;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < FGS_BLOCK / FGS_WAVE; ++w) s += part[w];
    __hip_atomic_store(partials + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    is_last = atomicAdd(counter, 1u) == gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  float v = 0.f;
  for (unsigned b = threadIdx.x; b < gridDim.x; b += FGS_BLOCK)
    v += __hip_atomic_load(partials + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < FGS_BLOCK / FGS_WAVE; ++w) s += part[w];
    loss[0] = s;
    *counter = 0u;                       // left as found: the next launch starts from zero again
  }
}

__device__ __forceinline__ void loss_rays_bwd(const LossArgs &L, int64_t i, float go, float *__restrict__ g_rgb_marched,
                                              float *__restrict__ g_sigmoid_rgb, float *__restrict__ g_last) {
  if (i < L.N * 3) {
    const float t = L.target[i];
    const float inv2 = 2.f / (float)(L.N * 3);
    g_rgb_marched[i] = go * L.w_main * inv2 * (L.rgb_marched[i] - t);
    g_sigmoid_rgb[i] = go * L.w_sig * inv2 * (L.sigmoid_rgb[i] - t);
  }
  if (i < L.N) {
    float g = 0.f;
    if (i == L.N - 1 && L.w_ent > 0.f) {
      const float raw = L.alphainv_cum[i];
      if (raw >= 1e-6f && raw <= 1.f - 1e-6f) g = go * L.w_ent * (logf(1.f - raw) - logf(raw));
    }
    g_last[i] = g;
  }
}

__device__ __forceinline__ void loss_surv_bwd(const LossArgs &L, int64_t m, float go, float *__restrict__ g_normal,
                                              float *__restrict__ g_raw_rgb) {
  if (m >= L.M) return;
  const float w = L.weights[m];
  const int64_t r = L.ray_id[m];
  const float v[3] = {L.viewdirs[3 * r], L.viewdirs[3 * r + 1], L.viewdirs[3 * r + 2]};
  float gn[3] = {0.f, 0.f, 0.f};
  if (L.w_ori > 0.f) {
    const float d = -((L.normal[3 * m] * v[0] + L.normal[3 * m + 1] * v[1]) + L.normal[3 * m + 2] * v[2]);
    if (d < 0.f) {
      const float k = go * L.w_ori * w * 2.f * d;
#pragma unroll
      for (int c = 0; c < 3; ++c) gn[c] = k * (-v[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) g_normal[3 * m + c] = gn[c];
  if (g_raw_rgb) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      g_raw_rgb[3 * m + c] = (L.w_rgbper > 0.f)
                                 ? go * L.w_rgbper * w * 2.f * (L.raw_rgb[3 * m + c] - L.target[3 * r + c]) / (float)L.N
                                 : 0.f;
  }
}

__global__ __launch_bounds__(FGS_BLOCK) void k_loss_bwd(LossArgs L, const float *__restrict__ grad_out,
                                                        float *__restrict__ g_rgb_marched, float *__restrict__ g_sigmoid_rgb,
                                                        float *__restrict__ g_last, float *__restrict__ g_normal,
                                                        float *__restrict__ g_raw_rgb) {
  L.M = fgs_rows(L.M, L.m_dev);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float go = grad_out[0];
  loss_rays_bwd(L, i, go, g_rgb_marched, g_sigmoid_rgb, g_last);
  if (g_normal) loss_surv_bwd(L, i, go, g_normal, g_raw_rgb);
}

int fill(LossArgs *L, int64_t N, int64_t M, const float *rgb_marched, const float *sigmoid_rgb, const float *target,
         const float *alphainv_cum, const float *weights, const float *normal, const float *raw_rgb,
         const int64_t *ray_id, const float *viewdirs, const float *w5, const fgs_dyn_t *dyn) {
  if (N <= 0 || N >= ((int64_t)1 << 31) || M < 0 || M >= ((int64_t)1 << 31))
    return fgs_set_error(FGS_E_RANGE, "fine loss: N=%lld M=%lld", (long long)N, (long long)M);
  if (!(rgb_marched && sigmoid_rgb && target && alphainv_cum && viewdirs && w5) ||
      (M > 0 && !(weights && normal && raw_rgb && ray_id)))
    return fgs_set_error(FGS_E_INVALID, "fine loss: null pointer");
  L->N = N; L->M = M; L->m_dev = fgs_dyn_rows(dyn); L->rgb_marched = rgb_marched; L->sigmoid_rgb = sigmoid_rgb; L->target = target;
  L->alphainv_cum = alphainv_cum; L->weights = weights; L->normal = normal; L->raw_rgb = raw_rgb; L->ray_id = ray_id;
  L->viewdirs = viewdirs;
  L->w_main = w5[0]; L->w_rgbper = w5[1]; L->w_ent = w5[2]; L->w_ori = w5[3]; L->w_sig = w5[4];
  return 0;
}

}  // namespace

// weights5_host = {weight_main, weight_rgbper, weight_entropy_last, weight_orientation, sigmoid_rgb_loss}.
// loss_out: device float.  scratch (optional): scratch_floats >= 1 + ceil(max(3 N, M) / 256) floats whose FIRST word is zero when
// first handed in (the kernel leaves it zero): per-block sums + a fixed-order final sum by the last block -- a deterministic
// scalar, one launch.  Without it (or too small): loss_out is zeroed by a stream-ordered memset and accumulated with atomics.
FGS_API int fgs_fine_loss_fwd(int64_t N, int64_t M, const float *rgb_marched, const float *sigmoid_rgb, const float *target,
                              const float *alphainv_cum, const float *weights, const float *normal, const float *raw_rgb,
                              const int64_t *ray_id, const float *viewdirs, const float *weights5_host, float *loss_out,
                              float *scratch, int64_t scratch_floats, const fgs_dyn_t *dyn, fgs_stream_t stream) {
  LossArgs L;
  if (int e = fill(&L, N, M, rgb_marched, sigmoid_rgb, target, alphainv_cum, weights, normal, raw_rgb, ray_id, viewdirs,
                   weights5_host, dyn)) return e;
  FGS_REQUIRE(loss_out, FGS_E_INVALID, "fgs_fine_loss_fwd: null loss_out");
  hipStream_t st = fgs_s(stream);
  const int with_surv = (M > 0 && (L.w_ori > 0.f || L.w_rgbper > 0.f)) ? 1 : 0;
  const int64_t n_thr = (with_surv && M > N * 3) ? M : N * 3;
  const unsigned blocks = fgs_blocks(n_thr);
  float *partials = nullptr;
  if (scratch && scratch_floats >= (int64_t)blocks + 1) partials = scratch + 1;
  if (!partials) {
    hipError_t he = hipMemsetAsync(loss_out, 0, sizeof(float), st);
    if (he != hipSuccess) return fgs_set_error((int)he, "fgs_fine_loss_fwd: %s", hipGetErrorString(he));
  }
  hipLaunchKernelGGL(k_loss_fwd, dim3(blocks), dim3(FGS_BLOCK), 0, st, L, loss_out, with_surv, partials,
                     reinterpret_cast<unsigned *>(scratch));
  FGS_LAUNCH_OK("fgs_fine_loss_fwd");
  return 0;
}

// grad_out: device float (d total / d loss).  g_raw_rgb may be NULL when weight_rgbper == 0.
FGS_API int fgs_fine_loss_bwd(int64_t N, int64_t M, const float *rgb_marched, const float *sigmoid_rgb, const float *target,
                              const float *alphainv_cum, const float *weights, const float *normal, const float *raw_rgb,
                              const int64_t *ray_id, const float *viewdirs, const float *weights5_host,
                              const float *grad_out, float *g_rgb_marched, float *g_sigmoid_rgb, float *g_last,
                              float *g_normal, float *g_raw_rgb, const fgs_dyn_t *dyn, fgs_stream_t stream) {
  LossArgs L;
  if (int e = fill(&L, N, M, rgb_marched, sigmoid_rgb, target, alphainv_cum, weights, normal, raw_rgb, ray_id, viewdirs,
                   weights5_host, dyn)) return e;
  FGS_REQUIRE(grad_out && g_rgb_marched && g_sigmoid_rgb && g_last && (M == 0 || g_normal), FGS_E_INVALID,
              "fgs_fine_loss_bwd: null pointer");
  hipStream_t st = fgs_s(stream);
  const int64_t n_thr = M > N * 3 ? M : N * 3;
  hipLaunchKernelGGL(k_loss_bwd, dim3(fgs_blocks(n_thr)), dim3(FGS_BLOCK), 0, st, L, grad_out, g_rgb_marched, g_sigmoid_rgb,
                     g_last, M > 0 ? g_normal : (float *)nullptr, g_raw_rgb);
  FGS_LAUNCH_OK("fgs_fine_loss_bwd");
  return 0;
}
