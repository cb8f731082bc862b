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
    // (agent-scope store, waited for, then the arrival: see k_render_loss for why there is no agent-scope fence)
    __hip_atomic_store(partials + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    is_last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!is_last) return;
  float v = fgs_partials_sum<float>(partials, threadIdx.x, gridDim.x, FGS_BLOCK);
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

// ------------------------------------------------------------------------------------------------ fused compositing + losses
// One launch for what a training step runs between its two MLP chains besides the 256 -> 3 head: per-ray compositing
// (model/nerf.py:888-920; k_composite_fwd), the loss terms above (k_loss_fwd), their gradients (k_loss_bwd) and the compositing
// backward (k_composite_bwd) -- five launches of a captured step, each of which costs ~5 us whatever it does.  One wave per ray,
// two passes over the ray's survivors: (1) the weighted sums, from which the ray's pixel, its loss terms and d loss / d pixel
// follow; (2) per survivor d loss / d (pre-sigmoid head output), d loss / d weight, the orientation term and its gradient w.r.t.
// the normal.  The loss gradients assume d total / d loss = *seed (NULL: 1); the loss scalar is summed per wave, per block and
// by the last block to arrive in a fixed order (as k_loss_fwd does): bit-reproducible, another order than the separate kernels'.
struct RenderLossArgs {
  int64_t N, M;
  const int64_t *m_dev;
  const int64_t *surv_off;                      // [N + 1]
  const float *weights, *rgb, *normal;          // per survivor; rgb = sigmoid(head output)
  const int64_t *step_id;                       // per survivor (depth) or null
  const float *viewdirs, *target, *alphainv_last;      // per ray
  const float *seed;
  float bg, dist;
  float w_main, w_rgbper, w_ent, w_ori, w_sig;
  float *rgb_marched, *sigmoid_rgb, *pre_rgb, *pre_sig, *normal_marched, *depth;      // per ray
  float *d_out, *d_w, *g_normal;                // per survivor
  float *g_last, *g_rm;                         // per ray
};

__device__ __forceinline__ float rl_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(FGS_BLOCK) void k_render_loss(RenderLossArgs A, float *loss, float *partials, unsigned *counter) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * (FGS_BLOCK / FGS_WAVE) + wv;
  float ray_loss = 0.f;
  if (ray < A.N) {
    const int64_t s0 = fgs_uniform(A.surv_off[ray]), s1 = fgs_uniform(A.surv_off[ray + 1]);
    // ---- pass 1: compositing sums (k_composite_fwd)
    float acc[3] = {0.f, 0.f, 0.f}, sig[3] = {0.f, 0.f, 0.f}, nrm[3] = {0.f, 0.f, 0.f}, wsum = 0.f, dep = 0.f;
    for (int64_t i = s0 + lane; i < s1; i += FGS_WAVE) {
      const float w = A.weights[i];
      wsum += w;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float v = A.rgb[3 * i + c];
        acc[c] += w * v;
        sig[c] += w * (1.f / (1.f + expf(-v)));
        if (A.normal_marched) nrm[c] += w * A.normal[3 * i + c];
      }
      if (A.depth) dep += (w * (float)A.step_id[i]) * A.dist;
    }
    wsum = rl_wave_sum(wsum);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      acc[c] = rl_wave_sum(acc[c]);
      sig[c] = rl_wave_sum(sig[c]);
      if (A.normal_marched) nrm[c] = rl_wave_sum(nrm[c]);
    }
    if (A.depth) dep = rl_wave_sum(dep);
    // ---- the ray: pixel, loss terms, d loss / d pixel (k_loss_fwd / k_loss_bwd, ray part)
    const float go = A.seed ? *A.seed : 1.f;
    const float bgterm = (1.f - wsum) * A.bg;
    const float inv = 1.f / (float)(A.N * 3), inv2 = 2.f / (float)(A.N * 3);
    float g1[3], g2[3], tg[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float a = acc[c] + bgterm, b = sig[c] + bgterm;
      const float rm = fminf(fmaxf(a, 0.f), 1.f), sr = fminf(fmaxf(b, 0.f), 1.f);
      tg[c] = A.target[3 * ray + c];
      const float da = rm - tg[c], db = sr - tg[c];
      ray_loss += A.w_main * (da * da) * inv + A.w_sig * (db * db) * inv;
      const float grm = go * A.w_main * inv2 * da, gsr = go * A.w_sig * inv2 * db;
      g1[c] = (a >= 0.f && a <= 1.f) ? grm : 0.f;        // clamp(0, 1) passes the gradient on the closed interval
      g2[c] = (b >= 0.f && b <= 1.f) ? gsr : 0.f;
      if (lane == 0) {
        A.pre_rgb[3 * ray + c] = a; A.pre_sig[3 * ray + c] = b;
        A.rgb_marched[3 * ray + c] = rm; A.sigmoid_rgb[3 * ray + c] = sr;
        if (A.normal_marched) A.normal_marched[3 * ray + c] = nrm[c];
        A.g_rm[3 * ray + c] = grm;
      }
    }
    if (lane == 0) {
      if (A.depth) A.depth[ray] = dep;
      float gl = 0.f;
      if (ray == A.N - 1 && A.w_ent > 0.f) {           // `alphainv_cum[..., -1]` on a 1-D tensor: ONE ray (nerf_training.py:316-319)
        const float raw = A.alphainv_last[ray];
        const float p = fminf(fmaxf(raw, 1e-6f), 1.f - 1e-6f);
        ray_loss += A.w_ent * (-(p * logf(p) + (1.f - p) * logf(1.f - p)));
        if (raw >= 1e-6f && raw <= 1.f - 1e-6f) gl = go * A.w_ent * (logf(1.f - raw) - logf(raw));
      }
      A.g_last[ray] = gl;
    } else {
      ray_loss = 0.f;                                    // (the ray's terms are counted once, by lane 0)
    }
    // ---- pass 2: per survivor (k_loss_fwd / k_loss_bwd survivor part, k_composite_bwd)
    const float vd[3] = {A.viewdirs[3 * ray], A.viewdirs[3 * ray + 1], A.viewdirs[3 * ray + 2]};
    for (int64_t i = s0 + lane; i < s1; i += FGS_WAVE) {
      const float w = A.weights[i];
      float gn[3] = {0.f, 0.f, 0.f};
      if (A.w_ori > 0.f) {
        const float d = -((A.normal[3 * i] * vd[0] + A.normal[3 * i + 1] * vd[1]) + A.normal[3 * i + 2] * vd[2]);
        const float q = fminf(0.f, d);
        ray_loss += A.w_ori * w * q * q;
        if (d < 0.f) {
          const float k = go * A.w_ori * w * 2.f * d;
#pragma unroll
          for (int c = 0; c < 3; ++c) gn[c] = k * (-vd[c]);
        }
      }
      float dw = 0.f, se = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        A.g_normal[3 * i + c] = gn[c];
        const float v = A.rgb[3 * i + c];
        const float sg = 1.f / (1.f + expf(-v));
        dw += v * g1[c] + sg * g2[c] - A.bg * (g1[c] + g2[c]);
        float d_rgb = w * g1[c] + (w * g2[c]) * (sg * (1.f - sg));
        if (A.w_rgbper > 0.f) {
          const float e = v - tg[c];
          se += e * e;
          d_rgb += go * A.w_rgbper * w * 2.f * e / (float)A.N;
        }
        A.d_out[3 * i + c] = d_rgb * (v * (1.f - v));     // rgb = sigmoid(out)
      }
      if (A.w_rgbper > 0.f) ray_loss += A.w_rgbper * se * w / (float)A.N;
      A.d_w[i] = dw;
    }
    ray_loss = rl_wave_sum(ray_loss);
  }
  // ---- the scalar: wave -> block -> the last block to arrive sums the blocks in a fixed order
  __shared__ float part[FGS_BLOCK / FGS_WAVE];
  __shared__ int is_last;
  if (lane == 0) part[wv] = ray_loss;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < FGS_BLOCK / FGS_WAVE; ++w) s += part[w];
    // The partial leaves as an agent-scope store (written through this XCD's L2) and is waited for before the arrival is
    // counted: no agent-scope release fence here -- that one writes back EVERY dirty line of the L2 (the activations the
    // kernels around this one leave there), once per workgroup: 1024 of them made this a 42 us launch.
    __hip_atomic_store(partials + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    is_last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!is_last) return;
  float v = fgs_partials_sum<float>(partials, threadIdx.x, gridDim.x, FGS_BLOCK);      // (agent scope: not from this XCD's L2)
  v = rl_wave_sum(v);
  __syncthreads();
  if (lane == 0) part[wv] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < FGS_BLOCK / FGS_WAVE; ++w) s += part[w];
    loss[0] = s;
    *counter = 0u;
  }
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

// The fused form of fgs_composite_fwd + fgs_fine_loss_fwd + fgs_fine_loss_bwd + fgs_composite_bwd (one launch; see k_render_loss).
// rgb [M,3] = sigmoid(head output); surv_off [N + 1] the per-ray survivor offsets; seed_dev: device float d total / d loss (NULL: 1).
// Outputs: the per-ray render (rgb_marched, sigmoid_rgb, pre_rgb, pre_sig; normal_marched / depth optional), loss_out, and the
// gradients the backward pass starts from: d_out [M,3] (w.r.t. the head's pre-sigmoid output), d_w [M], g_normal [M,3],
// g_last [N], g_rgb_marched [N,3] (d loss / d rgb_marched before the clamp gate: what fgs_fine_loss_bwd returns).
// scratch: >= 1 + ceil(N / 4) floats whose first word is zero when first handed in (left zero).
FGS_API int fgs_fine_render_loss(int64_t N, int64_t M, const int64_t *surv_off, const float *weights, const float *rgb,
                                 const float *normal, const int64_t *step_id, float bg, float dist, const float *viewdirs,
                                 const float *target, const float *alphainv_last, const float *weights5_host, const float *seed_dev,
                                 float *rgb_marched, float *sigmoid_rgb, float *pre_rgb, float *pre_sig, float *normal_marched,
                                 float *depth, float *loss_out, float *scratch, int64_t scratch_floats, float *d_out, float *d_w,
                                 float *g_normal, float *g_last, float *g_rgb_marched, const fgs_dyn_t *dyn, fgs_stream_t stream) {
  FGS_REQUIRE(N > 0 && N < ((int64_t)1 << 31) && M >= 0 && M < ((int64_t)1 << 31), FGS_E_RANGE, "fgs_fine_render_loss: N=%lld M=%lld",
              (long long)N, (long long)M);
  FGS_REQUIRE(surv_off && viewdirs && target && alphainv_last && weights5_host && rgb_marched && sigmoid_rgb && pre_rgb && pre_sig &&
                  loss_out && scratch && g_last && g_rgb_marched && (M == 0 || (weights && rgb && normal && d_out && d_w && g_normal)) &&
                  (!depth || step_id), FGS_E_INVALID, "fgs_fine_render_loss: null pointer");
  const unsigned blocks = fgs_blocks(N * FGS_WAVE);
  FGS_REQUIRE(scratch_floats >= (int64_t)blocks + 1, FGS_E_INVALID, "fgs_fine_render_loss: scratch %lld floats, need %u",
              (long long)scratch_floats, blocks + 1);
  RenderLossArgs A;
  A.N = N; A.M = M; A.m_dev = fgs_dyn_rows(dyn); A.surv_off = surv_off; A.weights = weights; A.rgb = rgb; A.normal = normal;
  A.step_id = step_id; A.viewdirs = viewdirs; A.target = target; A.alphainv_last = alphainv_last; A.seed = seed_dev; A.bg = bg;
  A.dist = dist;
  A.w_main = weights5_host[0]; A.w_rgbper = weights5_host[1]; A.w_ent = weights5_host[2]; A.w_ori = weights5_host[3];
  A.w_sig = weights5_host[4];
  A.rgb_marched = rgb_marched; A.sigmoid_rgb = sigmoid_rgb; A.pre_rgb = pre_rgb; A.pre_sig = pre_sig; A.normal_marched = normal_marched;
  A.depth = depth; A.d_out = d_out; A.d_w = d_w; A.g_normal = g_normal; A.g_last = g_last; A.g_rm = g_rgb_marched;
  hipLaunchKernelGGL(k_render_loss, dim3(blocks), dim3(FGS_BLOCK), 0, fgs_s(stream), A, loss_out, scratch + 1,
                     reinterpret_cast<unsigned *>(scratch));
  FGS_LAUNCH_OK("fgs_fine_render_loss");
  return 0;
}
