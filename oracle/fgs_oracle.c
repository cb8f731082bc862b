/*
 * fgs_oracle.c -- CPU restatement of the reference's native kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under fgs-nerf_amd/ (the product) may link,
 * import or call this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do, and only as the checker.
 *
 * PARITY STATUS: "parity unpinned" for everything in this file.  The reference
 * ships no tests, fixtures or golden vectors and its
 * model/cuda sources need nvcc + a torch-2.1 header set, neither present here,
 * so they are unbuildable in this image.  Each function
 * below therefore follows the reference .cu text statement by statement and
 * cites it; the only degrees of freedom are the places where nvcc contracts
 * a*b+c into an FMA.  Those are written out with explicit fmaf() here, the HIP
 * kernels use the same explicit fmaf() calls and both are compiled with
 * -ffp-contract=off, so the integer outputs (N_steps, ray_id, step_id, masks,
 * i_start/i_end) are bit-exact between oracle and HIP by construction.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------- sampling */

static inline float orc_rnorm(const float *d) {
  /* render_utils_kernel.cu:48-51 / :68-71 -- sqrt(dx*dx + dy*dy + dz*dz);
   * contraction pinned as fma(dz,dz, fma(dy,dy, dx*dx)). */
  return sqrtf(fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0])));
}

/* render_utils_kernel.cu:11-35 infer_t_minmax_cuda_kernel */
ORC_API void orc_infer_t_minmax(const float *rays_o, const float *rays_d,
                                const float *xyz_min, const float *xyz_max,
                                float near, float far, int64_t n_rays,
                                float *t_min, float *t_max) {
  for (int64_t r = 0; r < n_rays; ++r) {
    const float *o = rays_o + 3 * r, *d = rays_d + 3 * r;
    float v[3], a[3], b[3];
    for (int c = 0; c < 3; ++c) {
      v[c] = (d[c] == 0.f) ? (float)1e-6 : d[c]; /* double literal narrowed, :23-25 */
      a[c] = (xyz_max[c] - o[c]) / v[c];
      b[c] = (xyz_min[c] - o[c]) / v[c];
    }
    float lo = fmaxf(fmaxf(fminf(a[0], b[0]), fminf(a[1], b[1])), fminf(a[2], b[2]));
    float hi = fminf(fminf(fmaxf(a[0], b[0]), fmaxf(a[1], b[1])), fmaxf(a[2], b[2]));
    t_min[r] = fmaxf(fminf(lo, far), near); /* :32 */
    t_max[r] = fmaxf(fminf(hi, far), near); /* :33 */
  }
}

/* render_utils_kernel.cu:37-55 infer_n_samples_cuda_kernel */
ORC_API void orc_infer_n_samples(const float *rays_d, const float *t_min,
                                 const float *t_max, float stepdist,
                                 int64_t n_rays, int64_t *n_samples) {
  for (int64_t r = 0; r < n_rays; ++r) {
    const float rn = orc_rnorm(rays_d + 3 * r);
    const float c = ceilf((t_max[r] - t_min[r]) * rn / stepdist);
    const double m = fmax((double)c, 1.); /* max(float, double literal) :53 */
    n_samples[r] = (int64_t)m;
  }
}

/* render_utils_kernel.cu:57-79 infer_ray_start_dir_cuda_kernel */
ORC_API void orc_infer_ray_start_dir(const float *rays_o, const float *rays_d,
                                     const float *t_min, int64_t n_rays,
                                     float *rays_start, float *rays_dir) {
  for (int64_t r = 0; r < n_rays; ++r) {
    const float *o = rays_o + 3 * r, *d = rays_d + 3 * r;
    const float rn = orc_rnorm(d);
    for (int c = 0; c < 3; ++c) {
      rays_start[3 * r + c] = fmaf(d[c], t_min[r], o[c]); /* o + d*t :72-74 */
      rays_dir[3 * r + c] = d[c] / rn;                    /* :75-77 */
    }
  }
}

/* Phase 1 of render_utils_kernel.cu:196-242: t_min/t_max, N_steps and their
 * total (the .item<int>() at :212).  Returns the total sample count. */
ORC_API int64_t orc_sample_count(const float *rays_o, const float *rays_d,
                                 const float *xyz_min, const float *xyz_max,
                                 float near, float far, float stepdist,
                                 int64_t n_rays, int64_t *n_steps,
                                 float *t_min, float *t_max) {
  orc_infer_t_minmax(rays_o, rays_d, xyz_min, xyz_max, near, far, n_rays, t_min, t_max);
  orc_infer_n_samples(rays_d, t_min, t_max, stepdist, n_rays, n_steps);
  int64_t tot = 0;
  for (int64_t r = 0; r < n_rays; ++r) tot += n_steps[r];
  return tot;
}

/* Phase 2 of render_utils_kernel.cu:196-242: ray_id / step_id (:144-164 via the
 * two cumsums) and the points + out-of-bbox mask (:166-194). */
ORC_API void orc_sample_emit(const float *rays_o, const float *rays_d,
                             const float *xyz_min, const float *xyz_max,
                             float stepdist, int64_t n_rays,
                             const int64_t *n_steps, const float *t_min,
                             float *rays_pts, uint8_t *mask_outbbox,
                             int64_t *ray_id, int64_t *step_id) {
  int64_t idx = 0;
  for (int64_t r = 0; r < n_rays; ++r) {
    float start[3], dir[3];
    orc_infer_ray_start_dir(rays_o + 3 * r, rays_d + 3 * r, t_min + r, 1, start, dir);
    for (int64_t s = 0; s < n_steps[r]; ++s, ++idx) {
      ray_id[idx] = r;
      step_id[idx] = s;
      const float dist = stepdist * (float)(int)s; /* stepdist * i_step :184 */
      float p[3];
      for (int c = 0; c < 3; ++c) p[c] = fmaf(dir[c], dist, start[c]); /* :185-187 */
      rays_pts[3 * idx + 0] = p[0];
      rays_pts[3 * idx + 1] = p[1];
      rays_pts[3 * idx + 2] = p[2];
      mask_outbbox[idx] = (uint8_t)((xyz_min[0] > p[0]) | (xyz_min[1] > p[1]) | (xyz_min[2] > p[2]) |
                                    (xyz_max[0] < p[0]) | (xyz_max[1] < p[1]) | (xyz_max[2] < p[2]));
    }
  }
}

/* render_utils_kernel.cu:244-293 sample_ndc_pts_on_rays_cuda_kernel */
ORC_API void orc_sample_ndc_pts(const float *rays_o, const float *rays_d,
                                const float *xyz_min, const float *xyz_max,
                                int64_t n_samples, int64_t n_rays,
                                float *rays_pts, uint8_t *mask_outbbox) {
  for (int64_t idx = 0; idx < n_rays * n_samples; ++idx) {
    const int64_t r = idx / n_samples, s = idx % n_samples;
    const float dist = ((float)(int)s) / (float)(int)(n_samples - 1); /* :260 */
    float p[3];
    for (int c = 0; c < 3; ++c) p[c] = fmaf(rays_d[3 * r + c], dist, rays_o[3 * r + c]);
    for (int c = 0; c < 3; ++c) rays_pts[3 * idx + c] = p[c];
    mask_outbbox[idx] = (uint8_t)((xyz_min[0] > p[0]) | (xyz_min[1] > p[1]) | (xyz_min[2] > p[2]) |
                                  (xyz_max[0] < p[0]) | (xyz_max[1] < p[1]) | (xyz_max[2] < p[2]));
  }
}

/* render_utils_kernel.cu:300-340 sample_bg_pts_on_rays_cuda_kernel.
 * The reference mixes double literals (1., 1. - bg_preserve) into float math;
 * the promotions are kept. */
ORC_API void orc_sample_bg_pts(const float *rays_o, const float *rays_d,
                               const float *t_max, float bg_preserve,
                               int64_t n_samples, int64_t n_rays, float *rays_pts) {
  for (int64_t idx = 0; idx < n_rays * n_samples; ++idx) {
    const int64_t r = idx / n_samples, s = idx % n_samples;
    const float t_inner = t_max[r];
    const float frac = ((float)(int)s) / (float)(int)n_samples;
    const float ori_t_outer = (float)((double)t_inner - 1. + 1. / (1. - (double)frac)); /* :325 */
    float q[3];
    for (int c = 0; c < 3; ++c) q[c] = fmaf(rays_d[3 * r + c], ori_t_outer, rays_o[3 * r + c]);
    const float t_outer = sqrtf(fmaf(q[2], q[2], fmaf(q[1], q[1], q[0] * q[0]))); /* norm3 :296-298 */
    const float m = fmaxf(fabsf(q[0]), fmaxf(fabsf(q[1]), fabsf(q[2])));
    const float R = t_outer / m;
    const float o2i = (float)((double)(R * R / (t_outer * t_outer)) * (1. - (double)bg_preserve) +
                              (double)(R / t_outer * bg_preserve)); /* :332 */
    for (int c = 0; c < 3; ++c) rays_pts[3 * idx + c] = q[c] * o2i;
  }
}

/* ------------------------------------------------------------- mask lookup */

/* render_utils_kernel.cu:373-392 maskcache_lookup_cuda_kernel (out is
 * zero-initialised by the host wrapper at :405). */
ORC_API void orc_maskcache_lookup(const uint8_t *world, const float *xyz,
                                  const float *scale, const float *shift,
                                  int sz_i, int sz_j, int sz_k, int64_t n_pts,
                                  uint8_t *out) {
  for (int64_t p = 0; p < n_pts; ++p) {
    const int i = (int)roundf(fmaf(xyz[3 * p + 0], scale[0], shift[0]));
    const int j = (int)roundf(fmaf(xyz[3 * p + 1], scale[1], shift[1]));
    const int k = (int)roundf(fmaf(xyz[3 * p + 2], scale[2], shift[2]));
    out[p] = 0;
    if (0 <= i && i < sz_i && 0 <= j && j < sz_j && 0 <= k && k < sz_k)
      out[p] = world[(int64_t)i * sz_j * sz_k + (int64_t)j * sz_k + k];
  }
}

/* --------------------------------------------------------------- raw2alpha */

/* render_utils_kernel.cu:430-458 (uniform) and :445-458 (non-uniform interval) */
ORC_API void orc_raw2alpha(const float *density, float shift, float interval,
                           const float *interval_nonuni, int64_t n,
                           float *exp_d, float *alpha) {
  for (int64_t i = 0; i < n; ++i) {
    const float iv = interval_nonuni ? interval_nonuni[i] : interval;
    const float e = expf(density[i] + shift);
    exp_d[i] = e;
    alpha[i] = 1.f - powf(1.f + e, -iv);
  }
}

/* render_utils_kernel.cu:506-530: min(exp_d, 1e10) is a double min, the rest of
 * the product is carried in double and narrowed on store. */
ORC_API void orc_raw2alpha_bwd(const float *exp_d, const float *grad_back,
                               float interval, const float *interval_nonuni,
                               int64_t n, float *grad) {
  for (int64_t i = 0; i < n; ++i) {
    const float iv = interval_nonuni ? interval_nonuni[i] : interval;
    const double m = fmin((double)exp_d[i], 1e10);
    const float pw = powf(1.f + exp_d[i], -iv - 1.f);
    grad[i] = (float)(m * (double)pw * (double)iv * (double)grad_back[i]);
  }
}

/* ------------------------------------------------------------ alpha2weight */

/* render_utils_kernel.cu:607-617 + :633-635: segment boundaries of a sorted
 * ray_id list.  i_start / i_end are zero-initialised (:627-628). */
ORC_API void orc_segment_start_end(const int64_t *ray_id, int64_t n_pts,
                                   int64_t n_rays, int64_t *i_start, int64_t *i_end) {
  memset(i_start, 0, sizeof(int64_t) * (size_t)n_rays);
  memset(i_end, 0, sizeof(int64_t) * (size_t)n_rays);
  if (n_pts == 0) return;
  for (int64_t i = 1; i < n_pts; ++i)
    if (ray_id[i] != ray_id[i - 1]) {
      i_start[ray_id[i]] = i;
      i_end[ray_id[i - 1]] = i;
    }
  i_end[ray_id[n_pts - 1]] = n_pts;
}

/* render_utils_kernel.cu:576-605 alpha2weight_cuda_kernel; weight zero-init,
 * T one-init, alphainv_last one-init (:624-626).  `1. - alpha` and the `*=`
 * run in double and narrow to float every step; `T_cum<1e-3` compares in
 * double. */
ORC_API void orc_alpha2weight_fwd(const float *alpha, const int64_t *ray_id,
                                  int64_t n_pts, int64_t n_rays, float *weight,
                                  float *T, float *alphainv_last,
                                  int64_t *i_start, int64_t *i_end) {
  for (int64_t i = 0; i < n_pts; ++i) { weight[i] = 0.f; T[i] = 1.f; }
  for (int64_t r = 0; r < n_rays; ++r) alphainv_last[r] = 1.f;
  orc_segment_start_end(ray_id, n_pts, n_rays, i_start, i_end);
  if (n_pts == 0) return;
  for (int64_t r = 0; r < n_rays; ++r) {
    const int64_t i_s = i_start[r], i_e_max = i_end[r];
    float T_cum = 1.f;
    int64_t i;
    for (i = i_s; i < i_e_max; ++i) {
      T[i] = T_cum;
      weight[i] = T_cum * alpha[i];
      T_cum = (float)((double)T_cum * (1. - (double)alpha[i]));
      if ((double)T_cum < 1e-3) { i += 1; break; }
    }
    i_end[r] = i;
    alphainv_last[r] = T_cum;
  }
}

/* render_utils_kernel.cu:653-677 alpha2weight_backward_cuda_kernel; grad
 * zero-init (:684).  `1-alpha+1e-10` and the division are double; the
 * accumulation `back_cum += gw*w` is contracted to one fmaf. */
ORC_API void orc_alpha2weight_bwd(const float *alpha, const float *weight,
                                  const float *T, const float *alphainv_last,
                                  const int64_t *i_start, const int64_t *i_end,
                                  int64_t n_pts, int64_t n_rays,
                                  const float *grad_weights,
                                  const float *grad_last, float *grad) {
  for (int64_t i = 0; i < n_pts; ++i) grad[i] = 0.f;
  for (int64_t r = 0; r < n_rays; ++r) {
    float back_cum = grad_last[r] * alphainv_last[r];
    for (int64_t i = i_end[r] - 1; i >= i_start[r]; --i) {
      const double den = (double)(1.f - alpha[i]) + 1e-10;
      grad[i] = (float)((double)(grad_weights[i] * T[i]) - (double)back_cum / den);
      back_cum = fmaf(grad_weights[i], weight[i], back_cum);
    }
  }
}

/* --------------------------------------------------------- total variation */

static inline float orc_clamp1(float v) { return fminf(fmaxf(v, -1.f), 1.f); }

/* total_variation_kernel.cu:13-35 (mask==NULL) and :38-66 (mask given).
 * Weights are pre-divided by 6 by the host wrappers (:76-78, :112-114).
 * The unmasked kernel uses wz on the k axis, wy on j and wz on i (wx unused);
 * the masked one uses wx on k, wy on j, wz on i.  Kept as is. */
ORC_API void orc_tv_add_grad(const float *param, float *grad, const float *mask,
                             float wx, float wy, float wz, int dense_mode,
                             int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t N) {
  wx /= 6; wy /= 6; wz /= 6;
  /* grad is updated in place but only read at `index`, so a single pass is exact */
  for (int64_t index = 0; index < N; ++index) {
    if (!(dense_mode || grad[index] != 0.f)) continue;
    const int64_t k = index % sz_k, j = index / sz_k % sz_j, i = index / sz_k / sz_j % sz_i;
    const int64_t sj = sz_k, si = sz_k * sz_j;
    float g = 0.f;
    if (!mask) {
      g += (k == 0        ? 0.f : wz * orc_clamp1(param[index] - param[index - 1]));
      g += (k == sz_k - 1 ? 0.f : wz * orc_clamp1(param[index] - param[index + 1]));
      g += (j == 0        ? 0.f : wy * orc_clamp1(param[index] - param[index - sj]));
      g += (j == sz_j - 1 ? 0.f : wy * orc_clamp1(param[index] - param[index + sj]));
      g += (i == 0        ? 0.f : wz * orc_clamp1(param[index] - param[index - si]));
      g += (i == sz_i - 1 ? 0.f : wz * orc_clamp1(param[index] - param[index + si]));
    } else {
      const float m0 = mask[index];
      g += (k == 0        ? 0.f : wx * orc_clamp1(param[index] - param[index - 1]) * m0 * mask[index - 1]);
      g += (k == sz_k - 1 ? 0.f : wx * orc_clamp1(param[index] - param[index + 1]) * m0 * mask[index + 1]);
      g += (j == 0        ? 0.f : wy * orc_clamp1(param[index] - param[index - sj]) * m0 * mask[index - sj]);
      g += (j == sz_j - 1 ? 0.f : wy * orc_clamp1(param[index] - param[index + sj]) * m0 * mask[index + sj]);
      g += (i == 0        ? 0.f : wz * orc_clamp1(param[index] - param[index - si]) * m0 * mask[index - si]);
      g += (i == sz_i - 1 ? 0.f : wz * orc_clamp1(param[index] - param[index + si]) * m0 * mask[index + si]);
    }
    grad[index] += g;
  }
}

/* -------------------------------------------------------------------- Adam */

/* adam_upd_kernel.cu:72 (host): all-float step size. */
ORC_API float orc_adam_step_size(int step, float beta1, float beta2, float lr) {
  return lr * sqrtf(1.f - powf(beta2, (float)step)) / (1.f - powf(beta1, (float)step));
}

/* adam_upd_kernel.cu:8-23 (mode 0), :25-40 masked (mode 1), :42-58 per-voxel
 * lr (mode 2).  Contraction pinned: first product fused into the add. */
ORC_API void orc_adam_upd(float *param, const float *grad, float *exp_avg,
                          float *exp_avg_sq, const float *perlr, int64_t N,
                          int step, float beta1, float beta2, float lr, float eps,
                          int mode) {
  const float step_size = orc_adam_step_size(step, beta1, beta2, lr);
  for (int64_t i = 0; i < N; ++i) {
    const float g = grad[i];
    if (mode == 1 && g == 0.f) continue;
    const float m = fmaf(beta1, exp_avg[i], (1.f - beta1) * g);
    const float v = fmaf(beta2, exp_avg_sq[i], (1.f - beta2) * g * g);
    exp_avg[i] = m;
    exp_avg_sq[i] = v;
    const float num = (mode == 2) ? step_size * perlr[i] * m : step_size * m;
    param[i] -= num / (sqrtf(v) + eps);
  }
}
