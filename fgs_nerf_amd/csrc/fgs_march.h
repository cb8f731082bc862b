// fgs_march.h -- per-ray helpers shared by the fine- and coarse-stage march kernels (march.hip, march_coarse.hip).
#pragma once

#include "fgs_taps.h"

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

struct RaySetup {
  float start[3], dir[3];
  int64_t n_steps;
};

// ray/AABB entry + sample count + start/dir: render_utils_kernel.cu:11-79 (same statements as csrc/sampling.hip)
__device__ __forceinline__ RaySetup ray_setup(const float *o, const float *d, const SceneGeom &g, float near, float far,
                                              float stepdist) {
  float a[3], b[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = (d[c] == 0.f) ? (float)1e-6 : d[c];
    a[c] = (g.hi[c] - o[c]) / v;
    b[c] = (g.lo[c] - o[c]) / v;
  }
  const float en = fmaxf(fmaxf(fminf(a[0], b[0]), fminf(a[1], b[1])), fminf(a[2], b[2]));
  const float ex = fminf(fminf(fmaxf(a[0], b[0]), fmaxf(a[1], b[1])), fmaxf(a[2], b[2]));
  const float t_min = fmaxf(fminf(en, far), near), t_max = fmaxf(fminf(ex, far), near);
  const float rn = fgs_rnorm3(d[0], d[1], d[2]);
  RaySetup r;
  r.n_steps = (int64_t)fmax((double)ceilf((t_max - t_min) * rn / stepdist), 1.);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    r.start[c] = fmaf(d[c], t_min, o[c]);
    r.dir[c] = d[c] / rn;
  }
  return r;
}

// model/nerf.py:525-543 (cos_anneal_ratio = 1, use_mid)
__device__ __forceinline__ float neus_alpha(float sdf, float gx, float gy, float gz, float vx, float vy, float vz,
                                            float dist, float inv_s) {
  const float true_cos = (vx * gx + vy * gy) + vz * gz;
  const float iter_cos = -(fmaxf(-true_cos * 0.5f + 0.5f, 0.f) * 0.0f + fmaxf(-true_cos, 0.f) * 1.0f);
  const float half = iter_cos * dist * 0.5f;
  const float prev_cdf = sigmoidf_((sdf - half) * inv_s);
  const float next_cdf = sigmoidf_((sdf + half) * inv_s);
  const float a = ((prev_cdf - next_cdf) + 1e-5f) / (prev_cdf + 1e-5f);
  return fminf(fmaxf(a, 0.f), 1.f);
}

// Autograd of neus_alpha: g_alpha -> (d sdf, d gradient xyz, d inv_s)
struct AlphaGrad {
  float d_sdf, dgx, dgy, dgz, d_inv_s;
};

__device__ __forceinline__ AlphaGrad neus_alpha_bwd(float g_alpha, float sdf, float gx, float gy, float gz, float vx,
                                                    float vy, float vz, float dist, float inv_s) {
  const float true_cos = (vx * gx + vy * gy) + vz * gz;
  const float iter_cos = -(fmaxf(-true_cos, 0.f));
  const float half = iter_cos * dist * 0.5f;
  const float pc = sigmoidf_((sdf - half) * inv_s), nc = sigmoidf_((sdf + half) * inv_s);
  const float num = (pc - nc) + 1e-5f, dn = pc + 1e-5f, q = num / dn;
  AlphaGrad o = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (q >= 0.f && q <= 1.f) {  // clip passes the gradient on the closed interval
    const float d_p = g_alpha / dn;
    const float d_c = -g_alpha * num / (dn * dn);
    const float d_prev = (d_p + d_c) * (pc * (1.f - pc));
    const float d_next = -d_p * (nc * (1.f - nc));
    o.d_sdf = (d_prev + d_next) * inv_s;
    o.d_inv_s = d_prev * (sdf - half) + d_next * (sdf + half);   // the two sigmoid arguments are (sdf -/+ half) * inv_s
    const float d_half = (d_next - d_prev) * inv_s;
    const float d_iter = d_half * dist * 0.5f;
    const float d_cos = (true_cos < 0.f) ? d_iter : 0.f;  // iter_cos = cos where cos < 0, else 0
    o.dgx = d_cos * vx;
    o.dgy = d_cos * vy;
    o.dgz = d_cos * vz;
  }
  return o;
}

// Sum of `v` over the lanes of a wavefront, added to *dst by lane 0 (the s_learn gradient: d loss / d inv_s of model/nerf.py:
// 512-522, where s_val is a learnable parameter).
__device__ __forceinline__ void fgs_wave_atomic_sum(float v, float *dst) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((threadIdx.x & 63) == 0 && v != 0.f) atomicAdd(dst, v);
}
