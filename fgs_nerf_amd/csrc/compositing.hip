// compositing.hip -- density post-activation and early-terminating alpha compositing.
// Reference operators: model/cuda/render_utils_kernel.cu:430-707.
//
// alpha2weight on CDNA4: the reference runs ONE THREAD per ray (a 4096-ray batch is 16 wavefronts
// on a 256-CU part).  Here one 64-lane wavefront owns a ray: lanes load 64 consecutive alphas
// coalesced, the transmittance recurrence -- which the reference evaluates as a double-precision
// product narrowed to float every step -- is replayed in exactly that order by all lanes in
// lock-step (v_readlane broadcast of alpha_j), and the wave leaves the ray as soon as the running
// transmittance drops below 1e-3 (wave-uniform early exit).  Results are bit-identical to the
// sequential reference loop.
#include "fgs_common.h"

namespace {

// render_utils_kernel.cu:430-458
__global__ void k_raw2alpha(const float *__restrict__ density, float shift, float interval,
                            const float *__restrict__ interval_nonuni, int64_t n, float *__restrict__ exp_d,
                            float *__restrict__ alpha) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float iv = interval_nonuni ? interval_nonuni[i] : interval;
  const float e = expf(density[i] + shift);
  exp_d[i] = e;
  alpha[i] = 1.f - powf(1.f + e, -iv);
}

// render_utils_kernel.cu:506-530
__global__ void k_raw2alpha_bwd(const float *__restrict__ exp_d, const float *__restrict__ grad_back, float interval,
                                const float *__restrict__ interval_nonuni, int64_t n, float *__restrict__ grad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float iv = interval_nonuni ? interval_nonuni[i] : interval;
  const float e = exp_d[i];
  const double m = fmin((double)e, 1e10);
  const float pw = powf(1.f + e, -iv - 1.f);
  grad[i] = (float)(m * (double)pw * (double)iv * (double)grad_back[i]);
}

// Initialisation of the reference wrapper (render_utils_kernel.cu:624-628): weight=0, T=1,
// alphainv_last=1, i_start=i_end=0.
__global__ void k_a2w_init(int64_t n_pts, int64_t n_rays, float *__restrict__ weight, float *__restrict__ T,
                           float *__restrict__ alphainv_last, int64_t *__restrict__ i_start,
                           int64_t *__restrict__ i_end) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pts) {
    weight[i] = 0.f;
    T[i] = 1.f;
  }
  if (i < n_rays) {
    alphainv_last[i] = 1.f;
    i_start[i] = 0;
    i_end[i] = 0;
  }
}

// render_utils_kernel.cu:607-617 plus the host-side `i_end[ray_id[n_pts-1]] = n_pts` (:635)
__global__ void k_a2w_segments(const int64_t *__restrict__ ray_id, int64_t n_pts, int64_t *__restrict__ i_start,
                               int64_t *__restrict__ i_end) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pts) return;
  const int64_t r = ray_id[i];
  if (i > 0) {
    const int64_t rp = ray_id[i - 1];
    if (r != rp) {
      i_start[r] = i;
      i_end[rp] = i;
    }
  }
  if (i == n_pts - 1) i_end[r] = n_pts;
}

// render_utils_kernel.cu:576-605, one wavefront per ray.
__global__ __launch_bounds__(FGS_BLOCK) void k_a2w_fwd(const float *__restrict__ alpha, int64_t n_rays,
                                                       float *__restrict__ weight, float *__restrict__ T,
                                                       float *__restrict__ alphainv_last,
                                                       const int64_t *__restrict__ i_start, int64_t *__restrict__ i_end) {
  const int64_t ray = (int64_t)blockIdx.x * (FGS_BLOCK / FGS_WAVE) + fgs_uniform((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  if (ray >= n_rays) return;  // wave-uniform
  const int64_t i_s = fgs_uniform(i_start[ray]), i_e_max = fgs_uniform(i_end[ray]);
  float T_cum = 1.f;
  int64_t i_stop = i_e_max;
  for (int64_t base = i_s; base < i_e_max; base += FGS_WAVE) {
    const int64_t rem = i_e_max - base;
    const int cnt = rem < FGS_WAVE ? (int)rem : FGS_WAVE;
    const int64_t i = base + lane;
    const float a = (lane < cnt) ? alpha[i] : 0.f;
    float my_T = 1.f;
    int stop = -1;
    for (int j = 0; j < cnt; ++j) {  // uniform trip count; every lane replays the same chain
      const float aj = fgs_bcast_lane(a, j);
      if (lane == j) my_T = T_cum;
      T_cum = (float)((double)T_cum * (1. - (double)aj));
      if (fgs_uniform((double)T_cum < 1e-3 ? 1 : 0)) {  // same value in every lane
        stop = j;
        break;
      }
    }
    const int last = (stop >= 0) ? stop : cnt - 1;
    if (lane <= last) {
      T[i] = my_T;
      weight[i] = my_T * a;
    }
    if (stop >= 0) {
      i_stop = base + stop + 1;
      break;
    }
  }
  if (lane == 0) {
    i_end[ray] = i_stop;
    alphainv_last[ray] = T_cum;
  }
}

// render_utils_kernel.cu:653-677, one wavefront per ray, walking the segment back to front.
__global__ __launch_bounds__(FGS_BLOCK) void k_a2w_bwd(const float *__restrict__ alpha, const float *__restrict__ weight,
                                                       const float *__restrict__ T, const float *__restrict__ alphainv_last,
                                                       const int64_t *__restrict__ i_start, const int64_t *__restrict__ i_end,
                                                       int64_t n_rays, const float *__restrict__ grad_weights,
                                                       const float *__restrict__ grad_last, float *__restrict__ grad) {
  const int64_t ray = (int64_t)blockIdx.x * (FGS_BLOCK / FGS_WAVE) + fgs_uniform((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  if (ray >= n_rays) return;
  const int64_t i_s = fgs_uniform(i_start[ray]), i_e = fgs_uniform(i_end[ray]);
  float back_cum = grad_last[ray] * alphainv_last[ray];
  for (int64_t top = i_e; top > i_s; top -= FGS_WAVE) {
    const int64_t rem = top - i_s;
    const int cnt = rem < FGS_WAVE ? (int)rem : FGS_WAVE;
    const int64_t i = top - 1 - lane;  // lane 0 = last sample of the chunk
    float gw = 0.f, w = 0.f, tt = 0.f, a = 0.f;
    if (lane < cnt) {
      gw = grad_weights[i];
      w = weight[i];
      tt = T[i];
      a = alpha[i];
    }
    float my_back = 0.f;
    for (int j = 0; j < cnt; ++j) {
      if (lane == j) my_back = back_cum;
      back_cum = fmaf(fgs_bcast_lane(gw, j), fgs_bcast_lane(w, j), back_cum);
    }
    if (lane < cnt) {
      const double den = (double)(1.f - a) + 1e-10;
      grad[i] = (float)((double)(gw * tt) - (double)my_back / den);
    }
  }
}

__global__ void k_fill_zero(float *__restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

}  // namespace

FGS_API int fgs_raw2alpha(const float *density, float shift, float interval, const float *interval_nonuni, int64_t n_pts,
                          float *exp_d, float *alpha, fgs_stream_t stream) {
  FGS_REQUIRE(n_pts >= 0 && n_pts < FGS_MAX_ELEMS, FGS_E_RANGE, "fgs_raw2alpha: n_pts=%lld", (long long)n_pts);
  if (n_pts == 0) return 0;
  FGS_REQUIRE(density && exp_d && alpha, FGS_E_INVALID, "fgs_raw2alpha: null pointer");
  hipLaunchKernelGGL(k_raw2alpha, dim3(fgs_blocks(n_pts)), dim3(FGS_BLOCK), 0, fgs_s(stream), density, shift, interval,
                     interval_nonuni, n_pts, exp_d, alpha);
  FGS_LAUNCH_OK("fgs_raw2alpha");
  return 0;
}

FGS_API int fgs_raw2alpha_bwd(const float *exp_d, const float *grad_back, float interval, const float *interval_nonuni,
                              int64_t n_pts, float *grad, fgs_stream_t stream) {
  FGS_REQUIRE(n_pts >= 0 && n_pts < FGS_MAX_ELEMS, FGS_E_RANGE, "fgs_raw2alpha_bwd: n_pts=%lld", (long long)n_pts);
  if (n_pts == 0) return 0;
  FGS_REQUIRE(exp_d && grad_back && grad, FGS_E_INVALID, "fgs_raw2alpha_bwd: null pointer");
  hipLaunchKernelGGL(k_raw2alpha_bwd, dim3(fgs_blocks(n_pts)), dim3(FGS_BLOCK), 0, fgs_s(stream), exp_d, grad_back,
                     interval, interval_nonuni, n_pts, grad);
  FGS_LAUNCH_OK("fgs_raw2alpha_bwd");
  return 0;
}

FGS_API int fgs_alpha2weight_fwd(const float *alpha, const int64_t *ray_id, int64_t n_pts, int64_t n_rays, float *weight,
                                 float *T, float *alphainv_last, int64_t *i_start, int64_t *i_end, fgs_stream_t stream) {
  FGS_REQUIRE(n_pts >= 0 && n_pts < FGS_MAX_ELEMS && n_rays >= 0 && n_rays < FGS_MAX_ELEMS, FGS_E_RANGE,
              "fgs_alpha2weight_fwd: n_pts=%lld n_rays=%lld", (long long)n_pts, (long long)n_rays);
  const int64_t n_init = n_pts > n_rays ? n_pts : n_rays;
  if (n_init == 0) return 0;
  FGS_REQUIRE((n_pts == 0 || (alpha && ray_id && weight && T)) && (n_rays == 0 || (alphainv_last && i_start && i_end)),
              FGS_E_INVALID, "fgs_alpha2weight_fwd: null pointer");
  hipLaunchKernelGGL(k_a2w_init, dim3(fgs_blocks(n_init)), dim3(FGS_BLOCK), 0, fgs_s(stream), n_pts, n_rays, weight, T,
                     alphainv_last, i_start, i_end);
  FGS_LAUNCH_OK("fgs_alpha2weight_fwd/init");
  if (n_pts == 0 || n_rays == 0) return 0;  // render_utils_kernel.cu:629-631
  hipLaunchKernelGGL(k_a2w_segments, dim3(fgs_blocks(n_pts)), dim3(FGS_BLOCK), 0, fgs_s(stream), ray_id, n_pts, i_start,
                     i_end);
  FGS_LAUNCH_OK("fgs_alpha2weight_fwd/segments");
  hipLaunchKernelGGL(k_a2w_fwd, dim3(fgs_blocks(n_rays * FGS_WAVE)), dim3(FGS_BLOCK), 0, fgs_s(stream), alpha, n_rays,
                     weight, T, alphainv_last, (const int64_t *)i_start, i_end);
  FGS_LAUNCH_OK("fgs_alpha2weight_fwd/scan");
  return 0;
}

FGS_API int fgs_alpha2weight_bwd(const float *alpha, const float *weight, const float *T, const float *alphainv_last,
                                 const int64_t *i_start, const int64_t *i_end, int64_t n_pts, int64_t n_rays,
                                 const float *grad_weights, const float *grad_last, float *grad, fgs_stream_t stream) {
  FGS_REQUIRE(n_pts >= 0 && n_pts < FGS_MAX_ELEMS && n_rays >= 0 && n_rays < FGS_MAX_ELEMS, FGS_E_RANGE,
              "fgs_alpha2weight_bwd: n_pts=%lld n_rays=%lld", (long long)n_pts, (long long)n_rays);
  if (n_pts == 0) return 0;
  FGS_REQUIRE(grad, FGS_E_INVALID, "fgs_alpha2weight_bwd: null grad");
  hipLaunchKernelGGL(k_fill_zero, dim3(fgs_blocks(n_pts)), dim3(FGS_BLOCK), 0, fgs_s(stream), grad, n_pts);  // :684
  FGS_LAUNCH_OK("fgs_alpha2weight_bwd/zero");
  if (n_rays == 0) return 0;
  FGS_REQUIRE(alpha && weight && T && alphainv_last && i_start && i_end && grad_weights && grad_last, FGS_E_INVALID,
              "fgs_alpha2weight_bwd: null pointer");
  hipLaunchKernelGGL(k_a2w_bwd, dim3(fgs_blocks(n_rays * FGS_WAVE)), dim3(FGS_BLOCK), 0, fgs_s(stream), alpha, weight, T,
                     alphainv_last, i_start, i_end, n_rays, grad_weights, grad_last, grad);
  FGS_LAUNCH_OK("fgs_alpha2weight_bwd/scan");
  return 0;
}
