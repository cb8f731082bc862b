// gridopt.hip -- per-step dense grid work outside autograd: total-variation gradient and the
// (masked) Adam update.  Both are pure HBM streaming kernels.
// Reference operators: model/cuda/total_variation_kernel.cu:13-133, model/cuda/adam_upd_kernel.cu:8-133.
#include "fgs_common.h"

#include <math.h>

namespace {

__device__ __forceinline__ float clamp1(float v) { return fminf(fmaxf(v, -1.f), 1.f); }

// One thread per element, walked in MEMORY order so neighbouring lanes touch neighbouring addresses
// in either layout.  CHANNEL_LAST: memory index = ((x*Y + y)*Z + z)*C + c ; else ((c*X + x)*Y + y)*Z + z.
// Arithmetic follows total_variation_kernel.cu:22-33 (unmasked: wz,wz / wy,wy / wz,wz -- the
// reference's axis-weight quirk) and :47-58 (masked: wx / wy / wz, times mask[i]*mask[nbr]).
template <bool CHANNEL_LAST, bool MASKED>
__global__ __launch_bounds__(FGS_BLOCK) void k_tv_add_grad(const float *__restrict__ param, float *__restrict__ grad,
                                                           const float *__restrict__ mask, float wx, float wy, float wz,
                                                           int dense_mode, GridDesc d, int64_t N) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N) return;
  const float g0 = grad[idx];
  if (!(dense_mode || g0 != 0.f)) return;
  int64_t z, y, x;
  int64_t r = CHANNEL_LAST ? idx / d.C : idx;
  z = r % d.Z; r /= d.Z;
  y = r % d.Y; r /= d.Y;
  x = r % d.X;
  const float p = param[idx];
  float g = 0.f;
  if (!MASKED) {
    g += (z == 0       ? 0.f : wz * clamp1(p - param[idx - d.sZ]));
    g += (z == d.Z - 1 ? 0.f : wz * clamp1(p - param[idx + d.sZ]));
    g += (y == 0       ? 0.f : wy * clamp1(p - param[idx - d.sY]));
    g += (y == d.Y - 1 ? 0.f : wy * clamp1(p - param[idx + d.sY]));
    g += (x == 0       ? 0.f : wz * clamp1(p - param[idx - d.sX]));
    g += (x == d.X - 1 ? 0.f : wz * clamp1(p - param[idx + d.sX]));
  } else {
    const float m0 = mask[idx];
    g += (z == 0       ? 0.f : wx * clamp1(p - param[idx - d.sZ]) * m0 * mask[idx - d.sZ]);
    g += (z == d.Z - 1 ? 0.f : wx * clamp1(p - param[idx + d.sZ]) * m0 * mask[idx + d.sZ]);
    g += (y == 0       ? 0.f : wy * clamp1(p - param[idx - d.sY]) * m0 * mask[idx - d.sY]);
    g += (y == d.Y - 1 ? 0.f : wy * clamp1(p - param[idx + d.sY]) * m0 * mask[idx + d.sY]);
    g += (x == 0       ? 0.f : wz * clamp1(p - param[idx - d.sX]) * m0 * mask[idx - d.sX]);
    g += (x == d.X - 1 ? 0.f : wz * clamp1(p - param[idx + d.sX]) * m0 * mask[idx + d.sX]);
  }
  grad[idx] = g0 + g;
}

// Channel-first, unmasked, Z % 4 == 0, < 2^31 elements (the sdf / density grids): four consecutive z per thread.  The
// centre row and the four y / x neighbour rows are 16-byte loads, the two z neighbours outside the quad two scalars
// (7 load instructions per 4 elements instead of 28), 32-bit index arithmetic; the sums are formed element by element in
// the reference's order, so the result is bit-identical to k_tv_add_grad<false, false>.
template <bool DENSE>
__global__ __launch_bounds__(FGS_BLOCK) void k_tv_add_grad_cf4(const float *__restrict__ param, float *__restrict__ grad,
                                                               float wy, float wz, int X, int Y, int Z, unsigned n4) {
  constexpr int dense_mode = DENSE ? 1 : 0;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n4) return;
  const unsigned idx = 4u * t;
  const float4 g0 = *reinterpret_cast<const float4 *>(grad + idx);
  // (dense mode -- a template parameter -- must not wait for the gradient before it asks for the parameters: the test below made
  // every thread's seven loads two dependent round trips, 21 us for 48 MB at 160^3)
  if (!DENSE && !(g0.x != 0.f || g0.y != 0.f || g0.z != 0.f || g0.w != 0.f)) return;
  unsigned r = idx;
  const unsigned z0 = r % (unsigned)Z; r /= (unsigned)Z;
  const unsigned y = r % (unsigned)Y; r /= (unsigned)Y;
  const unsigned x = r % (unsigned)X;
  const unsigned sY = (unsigned)Z, sX = (unsigned)Z * (unsigned)Y;
  const float4 p = *reinterpret_cast<const float4 *>(param + idx);
  const float zm = (z0 == 0) ? 0.f : param[idx - 1];
  const float zp = (z0 + 4 >= (unsigned)Z) ? 0.f : param[idx + 4];
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 ym = (y == 0) ? zero : *reinterpret_cast<const float4 *>(param + idx - sY);
  const float4 yp = (y == (unsigned)Y - 1) ? zero : *reinterpret_cast<const float4 *>(param + idx + sY);
  const float4 xm = (x == 0) ? zero : *reinterpret_cast<const float4 *>(param + idx - sX);
  const float4 xp = (x == (unsigned)X - 1) ? zero : *reinterpret_cast<const float4 *>(param + idx + sX);
  const float pc[4] = {p.x, p.y, p.z, p.w};
  const float below[4] = {zm, p.x, p.y, p.z}, above[4] = {p.y, p.z, p.w, zp};
  const float yml[4] = {ym.x, ym.y, ym.z, ym.w}, ypl[4] = {yp.x, yp.y, yp.z, yp.w};
  const float xml[4] = {xm.x, xm.y, xm.z, xm.w}, xpl[4] = {xp.x, xp.y, xp.z, xp.w};
  const float gin[4] = {g0.x, g0.y, g0.z, g0.w};
  float out[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned z = z0 + k;
    float g = 0.f;
    g += (z == 0                 ? 0.f : wz * clamp1(pc[k] - below[k]));
    g += (z == (unsigned)Z - 1   ? 0.f : wz * clamp1(pc[k] - above[k]));
    g += (y == 0                 ? 0.f : wy * clamp1(pc[k] - yml[k]));
    g += (y == (unsigned)Y - 1   ? 0.f : wy * clamp1(pc[k] - ypl[k]));
    g += (x == 0                 ? 0.f : wz * clamp1(pc[k] - xml[k]));
    g += (x == (unsigned)X - 1   ? 0.f : wz * clamp1(pc[k] - xpl[k]));
    out[k] = (dense_mode || gin[k] != 0.f) ? gin[k] + g : gin[k];
  }
  *reinterpret_cast<float4 *>(grad + idx) = make_float4(out[0], out[1], out[2], out[3]);
}

// adam_upd_kernel.cu:8-58.  MODE: 0 dense, 1 masked (skip grad == 0), 2 per-voxel lr.
// Contraction pinned as in oracle/fgs_oracle.c orc_adam_upd.
template <int MODE>
__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float perlr, float step_size, float beta1,
                                         float beta2, float eps) {
  m = fmaf(beta1, m, (1.f - beta1) * g);
  v = fmaf(beta2, v, (1.f - beta2) * g * g);
  const float num = (MODE == 2) ? step_size * perlr * m : step_size * m;
  p -= num / (sqrtf(v) + eps);
}

// 4 elements per thread through 16-byte accesses; the masked form reads only grad when all four
// gradients are zero (the common case for a feature grid that rays touch sparsely).
template <int MODE>
__global__ __launch_bounds__(FGS_BLOCK) void k_adam_vec4(float4 *__restrict__ param, const float4 *__restrict__ grad,
                                                         float4 *__restrict__ exp_avg, float4 *__restrict__ exp_avg_sq,
                                                         const float4 *__restrict__ perlr, int64_t n4, float step_size,
                                                         float beta1, float beta2, float eps,
                                                         const float *__restrict__ ss_dev, const int *__restrict__ skip) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  if (skip && *skip) return;              // a step whose survivor list overflowed its capacity updates nothing
  if (ss_dev) step_size = *ss_dev;        // device-resident schedule (fgs_step_scalars_tick)
  const float4 g = grad[i];
  if (MODE == 1 && g.x == 0.f && g.y == 0.f && g.z == 0.f && g.w == 0.f) return;
  float4 p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
  float4 l = make_float4(1.f, 1.f, 1.f, 1.f);
  if (MODE == 2) l = perlr[i];
  if (MODE != 1 || g.x != 0.f) adam_one<MODE>(p.x, g.x, m.x, v.x, l.x, step_size, beta1, beta2, eps);
  if (MODE != 1 || g.y != 0.f) adam_one<MODE>(p.y, g.y, m.y, v.y, l.y, step_size, beta1, beta2, eps);
  if (MODE != 1 || g.z != 0.f) adam_one<MODE>(p.z, g.z, m.z, v.z, l.z, step_size, beta1, beta2, eps);
  if (MODE != 1 || g.w != 0.f) adam_one<MODE>(p.w, g.w, m.w, v.w, l.w, step_size, beta1, beta2, eps);
  param[i] = p;
  exp_avg[i] = m;
  exp_avg_sq[i] = v;
}

template <int MODE>
__global__ void k_adam_scalar(float *__restrict__ param, const float *__restrict__ grad, float *__restrict__ exp_avg,
                              float *__restrict__ exp_avg_sq, const float *__restrict__ perlr, int64_t begin, int64_t n,
                              float step_size, float beta1, float beta2, float eps, const float *__restrict__ ss_dev,
                              const int *__restrict__ skip) {
  const int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (skip && *skip) return;
  if (ss_dev) step_size = *ss_dev;
  const float g = grad[i];
  if (MODE == 1 && g == 0.f) return;
  float p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
  adam_one<MODE>(p, g, m, v, MODE == 2 ? perlr[i] : 1.f, step_size, beta1, beta2, eps);
  param[i] = p;
  exp_avg[i] = m;
  exp_avg_sq[i] = v;
}

template <int MODE>
int launch_adam(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, const float *perlr, int64_t n,
                float step_size, float beta1, float beta2, float eps, hipStream_t st, const float *ss_dev = nullptr,
                const int *skip = nullptr) {
  const bool aligned = (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq |
                         (uintptr_t)(MODE == 2 ? perlr : nullptr)) & 15) == 0;
  const int64_t n4 = aligned ? n / 4 : 0;
  if (n4 > 0) {
    hipLaunchKernelGGL(k_adam_vec4<MODE>, dim3(fgs_blocks(n4)), dim3(FGS_BLOCK), 0, st, (float4 *)param,
                       (const float4 *)grad, (float4 *)exp_avg, (float4 *)exp_avg_sq, (const float4 *)perlr, n4, step_size,
                       beta1, beta2, eps, ss_dev, skip);
    FGS_LAUNCH_OK("fgs_adam_upd/vec4");
  }
  const int64_t done = n4 * 4;
  if (done < n) {
    hipLaunchKernelGGL(k_adam_scalar<MODE>, dim3(fgs_blocks(n - done)), dim3(FGS_BLOCK), 0, st, param, grad, exp_avg,
                       exp_avg_sq, perlr, done, n, step_size, beta1, beta2, eps, ss_dev, skip);
    FGS_LAUNCH_OK("fgs_adam_upd/tail");
  }
  return 0;
}

}  // namespace

FGS_API int fgs_tv_add_grad(const float *param, float *grad, const float *mask, float wx, float wy, float wz,
                            int dense_mode, int64_t C, int64_t X, int64_t Y, int64_t Z, int64_t sC, int64_t sX, int64_t sY,
                            int64_t sZ, fgs_stream_t stream) {
  FGS_REQUIRE(C > 0 && X > 0 && Y > 0 && Z > 0, FGS_E_INVALID, "fgs_tv_add_grad: empty grid");
  const int64_t N = C * X * Y * Z;
  FGS_REQUIRE(N < FGS_MAX_ELEMS, FGS_E_RANGE, "fgs_tv_add_grad: %lld elements", (long long)N);
  FGS_REQUIRE(param && grad, FGS_E_INVALID, "fgs_tv_add_grad: null pointer");
  const bool ch_first = (sZ == 1 && sY == Z && sX == Y * Z && (C == 1 || sC == X * Y * Z));
  const bool ch_last = (sC == 1 && sZ == C && sY == Z * C && sX == Y * Z * C);
  FGS_REQUIRE(ch_first || ch_last, FGS_E_INVALID,
              "fgs_tv_add_grad: strides (%lld,%lld,%lld,%lld) are neither channel-first nor channel-last dense",
              (long long)sC, (long long)sX, (long long)sY, (long long)sZ);
  // total_variation_kernel.cu:76-78 / :112-114
  wx /= 6; wy /= 6; wz /= 6;
  const GridDesc d{C, X, Y, Z, sC, sX, sY, sZ};
  const dim3 g(fgs_blocks(N)), b(FGS_BLOCK);
  hipStream_t st = fgs_s(stream);
  // channel-first with C==1 is also a valid channel-last walk; prefer the cheaper decode
  if (ch_first && !mask && C == 1 && (Z % 4) == 0 && N < ((int64_t)1 << 31) &&
      (reinterpret_cast<uintptr_t>(param) & 15) == 0 && (reinterpret_cast<uintptr_t>(grad) & 15) == 0) {
    const unsigned n4 = (unsigned)(N / 4);
    if (dense_mode) hipLaunchKernelGGL(k_tv_add_grad_cf4<true>, dim3(fgs_blocks(n4)), b, 0, st, param, grad, wy, wz, (int)X, (int)Y, (int)Z, n4);
    else hipLaunchKernelGGL(k_tv_add_grad_cf4<false>, dim3(fgs_blocks(n4)), b, 0, st, param, grad, wy, wz, (int)X, (int)Y, (int)Z, n4);
  } else if (ch_first) {
    if (mask) hipLaunchKernelGGL((k_tv_add_grad<false, true>), g, b, 0, st, param, grad, mask, wx, wy, wz, dense_mode, d, N);
    else      hipLaunchKernelGGL((k_tv_add_grad<false, false>), g, b, 0, st, param, grad, mask, wx, wy, wz, dense_mode, d, N);
  } else {
    if (mask) hipLaunchKernelGGL((k_tv_add_grad<true, true>), g, b, 0, st, param, grad, mask, wx, wy, wz, dense_mode, d, N);
    else      hipLaunchKernelGGL((k_tv_add_grad<true, false>), g, b, 0, st, param, grad, mask, wx, wy, wz, dense_mode, d, N);
  }
  FGS_LAUNCH_OK("fgs_tv_add_grad");
  return 0;
}

FGS_API int fgs_adam_upd(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, const float *perlr, int64_t n,
                         int step, float beta1, float beta2, float lr, float eps, int mode, fgs_stream_t stream) {
  FGS_REQUIRE(n >= 0 && n < FGS_MAX_ELEMS, FGS_E_RANGE, "fgs_adam_upd: n=%lld", (long long)n);
  FGS_REQUIRE(mode >= 0 && mode <= 2, FGS_E_INVALID, "fgs_adam_upd: mode=%d", mode);
  if (n == 0) return 0;
  FGS_REQUIRE(param && grad && exp_avg && exp_avg_sq && (mode != FGS_ADAM_PERLR || perlr), FGS_E_INVALID,
              "fgs_adam_upd: null pointer");
  // adam_upd_kernel.cu:72 -- all-float host arithmetic
  const float step_size = lr * sqrtf(1.f - powf(beta2, (float)step)) / (1.f - powf(beta1, (float)step));
  hipStream_t st = fgs_s(stream);
  switch (mode) {
    case FGS_ADAM_DENSE:  return launch_adam<0>(param, grad, exp_avg, exp_avg_sq, perlr, n, step_size, beta1, beta2, eps, st);
    case FGS_ADAM_MASKED: return launch_adam<1>(param, grad, exp_avg, exp_avg_sq, perlr, n, step_size, beta1, beta2, eps, st);
    default:              return launch_adam<2>(param, grad, exp_avg, exp_avg_sq, perlr, n, step_size, beta1, beta2, eps, st);
  }
}

// ---- many small tensors in one launch (the 16 Linear weights / biases of the two MLPs) -----------------------------
namespace {
constexpr int MULTI_MAX = 32;
struct MultiAdam {
  int n_tensors;
  float *param[MULTI_MAX];
  const float *grad[MULTI_MAX];
  float *m[MULTI_MAX], *v[MULTI_MAX];
  int64_t size[MULTI_MAX];
  unsigned blk_start[MULTI_MAX + 1];
  float step_size[MULTI_MAX];
  const float *ss_dev[MULTI_MAX];   // device-resident step size per tensor (or null: step_size above)
  const int *skip;
  int masked[MULTI_MAX];
};

__global__ __launch_bounds__(FGS_BLOCK) void k_adam_multi(MultiAdam A, float beta1, float beta2, float eps) {
  int t = 0;
  while (t + 1 < A.n_tensors && blockIdx.x >= A.blk_start[t + 1]) ++t;  // uniform per block
  const int64_t i = (int64_t)(blockIdx.x - A.blk_start[t]) * blockDim.x + threadIdx.x;
  if (i >= A.size[t]) return;
  if (A.skip && *A.skip) return;
  const float g = A.grad[t][i];
  if (A.masked[t] && g == 0.f) return;
  float p = A.param[t][i], m = A.m[t][i], v = A.v[t][i];
  adam_one<0>(p, g, m, v, 1.f, A.ss_dev[t] ? *A.ss_dev[t] : A.step_size[t], beta1, beta2, eps);
  A.param[t][i] = p;
  A.m[t][i] = m;
  A.v[t][i] = v;
}
}  // namespace

static int adam_multi_impl(int n_tensors, float *const *params, const float *const *grads, float *const *exp_avgs,
                           float *const *exp_avg_sqs, const int64_t *sizes, const int *steps, const float *lrs,
                           const float *const *ss_dev, const int *skip, const int *masked, float beta1, float beta2,
                           float eps, fgs_stream_t stream) {
  FGS_REQUIRE(n_tensors >= 0, FGS_E_INVALID, "fgs_adam_upd_multi: n_tensors=%d", n_tensors);
  if (n_tensors == 0) return 0;
  FGS_REQUIRE(params && grads && exp_avgs && exp_avg_sqs && sizes && (ss_dev || (steps && lrs)) && masked, FGS_E_INVALID,
              "fgs_adam_upd_multi: null table");
  for (int base = 0; base < n_tensors; base += MULTI_MAX) {
    MultiAdam A;
    A.n_tensors = (n_tensors - base < MULTI_MAX) ? n_tensors - base : MULTI_MAX;
    unsigned blocks = 0;
    for (int t = 0; t < A.n_tensors; ++t) {
      const int s = base + t;
      FGS_REQUIRE(params[s] && grads[s] && exp_avgs[s] && exp_avg_sqs[s] && sizes[s] >= 0 && sizes[s] < ((int64_t)1 << 31),
                  FGS_E_INVALID, "fgs_adam_upd_multi: bad tensor %d", s);
      A.param[t] = params[s]; A.grad[t] = grads[s]; A.m[t] = exp_avgs[s]; A.v[t] = exp_avg_sqs[s];
      A.size[t] = sizes[s];
      A.masked[t] = masked[s];
      // adam_upd_kernel.cu:72, per tensor (each keeps its own step count)
      A.ss_dev[t] = ss_dev ? ss_dev[s] : nullptr;
      A.step_size[t] = ss_dev ? 0.f : lrs[s] * sqrtf(1.f - powf(beta2, (float)steps[s])) / (1.f - powf(beta1, (float)steps[s]));
      A.blk_start[t] = blocks;
      blocks += fgs_blocks(sizes[s]);
    }
    A.blk_start[A.n_tensors] = blocks;
    A.skip = skip;
    if (blocks == 0) continue;
    hipLaunchKernelGGL(k_adam_multi, dim3(blocks), dim3(FGS_BLOCK), 0, fgs_s(stream), A, beta1, beta2, eps);
    FGS_LAUNCH_OK("fgs_adam_upd_multi");
  }
  return 0;
}

FGS_API int fgs_adam_upd_multi(int n_tensors, float *const *params, const float *const *grads, float *const *exp_avgs,
                               float *const *exp_avg_sqs, const int64_t *sizes, const int *steps, const float *lrs,
                               const int *masked, float beta1, float beta2, float eps, fgs_stream_t stream) {
  return adam_multi_impl(n_tensors, params, grads, exp_avgs, exp_avg_sqs, sizes, steps, lrs, nullptr, nullptr, masked, beta1,
                         beta2, eps, stream);
}

// ---- device-resident schedule: the forms a captured (hipGraph) training step uses -----------------------------------
// The per-iteration scalars of a step -- Adam's step size lr_t sqrt(1 - b2^t) / (1 - b1^t) of every parameter group, NeuS
// 1/s -- are rows of a table the HOST computes once, with exactly the arithmetic of the per-call entry points
// (fgs_adam_step_size), for all iterations of a stage.  fgs_step_scalars_tick copies row min(counter, n_rows - 1) into `out`
// and advances the counter; the kernels of the step read their scalar from `out`.  Bit-identical to the host-driven step.
FGS_API float fgs_adam_step_size(int step, float beta1, float beta2, float lr) {
  return lr * sqrtf(1.f - powf(beta2, (float)step)) / (1.f - powf(beta1, (float)step));   // adam_upd_kernel.cu:72
}

namespace {
__global__ void k_scalars_tick(const float *__restrict__ table, int n_rows, int n_cols, int64_t *__restrict__ counter,
                               float *__restrict__ out, int mirror_col, float *__restrict__ mirror_dst, int latch_col,
                               float *__restrict__ latch_dst, const int *__restrict__ latch_flag_src,
                               int *__restrict__ latch_flag_dst) {
  // the PREVIOUS iteration's value of one column and of one flag, kept for an update of that iteration that is issued at the head
  // of this one (graph_step.CapturedFineStep: k0's Adam pass beside the next forward march)
  if (threadIdx.x == 0) {
    if (latch_dst) *latch_dst = out[latch_col];
    if (latch_flag_dst) *latch_flag_dst = *latch_flag_src;
  }
  __syncthreads();
  int64_t row = *counter;
  if (row > n_rows - 1) row = n_rows - 1;
  if (row < 0) row = 0;
  for (int c = threadIdx.x; c < n_cols; c += blockDim.x) {
    const float v = table[row * n_cols + c];
    out[c] = v;
    if (mirror_dst && c == mirror_col) *mirror_dst = v;     // e.g. the model's s_val parameter (model/nerf.py:520)
  }
  __syncthreads();
  if (threadIdx.x == 0) *counter = *counter + 1;
}

// offsets[0..n): the per-ray survivor offsets, offsets[n-1] = the survivor count.  flags[0]: sticky "some step overflowed";
// flags[1]: this step overflowed (the optimizer kernels skip on it); total += the rows the step really processed; and every
// offset is cut at the capacity, so that the per-ray segments every later kernel walks end inside the buffers
__global__ void k_count_guard(int64_t *__restrict__ offsets, int64_t n, int64_t capacity, int *__restrict__ flags,
                              int64_t *__restrict__ total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t v = offsets[i];
  if (i == n - 1) {                       // the total
    const int over = v > capacity ? 1 : 0;
    flags[1] = over;
    if (over) flags[0] = 1;
    if (total) *total += over ? capacity : v;
  }
  if (v > capacity) offsets[i] = capacity;   // every consumer of the offsets stays inside the buffers
}
}  // namespace

FGS_API int fgs_step_scalars_tick(const float *table, int n_rows, int n_cols, int64_t *counter, float *out, int mirror_col,
                                  float *mirror_dst, fgs_stream_t stream) {
  FGS_REQUIRE(table && counter && out && n_rows > 0 && n_cols > 0 && (!mirror_dst || (mirror_col >= 0 && mirror_col < n_cols)),
              FGS_E_INVALID, "fgs_step_scalars_tick: bad argument");
  hipLaunchKernelGGL(k_scalars_tick, dim3(1), dim3(64), 0, fgs_s(stream), table, n_rows, n_cols, counter, out, mirror_col,
                     mirror_dst, 0, (float *)nullptr, (const int *)nullptr, (int *)nullptr);
  FGS_LAUNCH_OK("fgs_step_scalars_tick");
  return 0;
}

FGS_API int fgs_step_scalars_tick2(const float *table, int n_rows, int n_cols, int64_t *counter, float *out, int mirror_col,
                                   float *mirror_dst, int latch_col, float *latch_dst, const int *latch_flag_src,
                                   int *latch_flag_dst, fgs_stream_t stream) {
  FGS_REQUIRE(table && counter && out && n_rows > 0 && n_cols > 0 && (!mirror_dst || (mirror_col >= 0 && mirror_col < n_cols)) &&
                  (!latch_dst || (latch_col >= 0 && latch_col < n_cols)) && (!latch_flag_dst || latch_flag_src),
              FGS_E_INVALID, "fgs_step_scalars_tick2: bad argument");
  hipLaunchKernelGGL(k_scalars_tick, dim3(1), dim3(64), 0, fgs_s(stream), table, n_rows, n_cols, counter, out, mirror_col,
                     mirror_dst, latch_col, latch_dst, latch_flag_src, latch_flag_dst);
  FGS_LAUNCH_OK("fgs_step_scalars_tick2");
  return 0;
}

namespace {
// mask[i][j][k] = every index inside its axis' closed range; the six bounds are floats holding small integers (a row of the
// schedule table): a captured iteration of the voxel-increment phase rebuilds the mask from them (model/nerf.py:1078-1088)
__global__ __launch_bounds__(FGS_BLOCK) void k_box_mask_fill(unsigned char *__restrict__ mask, int X, int Y, int Z,
                                                             const float *__restrict__ bounds) {
  const int64_t n = (int64_t)X * Y * Z;
  const int lo0 = (int)bounds[0], hi0 = (int)bounds[1], lo1 = (int)bounds[2], hi1 = (int)bounds[3], lo2 = (int)bounds[4],
            hi2 = (int)bounds[5];
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(v % Z), j = (int)((v / Z) % Y), i = (int)(v / ((int64_t)Z * Y));
    mask[v] = (i >= lo0 && i <= hi0 && j >= lo1 && j <= hi1 && k >= lo2 && k <= hi2) ? 1 : 0;
  }
}
}  // namespace

// The voxel-increment mask of one iteration (model/nerf.py:1078-1088, model/nerf_training.py:286-291) from six index bounds in
// DEVICE memory {lo_x, hi_x, lo_y, hi_y, lo_z, hi_z} (floats holding integers; lo > hi: empty axis): mask [X][Y][Z] bytes,
// 1 inside.  The host derives the bounds from the reference's own linspace comparison (fgs_nerf_amd.nerf.inc_index_bounds).
FGS_API int fgs_box_mask_fill(unsigned char *mask, int X, int Y, int Z, const float *bounds6_dev, fgs_stream_t stream) {
  FGS_REQUIRE(mask && bounds6_dev && X > 0 && Y > 0 && Z > 0, FGS_E_INVALID, "fgs_box_mask_fill: bad argument");
  const int64_t n = (int64_t)X * Y * Z;
  const int64_t blocks = (n + FGS_BLOCK * 4 - 1) / (FGS_BLOCK * 4);
  hipLaunchKernelGGL(k_box_mask_fill, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(FGS_BLOCK), 0, fgs_s(stream), mask,
                     X, Y, Z, bounds6_dev);
  FGS_LAUNCH_OK("fgs_box_mask_fill");
  return 0;
}

FGS_API int fgs_count_guard(int64_t *offsets, int64_t n, int64_t capacity, int *flags, int64_t *total, fgs_stream_t stream) {
  FGS_REQUIRE(offsets && flags && capacity >= 0 && n > 0, FGS_E_INVALID, "fgs_count_guard: bad argument");
  hipLaunchKernelGGL(k_count_guard, dim3(fgs_blocks(n)), dim3(FGS_BLOCK), 0, fgs_s(stream), offsets, n, capacity, flags, total);
  FGS_LAUNCH_OK("fgs_count_guard");
  return 0;
}

FGS_API int fgs_adam_upd_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, const float *perlr, int64_t n,
                             const float *step_size_dev, int step, float lr, float beta1, float beta2, float eps, int mode,
                             const int *skip_dev, fgs_stream_t stream) {
  FGS_REQUIRE(n >= 0 && n < FGS_MAX_ELEMS, FGS_E_RANGE, "fgs_adam_upd_dev: n=%lld", (long long)n);
  FGS_REQUIRE(mode >= 0 && mode <= 2, FGS_E_INVALID, "fgs_adam_upd_dev: mode=%d", mode);
  if (n == 0) return 0;
  FGS_REQUIRE(param && grad && exp_avg && exp_avg_sq && (mode != FGS_ADAM_PERLR || perlr), FGS_E_INVALID,
              "fgs_adam_upd_dev: null pointer");
  // no device-resident step size: the host arithmetic of fgs_adam_upd (adam_upd_kernel.cu:72), with the skip flag honoured
  const float ss = step_size_dev ? 0.f : fgs_adam_step_size(step, beta1, beta2, lr);
  hipStream_t st = fgs_s(stream);
  switch (mode) {
    case FGS_ADAM_DENSE:  return launch_adam<0>(param, grad, exp_avg, exp_avg_sq, perlr, n, ss, beta1, beta2, eps, st, step_size_dev, skip_dev);
    case FGS_ADAM_MASKED: return launch_adam<1>(param, grad, exp_avg, exp_avg_sq, perlr, n, ss, beta1, beta2, eps, st, step_size_dev, skip_dev);
    default:              return launch_adam<2>(param, grad, exp_avg, exp_avg_sq, perlr, n, ss, beta1, beta2, eps, st, step_size_dev, skip_dev);
  }
}

FGS_API int fgs_adam_upd_multi_dev(int n_tensors, float *const *params, const float *const *grads, float *const *exp_avgs,
                                   float *const *exp_avg_sqs, const int64_t *sizes, const float *const *step_size_dev,
                                   const int *steps, const float *lrs, const int *masked, float beta1, float beta2, float eps,
                                   const int *skip_dev, fgs_stream_t stream) {
  FGS_REQUIRE(step_size_dev || (steps && lrs), FGS_E_INVALID, "fgs_adam_upd_multi_dev: neither a step-size table nor steps + lrs");
  return adam_multi_impl(n_tensors, params, grads, exp_avgs, exp_avg_sqs, sizes, steps, lrs, step_size_dev, skip_dev,
                         masked, beta1, beta2, eps, stream);
}
