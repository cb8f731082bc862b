// tvloss.hip -- the autograd-form total-variation losses of the `ori_tv` configurations as a value pass and a gradient pass.
//
// Reference: total_variation(v, mask) of model/nerf.py:1212-1221 (sum over the three axes of |v[i+1] - v[i]| over the pairs
// whose two voxels are both inside the mask, divided by 3 and by mask.sum() -- or by v.sum() without a mask) and of
// model/dvgo.py:420-428 (per-axis MEANS over the valid pairs), called by density_total_variation (model/nerf.py:430-447,
// sdf_tv > 0; model/dvgo.py:206-208) and k0_total_variation (model/nerf.py:449-459; model/dvgo.py:210-215) and
// differentiated by autograd: diff -> abs -> boolean index -> sum, each a dense torch kernel over the grid and each saving
// a tensor of the grid's size for backward (for the 12-channel feature grid at 160^3: ~20 passes over 197 MB).
// Here: one streaming pass for the value (per-axis sums of |differences|, the plain sum of v, and the per-axis pair counts
// when a caller needs means), one for the gradient -- d/dv[j] of sum |v[i+1] - v[i]| is, per axis, sign(v[j] - v[j-1]) for
// the pair below minus sign(v[j+1] - v[j]) for the pair above (sign(0) = 0 as torch's abs backward has it) -- scaled per
// axis by device-resident factors, so neither pass needs a value from the host.  HBM-bound: 4 B read per element for the
// value (neighbours from cache), 4 B read + 4 B written for the gradient.
#include "fgs_common.h"

namespace {

__device__ __forceinline__ float sgnf(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

struct TvGrid {
  GridDesc d;
  const unsigned char *mask;   // [X][Y][Z] (shared by all channels) or null
  int64_t n;                   // C * X * Y * Z
};

// element index i -> (c, x, y, z) in the order that makes consecutive threads touch consecutive memory
template <bool CH_LAST>
__device__ __forceinline__ void tv_decode(const TvGrid &g, int64_t i, int64_t &c, int64_t &x, int64_t &y, int64_t &z) {
  if (g.n < ((int64_t)1 << 31)) {      // (uniform) 32-bit divisions: the 64-bit ones were most of the value pass's instructions
    unsigned j = (unsigned)i;
    const unsigned C = (unsigned)g.d.C, X = (unsigned)g.d.X, Y = (unsigned)g.d.Y, Z = (unsigned)g.d.Z;
    if (CH_LAST) {
      c = j % C; j /= C;
      z = j % Z; j /= Z;
      y = j % Y; x = j / Y;
    } else {
      z = j % Z; j /= Z;
      y = j % Y; j /= Y;
      x = j % X; c = j / X;
    }
    return;
  }
  if (CH_LAST) {
    c = i % g.d.C; i /= g.d.C;
    z = i % g.d.Z; i /= g.d.Z;
    y = i % g.d.Y; x = i / g.d.Y;
  } else {
    z = i % g.d.Z; i /= g.d.Z;
    y = i % g.d.Y; i /= g.d.Y;
    x = i % g.d.X; c = i / g.d.X;
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// What the last workgroup makes of the seven sums (the scalar algebra of total_variation and of its backward, in double):
struct TvFin {
  const int64_t *count;   // device: mask.sum() of the caller's mask tensor (the denominator with a mask), or null: v.sum()
  int per_axis_mean;      // model/dvgo.py:420-428: (mean_x + mean_y + mean_z) / 3 over the valid pairs
  double scale;           // factor on the term (sdf_tv / 2 / voxel_size * weight_tv_density ...)
  const float *add_in;    // device scalar added to the scaled term, or null
  float *loss_out;        // scale * tv (+ *add_in)
  float *w_out;           // 4 floats for k_tv_loss_grad: scale * d tv / d S_axis, and scale * d tv / d sum(v)
  double *sums_out;       // optional: the seven sums {S_x, S_y, S_z, sum(v), pairs_x, pairs_y, pairs_z}
};

// One partial of seven doubles per workgroup, summed in a fixed order by the workgroup that arrives last: S_axis = sum over the
// valid pairs along the axis of |v[i+1] - v[i]|, sum(v), pairs per axis.  (Until round 4 every workgroup added into seven
// doubles with atomics -- 28 K atomics on seven addresses were 50 of the pass's 56 us at 114^3 -- and ~30 small torch launches
// turned the sums into the loss and the backward factors.)
template <bool CH_LAST>
__global__ __launch_bounds__(FGS_BLOCK) void k_tv_loss_value(const float *__restrict__ v, TvGrid g, double *partials, unsigned *counter,
                                                             TvFin fin) {
  float s[3] = {0.f, 0.f, 0.f}, tot = 0.f;
  unsigned cnt[3] = {0u, 0u, 0u};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // Branch-free per element: the neighbour loads go to clamped (always valid) addresses and predicates select afterwards, so the
  // loads of the unrolled iterations are all in flight together.  (With `continue` and nested conditions every element was a
  // chain of three dependent cache round trips and a thread walked its six elements one after the other: 30 us at 114^3 for a
  // pass that moves 7.5 MB.)
  const bool has_mask = g.mask != nullptr;
#pragma unroll 3
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += stride) {
    int64_t c, x, y, z;
    tv_decode<CH_LAST>(g, i, c, x, y, z);
    const int64_t o = c * g.d.sC + x * g.d.sX + y * g.d.sY + z * g.d.sZ;
    const int64_t mo = (x * g.d.Y + y) * g.d.Z + z;
    const bool bx = x + 1 < g.d.X, by = y + 1 < g.d.Y, bz = z + 1 < g.d.Z;
    const float a = v[o];
    const float ax = v[bx ? o + g.d.sX : o], ay = v[by ? o + g.d.sY : o], az = v[bz ? o + g.d.sZ : o];
    bool m0 = true, mx = true, my = true, mz = true;
    if (has_mask) {        // (uniform)
      m0 = g.mask[mo] != 0;
      mx = g.mask[bx ? mo + g.d.Y * g.d.Z : mo] != 0;
      my = g.mask[by ? mo + g.d.Z : mo] != 0;
      mz = g.mask[bz ? mo + 1 : mo] != 0;
    }
    tot += a;
    const bool px = m0 && bx && mx, py = m0 && by && my, pz = m0 && bz && mz;
    s[0] += px ? fabsf(ax - a) : 0.f;
    s[1] += py ? fabsf(ay - a) : 0.f;
    s[2] += pz ? fabsf(az - a) : 0.f;
    cnt[0] += px ? 1u : 0u;
    cnt[1] += py ? 1u : 0u;
    cnt[2] += pz ? 1u : 0u;
  }
  __shared__ double part[FGS_BLOCK / FGS_WAVE][7];
  __shared__ int is_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double r[7] = {(double)s[0], (double)s[1], (double)s[2], (double)tot, (double)cnt[0], (double)cnt[1], (double)cnt[2]};
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    r[k] = wave_sum(r[k]);
    if (lane == 0) part[wave][k] = r[k];
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < FGS_BLOCK / FGS_WAVE; ++w) t += part[w][threadIdx.x];
    __hip_atomic_store(partials + 7 * (int64_t)blockIdx.x + threadIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
  }
  __syncthreads();
  if (threadIdx.x == 0) is_last = fgs_arrive_is_last(counter, gridDim.x) ? 1 : 0;
  __syncthreads();
  if (!is_last) return;
  // 7 x 36 threads: thread (k, j) sums partials b = j, j + 36, ... of sum k; then the 36 in order
  __shared__ double fin_part[7][36];
  if (threadIdx.x < 7 * 36) {
    const unsigned k = threadIdx.x % 7u, j = threadIdx.x / 7u;
    fin_part[k][j] = fgs_partials_sum<double>(partials + k, 7u * j, 7u * gridDim.x, 7u * 36u);
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  double S[7];
  for (int k = 0; k < 7; ++k) {
    double t = 0.0;
    for (int j = 0; j < 36; ++j) t += fin_part[k][j];
    S[k] = t;
    if (fin.sums_out) fin.sums_out[k] = t;
  }
  double loss, w[4];
  if (fin.per_axis_mean) {
    loss = 0.0;
    for (int k = 0; k < 3; ++k) {
      w[k] = 1.0 / (3.0 * S[4 + k]);
      loss += S[k] * w[k];
    }
    w[3] = 0.0;
  } else {
    const double den = fin.count ? (double)*fin.count : S[3];
    const double Ssum = (S[0] + S[1]) + S[2];
    loss = Ssum / (3.0 * den);
    w[0] = w[1] = w[2] = 1.0 / (3.0 * den);
    w[3] = fin.count ? 0.0 : -Ssum / (3.0 * den * den);     // without a mask the denominator is v.sum(): it has a derivative too
  }
  if (fin.loss_out) *fin.loss_out = (float)(fin.scale * loss) + (fin.add_in ? *fin.add_in : 0.f);
  if (fin.w_out)
    for (int k = 0; k < 4; ++k) fin.w_out[k] = (float)(fin.scale * w[k]);
  *counter = 0u;
}

// grad[j] (+)= sum over axes of w[axis] * (sign(v[j] - v[j-1]) [pair below valid] - sign(v[j+1] - v[j]) [pair above valid]) + w[3]
template <bool CH_LAST, bool ACCUMULATE>
__global__ __launch_bounds__(FGS_BLOCK) void k_tv_loss_grad(const float *__restrict__ v, TvGrid g, const float *__restrict__ w,
                                                            const float *__restrict__ upstream, float *__restrict__ grad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.n) return;
  const float up = upstream ? *upstream : 1.f;
  const float wx = w[0] * up, wy = w[1] * up, wz = w[2] * up, w0 = w[3] * up;
  int64_t c, x, y, z;
  tv_decode<CH_LAST>(g, i, c, x, y, z);
  const int64_t o = c * g.d.sC + x * g.d.sX + y * g.d.sY + z * g.d.sZ;
  const int64_t mo = (x * g.d.Y + y) * g.d.Z + z;
  float r = w0;
  if (!g.mask || g.mask[mo]) {
    const float a = v[o];
    const int64_t mx = g.d.Y * g.d.Z, my = g.d.Z;
    if (x > 0 && (!g.mask || g.mask[mo - mx])) r += wx * sgnf(a - v[o - g.d.sX]);
    if (x + 1 < g.d.X && (!g.mask || g.mask[mo + mx])) r -= wx * sgnf(v[o + g.d.sX] - a);
    if (y > 0 && (!g.mask || g.mask[mo - my])) r += wy * sgnf(a - v[o - g.d.sY]);
    if (y + 1 < g.d.Y && (!g.mask || g.mask[mo + my])) r -= wy * sgnf(v[o + g.d.sY] - a);
    if (z > 0 && (!g.mask || g.mask[mo - 1])) r += wz * sgnf(a - v[o - g.d.sZ]);
    if (z + 1 < g.d.Z && (!g.mask || g.mask[mo + 1])) r -= wz * sgnf(v[o + g.d.sZ] - a);
  }
  if (ACCUMULATE) grad[o] += r;
  else grad[o] = r;
}

int tv_grid(const char *who, int64_t C, int64_t X, int64_t Y, int64_t Z, int64_t sC, int64_t sX, int64_t sY, int64_t sZ,
            const unsigned char *mask, TvGrid *g, bool *ch_last) {
  FGS_REQUIRE(C > 0 && X > 0 && Y > 0 && Z > 0, FGS_E_INVALID, "%s: empty grid", who);
  const int64_t N = C * X * Y * Z;
  FGS_REQUIRE(N < FGS_MAX_ELEMS, FGS_E_RANGE, "%s: %lld elements", who, (long long)N);
  const bool first = (sZ == 1 && sY == Z && sX == Y * Z && (C == 1 || sC == X * Y * Z));
  const bool last = (sC == 1 && sZ == C && sY == Z * C && sX == Y * Z * C);
  FGS_REQUIRE(first || last, FGS_E_INVALID, "%s: strides (%lld,%lld,%lld,%lld) are neither channel-first nor channel-last dense",
              who, (long long)sC, (long long)sX, (long long)sY, (long long)sZ);
  g->d = GridDesc{C, X, Y, Z, sC, sX, sY, sZ};
  g->mask = mask;
  g->n = N;
  *ch_last = last && !(first && C == 1);
  return 0;
}

}  // namespace

// Value pass (see include/fgs_hip.h).  scratch: 8-byte aligned, >= 1 + 7 * FGS_TV_VALUE_WGS doubles, first word zero when first
// handed in (left zero).
constexpr unsigned FGS_TV_VALUE_WGS = 1024;     // (4 per CU: the pass is a chain of dependent cache hits per element -- latency, not bytes)
FGS_API int64_t fgs_tv_loss_scratch_doubles(void) { return 1 + 7 * (int64_t)FGS_TV_VALUE_WGS; }

FGS_API int fgs_tv_loss_value(const float *v, const unsigned char *mask, int64_t C, int64_t X, int64_t Y, int64_t Z, int64_t sC,
                              int64_t sX, int64_t sY, int64_t sZ, const int64_t *count_dev, int per_axis_mean, double scale,
                              const float *add_in_dev, double *scratch, int64_t scratch_doubles, float *loss_out, float *w_out,
                              double *sums_out, fgs_stream_t stream) {
  TvGrid g;
  bool ch_last;
  if (int e = tv_grid("fgs_tv_loss_value", C, X, Y, Z, sC, sX, sY, sZ, mask, &g, &ch_last)) return e;
  FGS_REQUIRE(v && scratch && (loss_out || w_out || sums_out), FGS_E_INVALID, "fgs_tv_loss_value: null pointer");
  FGS_REQUIRE(per_axis_mean || (count_dev != nullptr) == (mask != nullptr), FGS_E_INVALID,
              "fgs_tv_loss_value: the denominator of the masked form is count_dev, of the unmasked form sum(v)");
  const int64_t want = (g.n + FGS_BLOCK - 1) / FGS_BLOCK;
  const unsigned blocks = (unsigned)(want < FGS_TV_VALUE_WGS ? want : FGS_TV_VALUE_WGS);
  FGS_REQUIRE(scratch_doubles >= 1 + 7 * (int64_t)blocks, FGS_E_INVALID, "fgs_tv_loss_value: scratch %lld doubles, need %lld",
              (long long)scratch_doubles, (long long)(1 + 7 * (int64_t)blocks));
  const TvFin fin{count_dev, per_axis_mean, scale, add_in_dev, loss_out, w_out, sums_out};
  unsigned *counter = reinterpret_cast<unsigned *>(scratch);
  if (ch_last) hipLaunchKernelGGL(k_tv_loss_value<true>, dim3(blocks), dim3(FGS_BLOCK), 0, fgs_s(stream), v, g, scratch + 1, counter, fin);
  else hipLaunchKernelGGL(k_tv_loss_value<false>, dim3(blocks), dim3(FGS_BLOCK), 0, fgs_s(stream), v, g, scratch + 1, counter, fin);
  FGS_LAUNCH_OK("fgs_tv_loss_value");
  return 0;
}

// Gradient pass.  w: 4 device floats {w_x, w_y, w_z, w_0} (what the value pass left in w_out), upstream_dev: device scalar that
// multiplies them (d total / d term) or NULL: grad[j] = (accumulate ? grad[j] : 0) + sum_axis w_axis * (signed
// pair terms of element j) + w_0   (w_0: the derivative through a v.sum() denominator; 0 otherwise).  grad has v's strides.
FGS_API int fgs_tv_loss_grad(const float *v, const unsigned char *mask, int64_t C, int64_t X, int64_t Y, int64_t Z, int64_t sC,
                             int64_t sX, int64_t sY, int64_t sZ, const float *w, const float *upstream_dev, float *grad,
                             int accumulate, fgs_stream_t stream) {
  TvGrid g;
  bool ch_last;
  if (int e = tv_grid("fgs_tv_loss_grad", C, X, Y, Z, sC, sX, sY, sZ, mask, &g, &ch_last)) return e;
  FGS_REQUIRE(v && w && grad, FGS_E_INVALID, "fgs_tv_loss_grad: null pointer");
  const dim3 grid(fgs_blocks(g.n)), blk(FGS_BLOCK);
  hipStream_t st = fgs_s(stream);
  if (ch_last) {
    if (accumulate) hipLaunchKernelGGL((k_tv_loss_grad<true, true>), grid, blk, 0, st, v, g, w, upstream_dev, grad);
    else hipLaunchKernelGGL((k_tv_loss_grad<true, false>), grid, blk, 0, st, v, g, w, upstream_dev, grad);
  } else {
    if (accumulate) hipLaunchKernelGGL((k_tv_loss_grad<false, true>), grid, blk, 0, st, v, g, w, upstream_dev, grad);
    else hipLaunchKernelGGL((k_tv_loss_grad<false, false>), grid, blk, 0, st, v, g, w, upstream_dev, grad);
  }
  FGS_LAUNCH_OK("fgs_tv_loss_grad");
  return 0;
}
