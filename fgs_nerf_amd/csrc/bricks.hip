// bricks.hip -- brick-sparse view of a channel-last grid gradient for the multi-GPU gradient exchange
// (fgs-nerf_amd/dist.py).  The reference has no distributed code; this serves the north_star's "dense grid replicated,
// gradients all-reduced over xGMI" with the observation that rays touch only a thin shell of voxels, so only the
// occupied 4x4x4-voxel bricks need to travel.
//
// Storage: [X][Y][Z][C] floats (DenseGrid channel-last).  A brick is 16 runs of 4*C contiguous floats (4 z-voxels x C
// channels at fixed x', y').  One wavefront handles one brick; lanes walk its 16*C float4s.
//   fgs_brick_flags   : flags[b] = 1 if any element of brick b is non-zero (one streaming pass over the gradient)
//   fgs_brick_gather  : buf[i] (64*C floats, order x',y',z',c) = brick idx[i]
//   fgs_brick_scatter : brick idx[i] = buf[i] * scale
#include "fgs_taps.h"

namespace {

struct BrickGrid {
  int C, X, Y, Z, nbx, nby, nbz;
};

// float4 q of brick (bx,by,bz): run = q / C (x' = run / 4, y' = run % 4), within-run float4 = q % C
__device__ __forceinline__ int64_t brick_f4_offset(const BrickGrid &g, int bx, int by, int bz, int q) {
  const int run = q / g.C, w4 = q - run * g.C;
  const int x = bx * 4 + (run >> 2), y = by * 4 + (run & 3);
  return (((int64_t)x * g.Y + y) * g.Z + bz * 4) * g.C + (int64_t)w4 * 4;
}

__global__ __launch_bounds__(FGS_BLOCK) void k_brick_flags(const float *__restrict__ grad, BrickGrid g,
                                                           int *__restrict__ flags) {
  const int64_t b = (int64_t)blockIdx.x * (FGS_BLOCK / FGS_WAVE) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t total = (int64_t)g.nbx * g.nby * g.nbz;
  if (b >= total) return;
  const int bz = (int)(b % g.nbz), by = (int)((b / g.nbz) % g.nby), bx = (int)(b / ((int64_t)g.nbz * g.nby));
  bool nz = false;
  for (int q = lane; q < 16 * g.C; q += FGS_WAVE) {
    const float4 v = *reinterpret_cast<const float4 *>(grad + brick_f4_offset(g, bx, by, bz, q));
    nz |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
  }
  const unsigned long long any = __ballot(nz);
  if (lane == 0) flags[b] = any ? 1 : 0;
}

template <bool SCATTER>
__global__ __launch_bounds__(FGS_BLOCK) void k_brick_copy(float *__restrict__ grad, BrickGrid g,
                                                          const int64_t *__restrict__ idx, int64_t n,
                                                          float *__restrict__ buf, float scale) {
  const int64_t i = (int64_t)blockIdx.x * (FGS_BLOCK / FGS_WAVE) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const int64_t b = idx[i];
  const int bz = (int)(b % g.nbz), by = (int)((b / g.nbz) % g.nby), bx = (int)(b / ((int64_t)g.nbz * g.nby));
  float4 *dst = reinterpret_cast<float4 *>(buf + i * 64 * g.C);
  for (int q = lane; q < 16 * g.C; q += FGS_WAVE) {
    float4 *gp = reinterpret_cast<float4 *>(grad + brick_f4_offset(g, bx, by, bz, q));
    if (SCATTER) {
      float4 v = dst[q];
      v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
      *gp = v;
    } else {
      dst[q] = *gp;
    }
  }
}

// The same copies with the brick count in DEVICE memory (the captured multi-GPU step: nothing that sizes the exchange may pass
// through the host).  The launch covers `capacity` rows; rows at or beyond min(*count_dev, capacity) are not part of the
// exchange: the gather writes zeros there (the fixed-size collective then sums zeros, never stale or non-finite leftovers),
// the scatter skips them.
template <bool SCATTER>
__global__ __launch_bounds__(FGS_BLOCK) void k_brick_copy_dev(float *__restrict__ grad, BrickGrid g,
                                                              const int64_t *__restrict__ idx,
                                                              const int64_t *__restrict__ count_dev, int64_t capacity,
                                                              float *__restrict__ buf, float scale) {
  __builtin_amdgcn_s_setprio(2);     // runs beside k_mlp_wgrad (the exchange branch of the captured step): memory-bound, its
                                     // few vector instructions go first (221 -> us without: starved of issue slots)
  const int64_t i = (int64_t)blockIdx.x * (FGS_BLOCK / FGS_WAVE) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= capacity) return;
  const int64_t n = min(*count_dev, capacity);
  float4 *row = reinterpret_cast<float4 *>(buf + i * 64 * g.C);
  if (i >= n) {
    if (!SCATTER)
      for (int q = lane; q < 16 * g.C; q += FGS_WAVE) row[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const int64_t b = idx[i];
  const int bz = (int)(b % g.nbz), by = (int)((b / g.nbz) % g.nby), bx = (int)(b / ((int64_t)g.nbz * g.nby));
  for (int q = lane; q < 16 * g.C; q += FGS_WAVE) {
    float4 *gp = reinterpret_cast<float4 *>(grad + brick_f4_offset(g, bx, by, bz, q));
    if (SCATTER) {
      float4 v = row[q];
      v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
      *gp = v;
    } else {
      row[q] = *gp;
    }
  }
}

// Guard of a device-counted brick list against the capacity of its exchange buffer.  An exchange that did not fit cannot be
// repaired inside a captured step (every rank must issue the same fixed-size collective), and it leaves gradient behind in
// the persistent buffer's unlisted bricks: from then on EVERY step's update is skipped (sticky[0]) until the host has looked
// (flags[0]), cleaned up and chosen a larger capacity.  The count is all-reduced before it gets here (the union's), so all
// ranks take the same decision in the same step.  flags as in fgs_count_guard: [0] sticky "something overflowed", [1] "the
// optimizer kernels skip this step".
// peer_skip (optional): the MAX over ranks of every rank's own "skip this step" flag (its survivor list did not fit), which
// travelled with the occupancy all-reduce: a step one rank must skip is skipped by all, or the replicas would drift apart.
__global__ void k_brick_count_guard(int64_t *__restrict__ count, int64_t capacity, int *__restrict__ flags,
                                    int *__restrict__ sticky, const int *__restrict__ peer_skip) {
  const int64_t n = *count;
  if (n > capacity || *sticky) {
    *sticky = 1;
    flags[0] = 1;
    flags[1] = 1;
  }
  if (peer_skip && *peer_skip) {
    flags[0] = 1;
    flags[1] = 1;
  }
  if (n > capacity) *count = capacity;
}

// flags[b] = 1 for every brick holding one of the 8 trilinear corners of a sample point: the bricks a DenseGrid backward
// (fgs_trilerp_bwd / k_feat_k0_bwd) can write for these points -- known as soon as the forward has its survivor list,
// i.e. ~2 ms before the gradient itself exists.  Plain stores of the same value: the race is benign.
__global__ __launch_bounds__(FGS_BLOCK) void k_brick_flags_pts(const float *__restrict__ pts, int64_t M,
                                                               const int64_t *__restrict__ m_dev, SceneGeom sg,
                                                               BrickGrid g, int *__restrict__ flags) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= fgs_rows(M, m_dev)) return;
  const GridDesc d = fgs_sdf_desc(sg);
  const PointIdx p = fgs_point_to_index(pts[3 * m], pts[3 * m + 1], pts[3 * m + 2], sg.lo, sg.hi, d);
  const TriCorners t = fgs_tri_setup(p.fx, p.fy, p.fz);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int x = t.x0 + (k >> 2), y = t.y0 + ((k >> 1) & 1), z = t.z0 + (k & 1);
    if (fgs_in(x, g.X) && fgs_in(y, g.Y) && fgs_in(z, g.Z)) flags[((int64_t)(x >> 2) * g.nby + (y >> 2)) * g.nbz + (z >> 2)] = 1;
  }
}

// Ordered compaction of the set flags into idx[0..count) on the device (one workgroup; the brick count of a 320^3 grid
// is 512 K).  No host involvement: the count is fetched asynchronously by whoever sizes the exchange buffer.
// One workgroup per tile of 1024 flags.  A tile's base offset is the number of set flags in all EARLIER tiles, which the
// workgroup counts itself (coalesced re-read of the flags in front of it: 16 KB on average for 160^3, from L2) -- no
// second launch, no atomics, and the output order stays ascending, which every rank's packing of the exchange relies
// on.  Inside the tile: ballot / popcount ranks and 16 wave counts in LDS.  (History: per-thread sequential scan + a
// 1024-wide Hillis-Steele scan 92 us; one workgroup walking the 63 tiles in turn 46 us; this form ~6 us.)
constexpr int COMPACT_THREADS = 1024;
__global__ __launch_bounds__(COMPACT_THREADS) void k_brick_compact(const int *__restrict__ flags, int64_t total,
                                                                   int64_t *__restrict__ idx, int64_t *__restrict__ count) {
  __shared__ int wave_cnt[COMPACT_THREADS / FGS_WAVE];
  __shared__ int before[COMPACT_THREADS / FGS_WAVE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int64_t tile0 = (int64_t)blockIdx.x * COMPACT_THREADS;
  // set flags in front of this tile
  int mine = 0;
  for (int64_t i = t; i < tile0; i += COMPACT_THREADS) mine += flags[i] != 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
  const int64_t i = tile0 + t;
  const bool f = i < total && flags[i] != 0;
  const unsigned long long bal = __ballot(f);
  if (lane == 0) {
    before[wave] = mine;
    wave_cnt[wave] = __popcll(bal);
  }
  __syncthreads();
  int64_t base = 0;
  int woff = 0, tile_total = 0;
#pragma unroll
  for (int w = 0; w < COMPACT_THREADS / FGS_WAVE; ++w) {
    base += before[w];
    const int c = wave_cnt[w];
    woff += (w < wave) ? c : 0;
    tile_total += c;
  }
  if (f) idx[base + woff + __popcll(bal & lt_mask)] = i;
  if (t == 0 && tile0 + COMPACT_THREADS >= total) *count = base + tile_total;     // the last tile knows the total
}

int make_grid_any(const char *who, int C, int X, int Y, int Z, BrickGrid *g) {
  if (C <= 0 || X <= 0 || Y <= 0 || Z <= 0) return fgs_set_error(FGS_E_INVALID, "%s: bad grid %dx%dx%dx%d", who, X, Y, Z, C);
  g->C = C; g->X = X; g->Y = Y; g->Z = Z; g->nbx = (X + 3) / 4; g->nby = (Y + 3) / 4; g->nbz = (Z + 3) / 4;
  return 0;
}

int make_grid(const char *who, int C, int X, int Y, int Z, BrickGrid *g) {
  if (C <= 0 || X <= 0 || Y <= 0 || Z <= 0 || (X & 3) || (Y & 3) || (Z & 3))
    return fgs_set_error(FGS_E_INVALID, "%s: grid %dx%dx%dx%d must have sides that are multiples of 4", who, X, Y, Z, C);
  g->C = C; g->X = X; g->Y = Y; g->Z = Z; g->nbx = X / 4; g->nby = Y / 4; g->nbz = Z / 4;
  return 0;
}

}  // namespace

FGS_API int fgs_brick_flags(const float *grad, int C, int X, int Y, int Z, int *flags, fgs_stream_t stream) {
  BrickGrid g;
  if (int e = make_grid("fgs_brick_flags", C, X, Y, Z, &g)) return e;
  FGS_REQUIRE(grad && flags, FGS_E_INVALID, "fgs_brick_flags: null pointer");
  const int64_t total = (int64_t)g.nbx * g.nby * g.nbz;
  hipLaunchKernelGGL(k_brick_flags, dim3(fgs_blocks(total * FGS_WAVE)), dim3(FGS_BLOCK), 0, fgs_s(stream), grad, g, flags);
  FGS_LAUNCH_OK("fgs_brick_flags");
  return 0;
}

FGS_API int fgs_brick_gather(const float *grad, int C, int X, int Y, int Z, const int64_t *idx, int64_t n, float *buf,
                             fgs_stream_t stream) {
  BrickGrid g;
  if (int e = make_grid("fgs_brick_gather", C, X, Y, Z, &g)) return e;
  if (n == 0) return 0;
  FGS_REQUIRE(grad && idx && buf && n > 0, FGS_E_INVALID, "fgs_brick_gather: bad arguments");
  hipLaunchKernelGGL(k_brick_copy<false>, dim3(fgs_blocks(n * FGS_WAVE)), dim3(FGS_BLOCK), 0, fgs_s(stream),
                     const_cast<float *>(grad), g, idx, n, buf, 1.f);
  FGS_LAUNCH_OK("fgs_brick_gather");
  return 0;
}

FGS_API int fgs_brick_scatter(float *grad, int C, int X, int Y, int Z, const int64_t *idx, int64_t n, const float *buf,
                              float scale, fgs_stream_t stream) {
  BrickGrid g;
  if (int e = make_grid("fgs_brick_scatter", C, X, Y, Z, &g)) return e;
  if (n == 0) return 0;
  FGS_REQUIRE(grad && idx && buf && n > 0, FGS_E_INVALID, "fgs_brick_scatter: bad arguments");
  hipLaunchKernelGGL(k_brick_copy<true>, dim3(fgs_blocks(n * FGS_WAVE)), dim3(FGS_BLOCK), 0, fgs_s(stream), grad, g, idx, n,
                     const_cast<float *>(buf), scale);
  FGS_LAUNCH_OK("fgs_brick_scatter");
  return 0;
}

FGS_API int fgs_brick_gather_dev(const float *grad, int C, int X, int Y, int Z, const int64_t *idx, const int64_t *count_dev,
                                 int64_t capacity, float *buf, fgs_stream_t stream) {
  BrickGrid g;
  if (int e = make_grid("fgs_brick_gather_dev", C, X, Y, Z, &g)) return e;
  FGS_REQUIRE(grad && idx && count_dev && buf && capacity > 0 && capacity <= (int64_t)g.nbx * g.nby * g.nbz, FGS_E_INVALID,
              "fgs_brick_gather_dev: null pointer or capacity=%lld outside 1..bricks", (long long)capacity);
  hipLaunchKernelGGL(k_brick_copy_dev<false>, dim3(fgs_blocks(capacity * FGS_WAVE)), dim3(FGS_BLOCK), 0, fgs_s(stream),
                     const_cast<float *>(grad), g, idx, count_dev, capacity, buf, 1.f);
  FGS_LAUNCH_OK("fgs_brick_gather_dev");
  return 0;
}

FGS_API int fgs_brick_scatter_dev(float *grad, int C, int X, int Y, int Z, const int64_t *idx, const int64_t *count_dev,
                                  int64_t capacity, const float *buf, float scale, fgs_stream_t stream) {
  BrickGrid g;
  if (int e = make_grid("fgs_brick_scatter_dev", C, X, Y, Z, &g)) return e;
  FGS_REQUIRE(grad && idx && count_dev && buf && capacity > 0 && capacity <= (int64_t)g.nbx * g.nby * g.nbz, FGS_E_INVALID,
              "fgs_brick_scatter_dev: null pointer or capacity=%lld outside 1..bricks", (long long)capacity);
  hipLaunchKernelGGL(k_brick_copy_dev<true>, dim3(fgs_blocks(capacity * FGS_WAVE)), dim3(FGS_BLOCK), 0, fgs_s(stream), grad, g,
                     idx, count_dev, capacity, const_cast<float *>(buf), scale);
  FGS_LAUNCH_OK("fgs_brick_scatter_dev");
  return 0;
}

FGS_API int fgs_brick_count_guard(int64_t *count_dev, int64_t capacity, int *flags, int *sticky, const int *peer_skip,
                                  fgs_stream_t stream) {
  FGS_REQUIRE(count_dev && flags && sticky && capacity >= 0, FGS_E_INVALID, "fgs_brick_count_guard: bad argument");
  hipLaunchKernelGGL(k_brick_count_guard, dim3(1), dim3(1), 0, fgs_s(stream), count_dev, capacity, flags, sticky, peer_skip);
  FGS_LAUNCH_OK("fgs_brick_count_guard");
  return 0;
}

// Brick occupancy from sample points instead of from the gradient (see k_brick_flags_pts).  ORs into `flags`
// (caller zeroes it once per step); xyz_min/max on the host, grid [X,Y,Z] as for the trilinear kernels.
FGS_API int fgs_brick_flags_pts(const float *pts, int64_t M, const float *xyz_min_host, const float *xyz_max_host, int X,
                                int Y, int Z, int *flags, const fgs_dyn_t *dyn, fgs_stream_t stream) {
  BrickGrid g;      // sides that are not multiples of 4 are fine here: the last brick of an axis is then partial
  if (int e = make_grid_any("fgs_brick_flags_pts", 1, X, Y, Z, &g)) return e;
  FGS_REQUIRE(M >= 0 && M < ((int64_t)1 << 31), FGS_E_RANGE, "fgs_brick_flags_pts: M=%lld", (long long)M);
  if (M == 0) return 0;
  FGS_REQUIRE(pts && xyz_min_host && xyz_max_host && flags && X > 1 && Y > 1 && Z > 1, FGS_E_INVALID,
              "fgs_brick_flags_pts: bad arguments");
  SceneGeom sg;
  for (int c = 0; c < 3; ++c) { sg.lo[c] = xyz_min_host[c]; sg.hi[c] = xyz_max_host[c]; }
  sg.X = X; sg.Y = Y; sg.Z = Z; sg.voxel_size = 0.f;
  hipLaunchKernelGGL(k_brick_flags_pts, dim3(fgs_blocks(M)), dim3(FGS_BLOCK), 0, fgs_s(stream), pts, M, fgs_dyn_rows(dyn), sg, g,
                     flags);
  FGS_LAUNCH_OK("fgs_brick_flags_pts");
  return 0;
}

// idx[0..*count) = ascending indices of the non-zero flags; idx must hold `total` entries; count is a device int64.
FGS_API int fgs_brick_compact(const int *flags, int64_t total, int64_t *idx, int64_t *count, fgs_stream_t stream) {
  FGS_REQUIRE(total >= 0 && total < ((int64_t)1 << 31), FGS_E_RANGE, "fgs_brick_compact: total=%lld", (long long)total);
  FGS_REQUIRE(count && (total == 0 || (flags && idx)), FGS_E_INVALID, "fgs_brick_compact: null pointer");
  const unsigned tiles = (unsigned)((total + COMPACT_THREADS - 1) / COMPACT_THREADS);
  hipLaunchKernelGGL(k_brick_compact, dim3(tiles ? tiles : 1), dim3(COMPACT_THREADS), 0, fgs_s(stream), flags, total, idx, count);
  FGS_LAUNCH_OK("fgs_brick_compact");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// masked_adam_upd (model/cuda/adam_upd_kernel.cu:25-40) restricted to a LIST of bricks, and self-cleaning: the update of
// the multi-channel feature grid visits only the bricks the step's rays touched (the survivor points' trilinear corners,
// fgs_brick_flags_pts + fgs_brick_compact: ~18 % of a 160^3 grid, less at 320^3) instead of reading all of k0.grad, and
// zeroes every gradient element it consumed -- so the gradient buffer is all-zero again when the step ends and the next
// backward pass needs no 197 MB (1.57 GB at 320^3) zero fill.  Arithmetic per element = gridopt.hip adam_one<1>
// (skip where grad == 0): bit-identical to the dense masked update on the same gradient.
namespace {
__device__ __forceinline__ void adam_masked_one(float &p, float g, float &m, float &v, float step_size, float beta1,
                                                float beta2, float eps) {
  m = fmaf(beta1, m, (1.f - beta1) * g);
  v = fmaf(beta2, v, (1.f - beta2) * g * g);
  p -= (step_size * m) / (sqrtf(v) + eps);
}

// idx != NULL: the listed bricks (count from count_dev or n_host).  idx == NULL: every brick whose flag is set.  flags != NULL:
// the flag of a processed brick is cleared (the occupancy buffer is self-cleaning like the gradient).  Partial bricks at the
// upper faces of a grid whose sides are not multiples of 4: rows beyond X / Y and the float4s beyond Z are skipped (C % 4
// == 0 keeps every run 16-byte aligned).
__global__ __launch_bounds__(FGS_BLOCK) void k_adam_bricks(float *__restrict__ param, float *__restrict__ grad,
                                                           float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq,
                                                           BrickGrid g, const int64_t *__restrict__ idx,
                                                           const int64_t *__restrict__ count_dev, int64_t n_host,
                                                           int *__restrict__ flags, float step_size,
                                                           const float *__restrict__ ss_dev, float beta1, float beta2,
                                                           float eps, const int *__restrict__ skip) {
  __builtin_amdgcn_s_setprio(2);     // may run beside k_mlp_wgrad (fused.py _wgrad), whose fp32 matrix instructions occupy
                                     // the vector pipe: this memory-bound kernel's few vector instructions go first
  const int64_t total = (int64_t)g.nbx * g.nby * g.nbz;
  const int64_t n = idx ? (count_dev ? min(*count_dev, total) : n_host) : total;
  const bool no_update = skip && *skip;           // an overflowed step: consume (zero) the gradient, change nothing else
  if (ss_dev) step_size = *ss_dev;
  const int lane = threadIdx.x & 63;
  const int64_t waves = (int64_t)gridDim.x * (FGS_BLOCK / FGS_WAVE);
  for (int64_t i = (int64_t)blockIdx.x * (FGS_BLOCK / FGS_WAVE) + (threadIdx.x >> 6); i < n; i += waves) {
    int64_t b = i;
    if (idx) b = idx[i];
    else if (!flags[i]) continue;                 // wave-uniform
    const int bz = (int)(b % g.nbz), by = (int)((b / g.nbz) % g.nby), bx = (int)(b / ((int64_t)g.nbz * g.nby));
    const int zrun4 = min(4, g.Z - bz * 4) * g.C / 4;       // float4s of one (x', y') run that lie inside the grid
    for (int q = lane; q < 16 * g.C; q += FGS_WAVE) {
      const int run = q / g.C, w4 = q - run * g.C;
      if (bx * 4 + (run >> 2) >= g.X || by * 4 + (run & 3) >= g.Y || w4 >= zrun4) continue;
      const int64_t off = brick_f4_offset(g, bx, by, bz, q);
      const float4 gr = *reinterpret_cast<const float4 *>(grad + off);
      if (gr.x == 0.f && gr.y == 0.f && gr.z == 0.f && gr.w == 0.f) continue;
      if (!no_update) {
        float4 p = *reinterpret_cast<const float4 *>(param + off), m = *reinterpret_cast<const float4 *>(exp_avg + off),
               v = *reinterpret_cast<const float4 *>(exp_avg_sq + off);
        if (gr.x != 0.f) adam_masked_one(p.x, gr.x, m.x, v.x, step_size, beta1, beta2, eps);
        if (gr.y != 0.f) adam_masked_one(p.y, gr.y, m.y, v.y, step_size, beta1, beta2, eps);
        if (gr.z != 0.f) adam_masked_one(p.z, gr.z, m.z, v.z, step_size, beta1, beta2, eps);
        if (gr.w != 0.f) adam_masked_one(p.w, gr.w, m.w, v.w, step_size, beta1, beta2, eps);
        *reinterpret_cast<float4 *>(param + off) = p;
        *reinterpret_cast<float4 *>(exp_avg + off) = m;
        *reinterpret_cast<float4 *>(exp_avg_sq + off) = v;
      }
      *reinterpret_cast<float4 *>(grad + off) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (flags && lane == 0) flags[b] = 0;
  }
}
}  // namespace

FGS_API float fgs_adam_step_size(int step, float beta1, float beta2, float lr);

// param / grad / exp_avg / exp_avg_sq: channel-last [X][Y][Z][C] grids, C a multiple of 4.  Either idx (ascending brick
// indices, fgs_brick_compact; count read from count_dev when non-NULL, else n_host) or, with idx == NULL, flags (one int per
// brick, fgs_brick_flags_pts) selects the bricks; flags, when given, are cleared for the processed bricks.  step = the Adam
// step number (host scalar form) unless step_size_dev != NULL (device float, fgs_step_scalars_tick).  skip_dev as in
// fgs_adam_upd_dev, except that the gradient is consumed (zeroed) even then.
FGS_API int fgs_adam_upd_bricks(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int C, int X, int Y, int Z,
                                const int64_t *idx, const int64_t *count_dev, int64_t n_host, int *flags, int step,
                                float beta1, float beta2, float lr, float eps, const float *step_size_dev,
                                const int *skip_dev, fgs_stream_t stream) {
  BrickGrid g;
  if (int e = make_grid_any("fgs_adam_upd_bricks", C, X, Y, Z, &g)) return e;
  FGS_REQUIRE(param && grad && exp_avg && exp_avg_sq && (idx || flags) && (C & 3) == 0, FGS_E_INVALID,
              "fgs_adam_upd_bricks: null pointer, or C=%d not a multiple of 4", C);
  FGS_REQUIRE(!idx || count_dev || (n_host >= 0 && n_host <= (int64_t)g.nbx * g.nby * g.nbz), FGS_E_RANGE,
              "fgs_adam_upd_bricks: n=%lld", (long long)n_host);
  if (idx && !count_dev && n_host == 0) return 0;
  const float step_size = step_size_dev ? 0.f : fgs_adam_step_size(step, beta1, beta2, lr);     // gridopt.hip
  hipLaunchKernelGGL(k_adam_bricks, dim3(2048), dim3(FGS_BLOCK), 0, fgs_s(stream), param, grad, exp_avg, exp_avg_sq, g, idx,
                     count_dev, n_host, flags, step_size, step_size_dev, beta1, beta2, eps, skip_dev);
  FGS_LAUNCH_OK("fgs_adam_upd_bricks");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same update at VOXEL granularity.  A brick flag says "somewhere in these 64 voxels"; the scatter touches ~10 of them.
// fgs_brick_masks_pts records, per brick, which of its 64 voxels hold a trilinear corner of a survivor point (one byte per
// voxel, byte 16 x' + 4 y' + z' of the brick's 64), and fgs_adam_upd_voxels walks the recorded voxels only: per touched voxel C floats of grad, param and both
// moments instead of whole 3 KB bricks of grad -- 41 -> ~20 us at 160^3, 118 -> ~50 us at 320^3.  Same arithmetic per
// element, same self-cleaning contract (consumed gradient zeroed, mask cleared).
namespace {
__global__ __launch_bounds__(FGS_BLOCK) void k_brick_masks_pts(const float *__restrict__ pts, int64_t M,
                                                               const int64_t *__restrict__ m_dev, SceneGeom sg, BrickGrid g,
                                                               unsigned char *__restrict__ vox) {
  __builtin_amdgcn_s_setprio(2);     // may run beside k_mlp_wgrad (fused.py _wgrad), whose fp32 matrix instructions occupy
                                     // the vector pipe: this memory-bound kernel's few vector instructions go first
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= fgs_rows(M, m_dev)) return;
  const GridDesc d = fgs_sdf_desc(sg);
  const PointIdx p = fgs_point_to_index(pts[3 * m], pts[3 * m + 1], pts[3 * m + 2], sg.lo, sg.hi, d);
  const TriCorners t = fgs_tri_setup(p.fx, p.fy, p.fz);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int x = t.x0 + (k >> 2), y = t.y0 + ((k >> 1) & 1), z = t.z0 + (k & 1);
    // one BYTE per voxel, brick-major (64 bytes per brick, byte 16 x' + 4 y' + z'): plain stores of the same value, the race
    // is benign (as 64-bit masks with atomicOr -- ~50 attempts per hot word -- this pass took 26 us instead of 5)
    if (fgs_in(x, g.X) && fgs_in(y, g.Y) && fgs_in(z, g.Z))
      vox[(((int64_t)(x >> 2) * g.nby + (y >> 2)) * g.nbz + (z >> 2)) * 64 + 16 * (x & 3) + 4 * (y & 3) + (z & 3)] = 1;
  }
}

// the eight 0/1 bytes of w -> eight bits
__device__ __forceinline__ unsigned long long pack8(unsigned long long w) { return (w * 0x0102040810204080ull) >> 56; }

__global__ __launch_bounds__(FGS_BLOCK) void k_adam_voxels(float *__restrict__ param, float *__restrict__ grad,
                                                           float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq,
                                                           BrickGrid g, unsigned char *__restrict__ vox,
                                                           float step_size, const float *__restrict__ ss_dev, float beta1,
                                                           float beta2, float eps, const int *__restrict__ skip) {
  __builtin_amdgcn_s_setprio(2);     // may run beside k_mlp_wgrad (fused.py _wgrad), whose fp32 matrix instructions occupy
                                     // the vector pipe: this memory-bound kernel's few vector instructions go first
  const int64_t total = (int64_t)g.nbx * g.nby * g.nbz;
  const bool no_update = skip && *skip;
  if (ss_dev) step_size = *ss_dev;
  const int lane = threadIdx.x & 63;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  __shared__ unsigned char pos[FGS_BLOCK / FGS_WAVE][64];
  const int c4 = g.C / 4;                       // float4s per voxel
  const int vpp = FGS_WAVE / c4;                // voxels per pass of the wave
  const int slot = lane / c4, q = lane - slot * c4;
  // A wave looks at up to 64 bricks at a time, one per lane (lane j: brick base + j * waves + wave id, so the bricks of one wave
  // are scattered over the grid and the bricks of the launch's waves interleave: occupied bricks come in clusters along the
  // surface, and a wave that owned a run of neighbours would walk them one after the other while its peers idle), then walks
  // the non-empty ones of its batch (ballot), all lanes on one brick.  One vector load replaces the chain of dependent
  // per-brick flag reads that dominated the brick-flag form (8 / 64 sequential ~1 us reads per wave at 160^3 / 320^3).
  const int64_t waves = (int64_t)gridDim.x * (FGS_BLOCK / FGS_WAVE);
  const int64_t wave_id = (int64_t)blockIdx.x * (FGS_BLOCK / FGS_WAVE) + (threadIdx.x >> 6);
  for (int64_t base = 0; base < total; base += 64 * waves) {
    const int64_t mine = base + (int64_t)lane * waves + wave_id;
    unsigned long long my_mask = 0ull;
    if (mine < total) {         // this lane's brick: 64 flag bytes -> a 64-bit mask (bit = byte index)
      const uint4 *f = reinterpret_cast<const uint4 *>(vox + mine * 64);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint4 w = f[k];
        const unsigned long long lo = ((unsigned long long)w.y << 32) | w.x, hi = ((unsigned long long)w.w << 32) | w.z;
        my_mask |= (pack8(lo) | (pack8(hi) << 8)) << (16 * k);
      }
    }
    unsigned long long todo = __ballot(my_mask != 0ull);
    while (todo) {
      const int src = __builtin_ctzll(todo);
      todo &= todo - 1ull;
      const unsigned long long mask = ((unsigned long long)(unsigned)__shfl((int)(my_mask >> 32), src, 64) << 32) |
                                      (unsigned)__shfl((int)(my_mask & 0xffffffffull), src, 64);
      const int64_t b = base + (int64_t)src * waves + wave_id;
      const int n = __popcll(mask);
      const int bz = (int)(b % g.nbz), by = (int)((b / g.nbz) % g.nby), bx = (int)(b / ((int64_t)g.nbz * g.nby));
      // positions of the set bits in rank order, through LDS: lane i owns bit i (the wave's own 64 bytes; LDS operations of one
      // wave execute in order, so neither the reads below nor the next brick's writes need a barrier).  The loop it replaces --
      // "drop the s lowest set bits" -- ran max(s) iterations for the whole wave: ~120 of the ~250 vector instructions a brick
      // costs, and beside the weight-gradient launch vector instructions are what this kernel waits for.
      if ((mask >> lane) & 1ull) pos[threadIdx.x >> 6][__popcll(mask & lt_mask)] = (unsigned char)lane;
      for (int v0 = 0; v0 < n; v0 += vpp) {
        const int s = v0 + slot;
        if (slot < vpp && s < n) {
          const int bit = pos[threadIdx.x >> 6][s];
          const int x = bx * 4 + (bit >> 4), y = by * 4 + ((bit >> 2) & 3), z = bz * 4 + (bit & 3);
          const int64_t off = (((int64_t)x * g.Y + y) * g.Z + z) * g.C + 4 * q;
          const float4 gr = *reinterpret_cast<const float4 *>(grad + off);
          if (!(gr.x == 0.f && gr.y == 0.f && gr.z == 0.f && gr.w == 0.f)) {
            if (!no_update) {
              float4 p = *reinterpret_cast<const float4 *>(param + off), m = *reinterpret_cast<const float4 *>(exp_avg + off),
                     v = *reinterpret_cast<const float4 *>(exp_avg_sq + off);
              if (gr.x != 0.f) adam_masked_one(p.x, gr.x, m.x, v.x, step_size, beta1, beta2, eps);
              if (gr.y != 0.f) adam_masked_one(p.y, gr.y, m.y, v.y, step_size, beta1, beta2, eps);
              if (gr.z != 0.f) adam_masked_one(p.z, gr.z, m.z, v.z, step_size, beta1, beta2, eps);
              if (gr.w != 0.f) adam_masked_one(p.w, gr.w, m.w, v.w, step_size, beta1, beta2, eps);
              *reinterpret_cast<float4 *>(param + off) = p;
              *reinterpret_cast<float4 *>(exp_avg + off) = m;
              *reinterpret_cast<float4 *>(exp_avg_sq + off) = v;
            }
            *reinterpret_cast<float4 *>(grad + off) = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
    }
    if (my_mask != 0ull) {
      uint4 *f = reinterpret_cast<uint4 *>(vox + mine * 64);
#pragma unroll
      for (int k = 0; k < 4; ++k) f[k] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
}
}  // namespace

// masks: 64 bytes per 4x4x4-voxel brick ((X+3)/4 x (Y+3)/4 x (Z+3)/4 bricks, z fastest; 16-byte aligned); sets the bytes of the
// voxels that hold a trilinear corner of the points (the caller zeroes the buffer once; fgs_adam_upd_voxels leaves it zero).  Index mapping and
// row-count handling as fgs_brick_flags_pts.
FGS_API int fgs_brick_masks_pts(const float *pts, int64_t M, const float *xyz_min_host, const float *xyz_max_host, int X, int Y,
                                int Z, unsigned char *masks, const fgs_dyn_t *dyn, fgs_stream_t stream) {
  BrickGrid g;
  if (int e = make_grid_any("fgs_brick_masks_pts", 1, X, Y, Z, &g)) return e;
  FGS_REQUIRE(M >= 0 && M < ((int64_t)1 << 31), FGS_E_RANGE, "fgs_brick_masks_pts: M=%lld", (long long)M);
  if (M == 0) return 0;
  FGS_REQUIRE(pts && xyz_min_host && xyz_max_host && masks && X > 1 && Y > 1 && Z > 1, FGS_E_INVALID,
              "fgs_brick_masks_pts: bad arguments");
  SceneGeom sg;
  for (int c = 0; c < 3; ++c) { sg.lo[c] = xyz_min_host[c]; sg.hi[c] = xyz_max_host[c]; }
  sg.X = X; sg.Y = Y; sg.Z = Z; sg.voxel_size = 0.f;
  hipLaunchKernelGGL(k_brick_masks_pts, dim3(fgs_blocks(M)), dim3(FGS_BLOCK), 0, fgs_s(stream), pts, M, fgs_dyn_rows(dyn), sg, g,
                     masks);
  FGS_LAUNCH_OK("fgs_brick_masks_pts");
  return 0;
}

// masked_adam_upd over the voxels recorded in `masks` (fgs_brick_masks_pts); arguments otherwise as fgs_adam_upd_bricks
// (C a multiple of 4 and <= 64).
FGS_API int fgs_adam_upd_voxels(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int C, int X, int Y, int Z,
                                unsigned char *masks, int step, float beta1, float beta2, float lr, float eps,
                                const float *step_size_dev, const int *skip_dev, fgs_stream_t stream) {
  BrickGrid g;
  if (int e = make_grid_any("fgs_adam_upd_voxels", C, X, Y, Z, &g)) return e;
  FGS_REQUIRE(param && grad && exp_avg && exp_avg_sq && masks && (C & 3) == 0 && C <= 64 &&
                  (reinterpret_cast<uintptr_t>(masks) & 15) == 0, FGS_E_INVALID,
              "fgs_adam_upd_voxels: null / unaligned pointer, or C=%d not a multiple of 4 / above 64", C);
  const float step_size = step_size_dev ? 0.f : fgs_adam_step_size(step, beta1, beta2, lr);
  hipLaunchKernelGGL(k_adam_voxels, dim3(2048), dim3(FGS_BLOCK), 0, fgs_s(stream), param, grad, exp_avg, exp_avg_sq, g, masks,
                     step_size, step_size_dev, beta1, beta2, eps, skip_dev);
  FGS_LAUNCH_OK("fgs_adam_upd_voxels");
  return 0;
}

