// trilerp.hip -- trilinear dense-grid lookup, forward and scatter-add backward.
// Replaces F.grid_sample(grid[1,C,X,Y,Z], ind_norm[1,1,1,M,3], 'bilinear', align_corners=True,
// padding_mode='zeros') as used by DenseGrid.forward (model/grid.py:49-59), nerf.grid_sampler
// (model/nerf.py:654-657) and MaskCache.forward (model/nerf.py:1203-1209).
//
// Layout: the reference keeps [1,C,X,Y,Z] channel-first, so the 8*C reads of one sample are
// X*Y*Z*4 bytes apart.  This build stores multi-channel grids channel-last (sC == 1): the C values
// of one corner are one contiguous 4*C-byte run and neighbouring lanes (channels of the same
// sample) coalesce.  Kernels take element strides and serve both layouts.
#include "fgs_common.h"

namespace {

struct PointIdx {
  float fx, fy, fz;
};

__device__ __forceinline__ PointIdx point_to_index(const float *__restrict__ pts, int64_t m,
                                                   const float *__restrict__ lo, const float *__restrict__ hi,
                                                   const GridDesc &d) {
  PointIdx p;
  p.fx = fgs_world_to_index(pts[3 * m + 0], lo[0], hi[0], (int)d.X);
  p.fy = fgs_world_to_index(pts[3 * m + 1], lo[1], hi[1], (int)d.Y);
  p.fz = fgs_world_to_index(pts[3 * m + 2], lo[2], hi[2], (int)d.Z);
  return p;
}

// CHANNEL_FAST: thread = (sample, channel) with channel fastest  -> coalesced for channel-last grids.
// otherwise   : thread = sample, loop over channels              -> channel-first grids (and C == 1).
template <bool CHANNEL_FAST>
__global__ __launch_bounds__(FGS_BLOCK) void k_trilerp_fwd(const float *__restrict__ grid, GridDesc d,
                                                           const float *__restrict__ xyz_min,
                                                           const float *__restrict__ xyz_max,
                                                           const float *__restrict__ pts, int64_t M,
                                                           float *__restrict__ out) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (CHANNEL_FAST) {
    if (tid >= M * d.C) return;
    const int64_t m = tid / d.C, c = tid - m * d.C;
    const PointIdx p = point_to_index(pts, m, xyz_min, xyz_max, d);
    const TriCorners t = fgs_tri_setup(p.fx, p.fy, p.fz);
    out[tid] = fgs_tri_sample(grid, d, c, t);
  } else {
    if (tid >= M) return;
    const PointIdx p = point_to_index(pts, tid, xyz_min, xyz_max, d);
    const TriCorners t = fgs_tri_setup(p.fx, p.fy, p.fz);
    for (int64_t c = 0; c < d.C; ++c) out[tid * d.C + c] = fgs_tri_sample(grid, d, c, t);
  }
}

template <bool CHANNEL_FAST>
__global__ __launch_bounds__(FGS_BLOCK) void k_trilerp_bwd(float *__restrict__ grad_grid, GridDesc d,
                                                           const float *__restrict__ xyz_min,
                                                           const float *__restrict__ xyz_max,
                                                           const float *__restrict__ pts, int64_t M,
                                                           const float *__restrict__ grad_out) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (CHANNEL_FAST) {
    if (tid >= M * d.C) return;
    const int64_t m = tid / d.C, c = tid - m * d.C;
    const PointIdx p = point_to_index(pts, m, xyz_min, xyz_max, d);
    const TriCorners t = fgs_tri_setup(p.fx, p.fy, p.fz);
    fgs_tri_scatter(grad_grid, d, c, t, grad_out[tid]);
  } else {
    if (tid >= M) return;
    const PointIdx p = point_to_index(pts, tid, xyz_min, xyz_max, d);
    const TriCorners t = fgs_tri_setup(p.fx, p.fy, p.fz);
    for (int64_t c = 0; c < d.C; ++c) fgs_tri_scatter(grad_grid, d, c, t, grad_out[tid * d.C + c]);
  }
}

// ---- axis taps of a 1-channel grid (model/nerf.py:597-637 `sample_sdfs`) -------------------------------
// tap t in [0, 6K): pair = t / K (0,1: -z,+z   2,3: -y,+y   4,5: -x,+x  -- the reference offsets act on the
// zyx-flipped index), k = t % K.  The tap point is ind + sign*displace[k] on that axis, clamped to the volume
// in INDEX space, then pushed through the reference's index -> [-1,1] -> index round trip before the lookup.
struct TapPoint {
  float fx, fy, fz;   // index coordinates actually sampled
  float clamped;      // the clamped (pre round trip) coordinate on the displaced axis, for `diff`
};

__device__ __forceinline__ TapPoint tap_point(const PointIdx &p, const GridDesc &d, int pair, float disp) {
  const int axis_zyx = pair >> 1;              // 0 -> z, 1 -> y, 2 -> x
  const float off = (pair & 1) ? disp : -disp; // offset row (-1 | +1) * displace
  float iz = p.fz, iy = p.fy, ix = p.fx;
  if (axis_zyx == 0) iz = iz + off; else if (axis_zyx == 1) iy = iy + off; else ix = ix + off;
  iz = fminf(fmaxf(iz, 0.f), (float)(d.Z - 1));
  iy = fminf(fmaxf(iy, 0.f), (float)(d.Y - 1));
  ix = fminf(fmaxf(ix, 0.f), (float)(d.X - 1));
  TapPoint t;
  t.clamped = (axis_zyx == 0) ? iz : (axis_zyx == 1 ? iy : ix);
  t.fx = fgs_index_roundtrip(ix, (int)d.X);
  t.fy = fgs_index_roundtrip(iy, (int)d.Y);
  t.fz = fgs_index_roundtrip(iz, (int)d.Z);
  return t;
}

constexpr int MAX_DISPLACE = 8;
struct DisplaceList {
  int K;
  float v[MAX_DISPLACE];
};

// feat[m, pair*K + k] ; diff[m, axis_zyx*K + k] = clamped(+) - clamped(-)  (model/nerf.py:621-623)
__global__ __launch_bounds__(FGS_BLOCK) void k_taps_fwd(const float *__restrict__ grid, GridDesc d,
                                                        const float *__restrict__ xyz_min,
                                                        const float *__restrict__ xyz_max,
                                                        const float *__restrict__ pts, int64_t M, DisplaceList dl,
                                                        float *__restrict__ feat, float *__restrict__ diff) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int T = 6 * dl.K;
  if (tid >= M * T) return;
  const int64_t m = tid / T;
  const int t = (int)(tid - m * T);
  const int pair = t / dl.K, k = t - pair * dl.K;
  const PointIdx p = point_to_index(pts, m, xyz_min, xyz_max, d);
  const TapPoint tp = tap_point(p, d, pair, dl.v[k]);
  const TriCorners tc = fgs_tri_setup(tp.fx, tp.fy, tp.fz);
  feat[tid] = fgs_tri_sample(grid, d, 0, tc);
  if (diff && (pair & 1)) {
    const TapPoint tm = tap_point(p, d, pair - 1, dl.v[k]);
    diff[m * (3 * dl.K) + (pair >> 1) * dl.K + k] = tp.clamped - tm.clamped;
  }
}

__global__ __launch_bounds__(FGS_BLOCK) void k_taps_bwd(float *__restrict__ grad_grid, GridDesc d,
                                                        const float *__restrict__ xyz_min,
                                                        const float *__restrict__ xyz_max,
                                                        const float *__restrict__ pts, int64_t M, DisplaceList dl,
                                                        const float *__restrict__ grad_feat) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int T = 6 * dl.K;
  if (tid >= M * T) return;
  const int64_t m = tid / T;
  const int t = (int)(tid - m * T);
  const int pair = t / dl.K, k = t - pair * dl.K;
  const float g = grad_feat[tid];
  if (g == 0.f) return;
  const PointIdx p = point_to_index(pts, m, xyz_min, xyz_max, d);
  const TapPoint tp = tap_point(p, d, pair, dl.v[k]);
  const TriCorners tc = fgs_tri_setup(tp.fx, tp.fy, tp.fz);
  fgs_tri_scatter(grad_grid, d, 0, tc, g);
}

int check_grid(const char *who, int64_t C, int64_t X, int64_t Y, int64_t Z, int64_t M) {
  if (C <= 0 || X <= 0 || Y <= 0 || Z <= 0) return fgs_set_error(FGS_E_INVALID, "%s: empty grid", who);
  if (X >= (1 << 30) || Y >= (1 << 30) || Z >= (1 << 30) || C * X * Y * Z >= FGS_MAX_ELEMS || M < 0 ||
      M * C >= FGS_MAX_ELEMS)
    return fgs_set_error(FGS_E_RANGE, "%s: size out of range", who);
  return 0;
}

}  // namespace

FGS_API int fgs_trilerp_fwd(const float *grid, int64_t C, int64_t X, int64_t Y, int64_t Z, int64_t sC, int64_t sX,
                            int64_t sY, int64_t sZ, const float *xyz_min, const float *xyz_max, const float *pts,
                            int64_t M, float *out, fgs_stream_t stream) {
  if (int e = check_grid("fgs_trilerp_fwd", C, X, Y, Z, M)) return e;
  if (M == 0) return 0;
  FGS_REQUIRE(grid && xyz_min && xyz_max && pts && out, FGS_E_INVALID, "fgs_trilerp_fwd: null pointer");
  const GridDesc d{C, X, Y, Z, sC, sX, sY, sZ};
  if (C > 1 && sC == 1) {
    hipLaunchKernelGGL(k_trilerp_fwd<true>, dim3(fgs_blocks(M * C)), dim3(FGS_BLOCK), 0, fgs_s(stream), grid, d, xyz_min,
                       xyz_max, pts, M, out);
  } else {
    hipLaunchKernelGGL(k_trilerp_fwd<false>, dim3(fgs_blocks(M)), dim3(FGS_BLOCK), 0, fgs_s(stream), grid, d, xyz_min,
                       xyz_max, pts, M, out);
  }
  FGS_LAUNCH_OK("fgs_trilerp_fwd");
  return 0;
}

FGS_API int fgs_trilerp_bwd(float *grad_grid, int64_t C, int64_t X, int64_t Y, int64_t Z, int64_t sC, int64_t sX,
                            int64_t sY, int64_t sZ, const float *xyz_min, const float *xyz_max, const float *pts,
                            int64_t M, const float *grad_out, fgs_stream_t stream) {
  if (int e = check_grid("fgs_trilerp_bwd", C, X, Y, Z, M)) return e;
  if (M == 0) return 0;
  FGS_REQUIRE(grad_grid && xyz_min && xyz_max && pts && grad_out, FGS_E_INVALID, "fgs_trilerp_bwd: null pointer");
  const GridDesc d{C, X, Y, Z, sC, sX, sY, sZ};
  if (C > 1 && sC == 1) {
    hipLaunchKernelGGL(k_trilerp_bwd<true>, dim3(fgs_blocks(M * C)), dim3(FGS_BLOCK), 0, fgs_s(stream), grad_grid, d,
                       xyz_min, xyz_max, pts, M, grad_out);
  } else {
    hipLaunchKernelGGL(k_trilerp_bwd<false>, dim3(fgs_blocks(M)), dim3(FGS_BLOCK), 0, fgs_s(stream), grad_grid, d, xyz_min,
                       xyz_max, pts, M, grad_out);
  }
  FGS_LAUNCH_OK("fgs_trilerp_bwd");
  return 0;
}

static int make_displace(const char *who, const float *displace, int K, DisplaceList *dl) {
  if (K <= 0 || K > MAX_DISPLACE) return fgs_set_error(FGS_E_RANGE, "%s: K=%d displacements (1..%d supported)", who, K, MAX_DISPLACE);
  if (!displace) return fgs_set_error(FGS_E_INVALID, "%s: null displace", who);
  dl->K = K;
  for (int i = 0; i < K; ++i) dl->v[i] = displace[i];
  return 0;
}

FGS_API int fgs_sdf_taps_fwd(const float *grid, int64_t X, int64_t Y, int64_t Z, const float *xyz_min,
                             const float *xyz_max, const float *pts, int64_t M, const float *displace_host, int K,
                             float *feat, float *diff, fgs_stream_t stream) {
  if (int e = check_grid("fgs_sdf_taps_fwd", 1, X, Y, Z, M * 6 * (K > 0 ? K : 1))) return e;
  DisplaceList dl;
  if (int e = make_displace("fgs_sdf_taps_fwd", displace_host, K, &dl)) return e;
  if (M == 0) return 0;
  FGS_REQUIRE(grid && xyz_min && xyz_max && pts && feat, FGS_E_INVALID, "fgs_sdf_taps_fwd: null pointer");
  const GridDesc d{1, X, Y, Z, X * Y * Z, Y * Z, Z, 1};
  hipLaunchKernelGGL(k_taps_fwd, dim3(fgs_blocks(M * 6 * K)), dim3(FGS_BLOCK), 0, fgs_s(stream), grid, d, xyz_min, xyz_max,
                     pts, M, dl, feat, diff);
  FGS_LAUNCH_OK("fgs_sdf_taps_fwd");
  return 0;
}

FGS_API int fgs_sdf_taps_bwd(float *grad_grid, int64_t X, int64_t Y, int64_t Z, const float *xyz_min,
                             const float *xyz_max, const float *pts, int64_t M, const float *displace_host, int K,
                             const float *grad_feat, fgs_stream_t stream) {
  if (int e = check_grid("fgs_sdf_taps_bwd", 1, X, Y, Z, M * 6 * (K > 0 ? K : 1))) return e;
  DisplaceList dl;
  if (int e = make_displace("fgs_sdf_taps_bwd", displace_host, K, &dl)) return e;
  if (M == 0) return 0;
  FGS_REQUIRE(grad_grid && xyz_min && xyz_max && pts && grad_feat, FGS_E_INVALID, "fgs_sdf_taps_bwd: null pointer");
  const GridDesc d{1, X, Y, Z, X * Y * Z, Y * Z, Z, 1};
  hipLaunchKernelGGL(k_taps_bwd, dim3(fgs_blocks(M * 6 * K)), dim3(FGS_BLOCK), 0, fgs_s(stream), grad_grid, d, xyz_min,
                     xyz_max, pts, M, dl, grad_feat);
  FGS_LAUNCH_OK("fgs_sdf_taps_bwd");
  return 0;
}
