// features.hip -- per-survivor feature assembly, the 3-wide output head and the per-ray compositing of
// nerf.forward_fine (model/nerf.py:835-903), forward and backward.
//
// Reference chain being fused (all over the M_s samples that survive both thresholds):
//   normal / l2_normalize (:835), xyz positional encoding (:837-839), k0 = DenseGrid(ray_pts) (:841, C = 12),
//   sample_sdfs with 4 displacements = 24 trilinear taps + 12 normalised finite differences (:851),
//   view-direction encoding gathered by ray_id (:871-873), torch.cat into rgb_feat [M_s,106] (:874),
//   reflection direction + its encoding (:879-881), cat with the rgbnet output (:883),
//   last refnet Linear (256 -> 3) + two sigmoids (:884-886), three segment_coo sums + background (:888-903).
// The kernels write straight into the MLP operand buffers X0 [M_s, ldx0] and Z [M_s, ldz] (Z columns
// [0, off_ref) are filled by the rgbnet's last GEMM), so no concatenation copy exists.
#include "fgs_taps.h"

namespace {

constexpr int MAXK = 8;

struct FeatLayout {
  int k0_dim, n_posfreq, n_viewfreq, n_reffreq, use_viewdir, center_sdf, use_grad_norm;
  int K;
  float disp[MAXK];
  int off_k0, off_xyz, off_view, off_sdf, off_feat, off_hgrad, off_grad, x0_cols, ldx0;
  int dx_ld, dx_gap;     // dX0 as the backward kernels read it: row pitch, and what to subtract from a column behind the k0
                         // block (fgs_dyn_t.dx0_compact: the xyz / view-direction encodings -- functions of the fixed ray inputs,
                         // no gradient needed -- are left out of dX0: 12 + 40 of 106 columns at the fine stage)
  int off_ref, z_cols, ldz;
  // coarse stages (model/nerf.py:993-1009): ONE operand buffer [k0, xyz_emb, reflect_emb, normal, viewdirs_emb];
  // off_ref / ldz then address the reflection block inside X0 and off_grad holds the NORMAL, not the raw gradient
  int coarse;
};

struct SurvArgs {
  int64_t M;  // survivors (the capacity of the buffers when m_dev is set: fgs_dyn_t.row_count)
  const int64_t *m_dev;
  const int64_t *ray_id;
  const float *pts, *sdf, *gradient, *viewdirs;  // pts/sdf/gradient per survivor, viewdirs per ray
  SceneGeom geom;
  FeatLayout L;
};

// thread id -> survivor index for the (survivor, channel) kernels: a 32-bit division whenever M * C fits (always, in
// practice); the 64-bit one is emulated and costs more than the rest of the index arithmetic together
__device__ __forceinline__ int64_t surv_of(int64_t tid, int64_t M, int64_t C) {
  if (M * C < ((int64_t)1 << 32)) return (int64_t)((unsigned)tid / (unsigned)C);
  return tid / C;
}

// consecutive survivors one thread of the k0 scatter walks (FGS_K0_BWD_RUN, default 4; 1 = one survivor per thread)
static int k0_run() {
  static const int r = [] { const int v = fgs_env_int("FGS_K0_BWD_RUN", 4); return v < 1 ? 1 : (v > 64 ? 64 : v); }();
  return r;
}

// ---------------------------------------------------------------------------------------------- k0 lookup
__global__ __launch_bounds__(FGS_BLOCK) void k_feat_k0_fwd(SurvArgs S, const float *__restrict__ k0, GridDesc kd,
                                                           float *__restrict__ X0) {
  S.M = fgs_rows(S.M, S.m_dev);
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= S.M * kd.C) return;
  const int64_t m = surv_of(tid, S.M, kd.C), c = tid - m * kd.C;
  const PointIdx p = fgs_point_to_index(S.pts[3 * m], S.pts[3 * m + 1], S.pts[3 * m + 2], S.geom.lo, S.geom.hi, kd);
  X0[m * S.L.ldx0 + S.L.off_k0 + c] = fgs_tri_sample(k0, kd, c, fgs_tri_setup(p.fx, p.fy, p.fz));
}

// One thread per (run of R consecutive survivors, z corner, channel), four accumulators -- the (x, y) corners of the current cell.
// Float atomics execute at the memory side and are priced per 64-byte request (MI355X_MICROARCH.md, global float atomics: ~20 G
// requests/s chip-wide); the kernel runs at that request rate, nothing else.  Two things cut the requests:
// * the two z corners of an (x, y) pair sit in ONE wave-instruction: their 2 x C contiguous floats (96 bytes at C = 12,
//   channel-last grid) leave as 2.25 requests on average instead of 2 x 1.5 (12 requests per survivor -> 9: 51 -> 41 us at 57 K
//   survivors beside the weight-gradient launch, 290 -> 231 us at 354 K);
// * consecutive survivors of a ray sit half a voxel apart, so about every second one falls into the cell of its predecessor: a
//   thread walks R of them and sends a cell's four sums when the cell changes.
__global__ __launch_bounds__(FGS_BLOCK) void k_feat_k0_bwd(SurvArgs S, float *__restrict__ k0_grad, GridDesc kd,
                                                           const float *__restrict__ dX0, int R) {
  __builtin_amdgcn_s_setprio(2);     // may run beside k_mlp_wgrad (fused.py _wgrad), whose fp32 matrix instructions occupy
                                     // the vector pipe: this memory-bound kernel's few vector instructions go first
  const int64_t M = fgs_rows(S.M, S.m_dev);
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t C2 = 2 * kd.C;
  const int64_t runs = (M + R - 1) / R;
  if (tid >= runs * C2) return;
  const int64_t run = surv_of(tid, runs, C2);
  const int rem = (int)(tid - run * C2);
  const int zc = rem >= (int)kd.C ? 1 : 0, c = rem - zc * (int)kd.C;
  float *const base = k0_grad + c * kd.sC;
  int kx = 0, ky = 0, kz = -1;                 // the cell whose sums are held (kz < 0: none)
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t m_end = (run + 1) * R < M ? (run + 1) * R : M;
  for (int64_t m = run * R; m <= m_end; ++m) {
    float g = 0.f;
    TriCorners t;
    t.x0 = 0; t.y0 = 0; t.z0 = -2;            // (the pass behind the last survivor only sends what is held)
    if (m < m_end) {
      g = dX0[m * S.L.dx_ld + S.L.off_k0 + c];
      const PointIdx p = fgs_point_to_index(S.pts[3 * m], S.pts[3 * m + 1], S.pts[3 * m + 2], S.geom.lo, S.geom.hi, kd);
      t = fgs_tri_setup(p.fx, p.fy, p.fz);
    }
    const int z = t.z0 + zc;
    if (kz >= 0 && (m == m_end || t.x0 != kx || t.y0 != ky || z != kz)) {
#pragma unroll
      for (int kxy = 0; kxy < 4; ++kxy) {
        const int x = kx + (kxy >> 1), y = ky + (kxy & 1);
        if (acc[kxy] != 0.f && fgs_in(x, (int)kd.X) && fgs_in(y, (int)kd.Y)) atomicAdd(base + x * kd.sX + y * kd.sY + kz * kd.sZ, acc[kxy]);
        acc[kxy] = 0.f;
      }
      kz = -1;
    }
    if (m < m_end && g != 0.f && fgs_in(z, (int)kd.Z)) {
      kx = t.x0; ky = t.y0; kz = z;
#pragma unroll
      for (int kxy = 0; kxy < 4; ++kxy) acc[kxy] += (zc ? t.w[2 * kxy + 1] : t.w[2 * kxy]) * g;      // (selects, not a run-time index)
    }
  }
}

// (Combining the corners of four consecutive survivors in a 4^3-voxel LDS brick first -- the recipe of k_feat_taps_bwd below --
// was built and measured for this scatter: 2.3 x fewer memory-side atomics, but 52.8 us instead of 45.1: with 8 x C = 96
// atomics per survivor in contiguous 48-byte runs the kernel is not bound by the atomic units, and clearing, filling and
// scanning a 3 KB brick per group costs more than it saves.  Removed.)

// ------------------------------------------------------------------------------- hierarchical SDF taps (K <= 5)
// 32 lanes per survivor (two survivors per wavefront): lane j < 6K evaluates tap j = pair*K + k; lanes j < 3K then
// form the finite difference of axis a = j / K, displacement k = j % K from the tap lanes with in-group shuffles.
struct TapLane {
  float f, clamped;
};

__device__ __forceinline__ float group_shfl(float v, int src_in_group) { return __shfl(v, src_in_group, 32); }

__global__ __launch_bounds__(FGS_BLOCK) void k_feat_taps_fwd(SurvArgs S, const float *__restrict__ sdf_grid,
                                                             float *__restrict__ X0) {
  S.M = fgs_rows(S.M, S.m_dev);
  const int64_t m = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  const int j = threadIdx.x & 31;
  const int K = S.L.K;
  const bool row_ok = m < S.M;
  const GridDesc gd = fgs_sdf_desc(S.geom);
  float f = 0.f, cl = 0.f;
  if (row_ok && j < 6 * K) {
    const PointIdx p = fgs_point_to_index(S.pts[3 * m], S.pts[3 * m + 1], S.pts[3 * m + 2], S.geom.lo, S.geom.hi, gd);
    const TapPoint tp = fgs_tap_point(p, gd, j / K, S.L.disp[j % K]);
    f = fgs_tap_value(sdf_grid, gd, tp);
    cl = tp.clamped;
    X0[m * S.L.ldx0 + S.L.off_feat + j] = f;
  }
  // finite differences (model/nerf.py:621-626): lane j < 3K -> axis a, displacement k
  const int a = (j < 3 * K) ? j / K : 0, k = (j < 3 * K) ? j % K : 0;
  const float fp = group_shfl(f, (2 * a + 1) * K + k), fm = group_shfl(f, (2 * a) * K + k);
  const float cp = group_shfl(cl, (2 * a + 1) * K + k), cm = group_shfl(cl, (2 * a) * K + k);
  float g = ((fp - fm) / (cp - cm)) / S.geom.voxel_size;
  if (S.L.use_grad_norm) {  // grad / (grad.norm(dim=axis) + 1e-5), model/nerf.py:631-632
    const float g0 = group_shfl(g, k), g1 = group_shfl(g, K + k), g2 = group_shfl(g, 2 * K + k);
    const float nrm = sqrtf((g0 * g0 + g1 * g1) + g2 * g2);
    g = g / (nrm + 1e-5f);
  }
  if (row_ok && j < 3 * K) X0[m * S.L.ldx0 + S.L.off_hgrad + j] = g;
}

constexpr int BRICK = 10;      // voxels per side of the accumulation brick of a group: floor(first centre) - 4 .. + 5
constexpr int BRICK_LO = 4;
constexpr int TAPS_GROUP = 4;  // consecutive survivors per 32-lane group: neighbours on a ray sit 0.5 voxel apart, so the
                               // footprints of four of them (+/-2 voxel taps, +1 corner) fit the brick of the first

// trilinear scatter of `go` into the LDS brick (origin bx0,by0,bz0); a corner outside the brick (a survivor of the group
// that belongs to the next ray) goes straight to global memory
__device__ __forceinline__ void brick_scatter(float *brick, int bx0, int by0, int bz0, float *__restrict__ grid,
                                              const GridDesc &d, const TriCorners &t, float go) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int x = t.x0 + (k >> 2), y = t.y0 + ((k >> 1) & 1), z = t.z0 + (k & 1);
    if (!(fgs_in(x, (int)d.X) && fgs_in(y, (int)d.Y) && fgs_in(z, (int)d.Z))) continue;
    const int rx = x - bx0, ry = y - by0, rz = z - bz0;
    if (fgs_in(rx, BRICK) && fgs_in(ry, BRICK) && fgs_in(rz, BRICK))
      atomicAdd(brick + (rx * BRICK + ry) * BRICK + rz, t.w[k] * go);
    else
      atomicAdd(grid + x * d.sX + y * d.sY + z * d.sZ, t.w[k] * go);
  }
}

// Scatter of every sdf.grad contribution of the survivors: hierarchical taps (dX0 columns) plus, when tot_sdf /
// tot_grad are given, the centre lookup and +/-1 taps whose gradients k_march_fine_bwd accumulated per survivor.
// 32 lanes walk TAPS_GROUP consecutive survivors and sum everything they add to sdf.grad in one LDS brick, flushed
// row-wise.  Memory-side float atomics are priced per 64-byte line per wave-instruction (MI355X_MICROARCH.md, global
// float atomics): the ~30 trilinear footprints of one survivor (24 hierarchical taps, the centre lookup and its six
// +/-1 taps) overlap heavily, and so do those of its neighbours on the ray -- ~250 line requests per survivor become
// ~30 with a brick per survivor and ~12 with a brick per four.  (Timing, 50 K survivors: 103 us either way -- the kernel
// was bound by the LDS float atomics of the 24 tap footprints, 55 us of it, not by the memory-side ones; removing the
// flush entirely saves 3 us.  Summing the eight taps of an axis per slab before the scatter halves those: 77 us.)
__global__ __launch_bounds__(FGS_BLOCK) void k_feat_taps_bwd(SurvArgs S, const float *__restrict__ X0,
                                                             const float *__restrict__ dX0,
                                                             const float *__restrict__ tot_sdf,
                                                             const float *__restrict__ tot_grad,
                                                             float *__restrict__ sdf_grad_grid, int prio) {
  fgs_setprio(prio);      // (FGS_PRIO_TAPS_BWD: the kernel runs beside k_mlp_wgrad, see fused.py _wgrad)
  // (S is only read: a kernel-argument struct that is written to AND indexed with a run-time index -- disp[j % K] -- is copied
  // to scratch memory, 232 bytes per lane, and every later field access becomes a scratch load)
  const int64_t M_rows = fgs_rows(S.M, S.m_dev);
  // (launched for a CAPACITY of rows under a device-side count: workgroups wholly beyond the count leave before they clear,
  // walk and flush their bricks -- 14 us of a 110 us launch at capacity = 1.5 x count)
  if ((int64_t)blockIdx.x * (FGS_BLOCK / 32) * TAPS_GROUP >= M_rows) return;
  __shared__ float brick_all[FGS_BLOCK / 32][BRICK * BRICK * BRICK];
  const int64_t m_first = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5) * TAPS_GROUP;
  const int j = threadIdx.x & 31;
  const int K = S.L.K;
  const GridDesc gd = fgs_sdf_desc(S.geom);
  float *brick = brick_all[threadIdx.x >> 5];
  int bx0 = 0, by0 = 0, bz0 = 0;
  if (m_first < M_rows) {
    const PointIdx p0 = fgs_point_to_index(S.pts[3 * m_first], S.pts[3 * m_first + 1], S.pts[3 * m_first + 2], S.geom.lo,
                                           S.geom.hi, gd);
    bx0 = (int)fgs_safe_floor(p0.fx) - BRICK_LO;
    by0 = (int)fgs_safe_floor(p0.fy) - BRICK_LO;
    bz0 = (int)fgs_safe_floor(p0.fz) - BRICK_LO;
  }
  for (int e = j; e < BRICK * BRICK * BRICK; e += 32) brick[e] = 0.f;
  __syncthreads();
  for (int gi = 0; gi < TAPS_GROUP; ++gi) {
    const int64_t m = m_first + gi;
    const bool row_ok = m < M_rows;
    const bool tap_lane = row_ok && j < 6 * K;
    float f = 0.f, cl = 0.f, d_f = 0.f;
    TapPoint tp = {0.f, 0.f, 0.f, 0.f};
    PointIdx pc = {0.f, 0.f, 0.f};
    if (row_ok) pc = fgs_point_to_index(S.pts[3 * m], S.pts[3 * m + 1], S.pts[3 * m + 2], S.geom.lo, S.geom.hi, gd);
    if (tap_lane) {
      tp = fgs_tap_point(pc, gd, j / K, S.L.disp[j % K]);
      f = X0[m * S.L.ldx0 + S.L.off_feat + j];  // saved forward value
      cl = tp.clamped;
      d_f = dX0[m * S.L.dx_ld + S.L.off_feat - S.L.dx_gap + j];
    }
    // lanes j < 3K: gradient of the (optionally normalised) finite difference w.r.t. the raw difference
    const int a = (j < 3 * K) ? j / K : 0, k = (j < 3 * K) ? j % K : 0;
    const float fp = group_shfl(f, (2 * a + 1) * K + k), fm = group_shfl(f, (2 * a) * K + k);
    const float cp = group_shfl(cl, (2 * a + 1) * K + k), cm = group_shfl(cl, (2 * a) * K + k);
    const float diff = cp - cm;
    const float g_raw = ((fp - fm) / diff) / S.geom.voxel_size;
    float dy = (row_ok && j < 3 * K) ? dX0[m * S.L.dx_ld + S.L.off_hgrad - S.L.dx_gap + j] : 0.f;
    float dg = dy;
    if (S.L.use_grad_norm) {
      const float g0 = group_shfl(g_raw, k), g1 = group_shfl(g_raw, K + k), g2 = group_shfl(g_raw, 2 * K + k);
      const float y0 = group_shfl(dy, k), y1 = group_shfl(dy, K + k), y2 = group_shfl(dy, 2 * K + k);
      const float r = sqrtf((g0 * g0 + g1 * g1) + g2 * g2);
      const float dot = (y0 * g0 + y1 * g1) + y2 * g2;
      dg = dy / (r + 1e-5f);
      if (r > 0.f) dg -= dot / (r * (r + 1e-5f) * (r + 1e-5f)) * g_raw;
    }
    const float coef = (j < 3 * K && K > 0) ? (dg / S.geom.voxel_size) / diff : 0.f;
    // back to the tap lanes: tap (pair, k) takes +/- coef of axis pair>>1
    const int pair = (j < 6 * K) ? j / K : 0, kk = (j < 6 * K) ? j % K : 0;
    const float c_axis = group_shfl(coef, (pair >> 1) * K + kk);
    // The eight taps of an axis differ only in their coordinate ALONG that axis: their footprints share the 2x2 weights
    // of the other two axes.  Each group of 8 lanes first sums its 16 along-axis contributions into the (at most 8)
    // slabs they touch (shuffles), then lane q adds slab q to the brick through the four shared cross weights: 4 LDS
    // atomics per tap lane instead of 8, and no two lanes of a group on the same cell.
    {
      const float total = tap_lane ? d_f + ((pair & 1) ? c_axis : -c_axis) : 0.f;
      const int axis = pair >> 1;                                        // 0 -> z, 1 -> y, 2 -> x (tap lanes only)
      const float flx = fgs_safe_floor(tp.fx), fly = fgs_safe_floor(tp.fy), flz = fgs_safe_floor(tp.fz);
      const int x0 = (int)flx, y0 = (int)fly, z0 = (int)flz;
      const float wx[2] = {(flx + 1.f) - tp.fx, tp.fx - flx}, wy[2] = {(fly + 1.f) - tp.fy, tp.fy - fly};
      const float wz[2] = {(flz + 1.f) - tp.fz, tp.fz - flz};
      const int s_t = (axis == 0) ? z0 : (axis == 1 ? y0 : x0);
      const float a0 = (axis == 0) ? wz[0] : (axis == 1 ? wy[0] : wx[0]), a1 = (axis == 0) ? wz[1] : (axis == 1 ? wy[1] : wx[1]);
      const float c0 = total * a0, c1 = total * a1;
      int base = s_t;
#pragma unroll
      for (int off = 4; off > 0; off >>= 1) base = min(base, __shfl_xor(base, off, 8));
      const int q = j & 7;
      float A = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int su = __shfl(s_t, u, 8);
        const float c0u = __shfl(c0, u, 8), c1u = __shfl(c1, u, 8);
        A += (su == base + q) ? c0u : 0.f;
        A += (su + 1 == base + q) ? c1u : 0.f;
      }
      bool spill = false;                                                // a tap more than 7 slabs from the lowest: cannot
#pragma unroll                                                           // happen for displacements <= 2 voxels
      for (int u = 0; u < 8; ++u) spill |= __shfl(s_t, u, 8) + 1 > base + 7;
      if (tap_lane && spill) {
        if (total != 0.f) brick_scatter(brick, bx0, by0, bz0, sdf_grad_grid, gd, fgs_tri_setup(tp.fx, tp.fy, tp.fz), total);
      } else if (tap_lane && A != 0.f) {
        const int sl = base + q;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i1 = k >> 1, i2 = k & 1;                             // offsets on the two other axes
          int x, y, z;
          float cw;
          if (axis == 0) { z = sl; y = y0 + i1; x = x0 + i2; cw = wy[i1] * wx[i2]; }
          else if (axis == 1) { y = sl; z = z0 + i1; x = x0 + i2; cw = wz[i1] * wx[i2]; }
          else { x = sl; z = z0 + i1; y = y0 + i2; cw = wz[i1] * wy[i2]; }
          if (!(fgs_in(x, (int)gd.X) && fgs_in(y, (int)gd.Y) && fgs_in(z, (int)gd.Z))) continue;
          const int rx = x - bx0, ry = y - by0, rz = z - bz0;
          const float v = A * cw;
          if (fgs_in(rx, BRICK) && fgs_in(ry, BRICK) && fgs_in(rz, BRICK)) atomicAdd(brick + (rx * BRICK + ry) * BRICK + rz, v);
          else atomicAdd(sdf_grad_grid + x * gd.sX + y * gd.sY + z * gd.sZ, v);
        }
      }
    }
    if (row_ok && tot_sdf && j >= 24 && j < 31) {
      // lane 24: centre lookup (d sdf); lanes 25..30: the six +/-1 voxel taps of the finite-difference gradient
      if (j == 24) {
        const float g = tot_sdf[m];
        if (g != 0.f) brick_scatter(brick, bx0, by0, bz0, sdf_grad_grid, gd, fgs_tri_setup(pc.fx, pc.fy, pc.fz), g);
      } else {
        const int pr = j - 25, ax = pr >> 1;                       // ax: 0 -> z, 1 -> y, 2 -> x
        const float dgax = tot_grad[3 * m + (2 - ax)];
        if (dgax != 0.f) {
          const TapPoint tm = fgs_tap_point(pc, gd, 2 * ax, 1.0f), tq = fgs_tap_point(pc, gd, 2 * ax + 1, 1.0f);
          const float cf = (dgax / S.geom.voxel_size) / (tq.clamped - tm.clamped);
          const bool hi = pr & 1;
          brick_scatter(brick, bx0, by0, bz0, sdf_grad_grid, gd,
                        fgs_tri_setup(hi ? tq.fx : tm.fx, hi ? tq.fy : tm.fy, hi ? tq.fz : tm.fz), hi ? cf : -cf);
        }
      }
    }
  }
  __syncthreads();
  if (m_first < M_rows) {
    for (int e = j; e < BRICK * BRICK * BRICK; e += 32) {  // consecutive lanes = consecutive z of one (x,y) row
      const float v = brick[e];
      if (v == 0.f) continue;
      const int x = bx0 + e / (BRICK * BRICK), y = by0 + (e / BRICK) % BRICK, z = bz0 + e % BRICK;
      atomicAdd(sdf_grad_grid + (int64_t)x * gd.sX + (int64_t)y * gd.sY + z, v);  // nonzero entries are in-volume
    }
  }
}

// --------------------------------------------------------- encodings, normal, reflection (16 lanes per survivor)
struct Normal3 {
  float n[3];
};

// normal = l2_normalize(g / (|g| + 1e-7)), model/nerf.py:835,480-483
__device__ __forceinline__ Normal3 normal_of(float gx, float gy, float gz) {
  const float r = sqrtf((gx * gx + gy * gy) + gz * gz);
  const float x0 = gx / (r + 1e-7f), x1 = gy / (r + 1e-7f), x2 = gz / (r + 1e-7f);
  const float q = (x0 * x0 + x1 * x1) + x2 * x2;
  const float nr = sqrtf(fmaxf(q, 1.1920928955078125e-07f));
  Normal3 o;
  o.n[0] = x0 / nr; o.n[1] = x1 / nr; o.n[2] = x2 / nr;
  return o;
}

// 64 survivors per workgroup, two phases.  Phase 1, one lane per survivor (the first wavefront): the normal, the scaled
// position and the reflection direction -- two square roots and nine correctly rounded divisions -- once per survivor, into LDS;
// normal_out is written from here.  Phase 2, all four wavefronts over the flat list of (survivor, item) pairs, consecutive
// lanes = consecutive items of one survivor = consecutive columns: an item is one (sin, cos) pair of the 3 (F_pos + F_view +
// F_ref) the row holds, or one of the raw / scalar / padding columns.  (Until round 4: 32 lanes per survivor, every lane
// repeating phase 1 -- ~200 of the ~400 vector instructions a wavefront issued for TWO survivors; a wave64 instruction
// occupies its SIMD for four cycles, and the kernel was bound by exactly that: 32 us per 57 K survivors, 154 us per 354 K.)
constexpr int ENC_SV = 64;        // survivors per workgroup
constexpr int ENC_LD = 17;        // floats per survivor in LDS: u[3] v[3] refl[3] col_grad[3] sdf, pad (odd pitch)
__global__ __launch_bounds__(FGS_BLOCK) void k_feat_enc_fwd(SurvArgs S, float *X0, float *Z /* == X0 in coarse mode */,
                                                            float *__restrict__ normal_out) {
  const int64_t M = fgs_rows(S.M, S.m_dev);
  const int64_t m0 = (int64_t)blockIdx.x * ENC_SV;
  if (m0 >= M) return;
  const FeatLayout &L = S.L;
  __shared__ float sv[ENC_SV * ENC_LD];
  const int n_here = (int)((M - m0 < ENC_SV) ? (M - m0) : ENC_SV);
  if (threadIdx.x < n_here) {
    const int64_t m = m0 + threadIdx.x;
    const int64_t r = S.ray_id[m];
    const float v[3] = {S.viewdirs[3 * r], S.viewdirs[3 * r + 1], S.viewdirs[3 * r + 2]};
    const float g[3] = {S.gradient[3 * m], S.gradient[3 * m + 1], S.gradient[3 * m + 2]};
    const Normal3 nn = normal_of(g[0], g[1], g[2]);
    const float dot = (v[0] * nn.n[0] + v[1] * nn.n[1]) + v[2] * nn.n[2];
    float *o = sv + threadIdx.x * ENC_LD;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      o[c] = (S.pts[3 * m + c] - S.geom.lo[c]) / (S.geom.hi[c] - S.geom.lo[c]);   // rays_xyz, model/nerf.py:837
      o[3 + c] = v[c];
      o[6 + c] = v[c] - (2.f * dot) * nn.n[c];                                    // model/nerf.py:879
      o[9 + c] = L.coarse ? nn.n[c] : g[c];
      normal_out[3 * m + c] = nn.n[c];
    }
    o[12] = L.center_sdf ? S.sdf[m] : 0.f;
  }
  __syncthreads();
  const int Fp = L.n_posfreq, Fv = L.use_viewdir ? L.n_viewfreq : 0, Fr = L.n_reffreq;
  const int n_pairs = 3 * (Fp + Fv + Fr);
  // raw items behind the pairs: u[3], v[3] (when used), refl[3], the gradient / normal columns [3], sdf (when used), the padding
  // columns of X0 and (fine layout) of Z
  const int pad_x = L.ldx0 - L.x0_cols, pad_z = L.coarse ? 0 : L.ldz - L.z_cols;
  const int n_items = n_pairs + 13 + pad_x + pad_z;
  const float inv_items = 1.f / (float)n_items;
  const int total = n_here * n_items;
  for (int i = threadIdx.x; i < total; i += FGS_BLOCK) {
    int ml = (int)(((float)i + 0.5f) * inv_items);           // i / n_items (i < 2^16: the estimate is off by at most one)
    int p = i - ml * n_items;
    if (p < 0) { --ml; p += n_items; }
    if (p >= n_items) { ++ml; p -= n_items; }
    const float *o = sv + ml * ENC_LD;
    float *x0 = X0 + (m0 + ml) * L.ldx0;
    float *zr = Z + (m0 + ml) * L.ldz;
    if (p < n_pairs) {
      int q = p, F = Fp, set = 0;
      float *row = x0 + L.off_xyz;
      if (q >= 3 * Fp) {
        q -= 3 * Fp; F = Fv; row = x0 + L.off_view; set = 3;
        if (q >= 3 * Fv) { q -= 3 * Fv; F = Fr; row = zr + L.off_ref; set = 6; }
      }
      const int c = q / F, f = q - c * F;
      float sn, cs;
      sincosf(o[set + c] * (float)(1 << f), &sn, &cs);     // freq = 2^f exactly, as the repeated doubling gives
      row[3 + c * F + f] = sn;
      row[3 + 3 * F + c * F + f] = cs;
      continue;
    }
    const int e = p - n_pairs;
    if (e < 3) x0[L.off_xyz + e] = o[e];
    else if (e < 6) { if (L.use_viewdir) x0[L.off_view + (e - 3)] = o[e]; }
    else if (e < 9) zr[L.off_ref + (e - 6)] = o[e];
    else if (e < 12) x0[L.off_grad + (e - 9)] = o[e];
    else if (e == 12) { if (L.center_sdf) x0[L.off_sdf] = o[12]; }
    else if (e < 13 + pad_x) x0[L.x0_cols + (e - 13)] = 0.f;
    else zr[L.z_cols + (e - 13 - pad_x)] = 0.f;
  }
}

// One thread per survivor: gradients reaching sdf and the raw gradient vector through the feature columns, the normal
// (orientation loss, reflection) and the reflection encoding.  (The first form dealt the 3 F_ref (sin, cos) column pairs to 32
// lanes per survivor, summed them with shuffles and left the divisions and square roots of the tail to lane 0: 2 of 64 lanes
// busy through ~400 instructions, VALU-bound at 27 us -- and the fp32 matrix instructions of k_mlp_wgrad, which this kernel
// runs beside, execute on the same vector pipe.)  A thread reads its row's 6 F_ref + 3 columns of Z and dZ itself: four cache
// lines per survivor.
// (FREF: the number of reflection frequencies as a compile-time constant, 0 = read it from the layout -- with the loops
// unrolled all 4 * 3 F loads of a thread are in flight together instead of one dependent round trip per frequency)
template <int FREF>
__global__ __launch_bounds__(FGS_BLOCK) void k_feat_enc_bwd(SurvArgs S, const float *__restrict__ Z,
                                                            const float *__restrict__ dX0, const float *__restrict__ dZ,
                                                            const float *__restrict__ g_normal, float *__restrict__ g_sdf,
                                                            float *__restrict__ g_gradient) {
  __builtin_amdgcn_s_setprio(2);             // beside the matrix kernel: first pick of the shared vector pipe (a few us)
  S.M = fgs_rows(S.M, S.m_dev);
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= S.M) return;
  const FeatLayout &L = S.L;
  const int F = FREF ? FREF : L.n_reffreq;
  const float *z = Z + m * L.ldz + L.off_ref;
  // (coarse stages: the reflection encoding is a block of X0 itself, so its gradient is a block of dX0 -- read in dX0's form)
  const float *dz = L.coarse ? dZ + m * L.dx_ld + L.off_ref - L.dx_gap : dZ + m * L.ldz + L.off_ref;
  // d reflect_c = dE[c] + sum_f 2^f (cos * dE_sin - sin * dE_cos)
  float part[3] = {0.f, 0.f, 0.f};
  if (FREF) {
    float zs[3 * (FREF ? FREF : 1)], zc[3 * (FREF ? FREF : 1)], ds[3 * (FREF ? FREF : 1)], dc[3 * (FREF ? FREF : 1)];
#pragma unroll
    for (int i = 0; i < 3 * FREF; ++i) { zs[i] = z[3 + i]; zc[i] = z[3 + 3 * FREF + i]; ds[i] = dz[3 + i]; dc[i] = dz[3 + 3 * FREF + i]; }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int f = 0; f < FREF; ++f)
        part[c] += (float)(1 << f) * (zc[c * FREF + f] * ds[c * FREF + f] - zs[c * FREF + f] * dc[c * FREF + f]);
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      for (int f = 0; f < F; ++f) {
        const float sn = z[3 + c * F + f], cs = z[3 + 3 * F + c * F + f];
        part[c] += (float)(1 << f) * (cs * dz[3 + c * F + f] - sn * dz[3 + 3 * F + c * F + f]);
      }
  }
  const float dr[3] = {dz[0] + part[0], dz[1] + part[1], dz[2] + part[2]};
  const int64_t r = S.ray_id[m];
  const float v[3] = {S.viewdirs[3 * r], S.viewdirs[3 * r + 1], S.viewdirs[3 * r + 2]};
  const float g[3] = {S.gradient[3 * m], S.gradient[3 * m + 1], S.gradient[3 * m + 2]};
  // recompute the two-stage normalisation
  const float rn = sqrtf((g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]);
  const float x[3] = {g[0] / (rn + 1e-7f), g[1] / (rn + 1e-7f), g[2] / (rn + 1e-7f)};
  const float q = (x[0] * x[0] + x[1] * x[1]) + x[2] * x[2];
  const float eps = 1.1920928955078125e-07f;
  const float nr = sqrtf(fmaxf(q, eps));
  const float n[3] = {x[0] / nr, x[1] / nr, x[2] / nr};
  // reflect = v - 2 (v.n) n
  const float s = (v[0] * n[0] + v[1] * n[1]) + v[2] * n[2];
  const float drn = (dr[0] * n[0] + dr[1] * n[1]) + dr[2] * n[2];
  float dn[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    dn[c] = -2.f * (v[c] * drn + s * dr[c]) + (g_normal ? g_normal[3 * m + c] : 0.f);
    if (L.coarse) dn[c] += dX0[m * L.dx_ld + L.off_grad - L.dx_gap + c];  // the normal itself is an MLP input column
  }
  // normal = x / sqrt(max(sum x^2, eps))
  const float dnx = (dn[0] * x[0] + dn[1] * x[1]) + dn[2] * x[2];
  float dx[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) dx[c] = dn[c] / nr - ((q > eps) ? dnx / (nr * nr * nr) * x[c] : 0.f);
  // x = g / (|g| + 1e-7)
  const float dxg = (dx[0] * g[0] + dx[1] * g[1]) + dx[2] * g[2];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float dg = dx[c] / (rn + 1e-7f);
    if (rn > 0.f) dg -= dxg / ((rn + 1e-7f) * (rn + 1e-7f)) * (g[c] / rn);
    g_gradient[3 * m + c] = L.coarse ? dg : dg + dX0[m * L.dx_ld + L.off_grad - L.dx_gap + c];
  }
  if (g_sdf) g_sdf[m] = L.center_sdf ? dX0[m * L.dx_ld + L.off_sdf - L.dx_gap] : 0.f;
}

// ------------------------------------------------------------------------- 3-wide output head (refnet last Linear)
// out = R3 . V4^T + c4 ; rgb = sigmoid(out).  16 lanes per row, four rows per wavefront and pass, two passes in flight: lane l
// of a row's group owns columns 4l + 64j (j < 4, W <= 256), the three dot products are summed over the 16 lanes of a DPP row
// with quad_perm / row_half_mirror / row_mirror adds -- no LDS traffic.  (Until round 4 one wavefront per row: one 16-byte load
// per lane and 18 ds_bpermute per row, a chain of dependent latencies: 91 us for 354 K rows of 192, 3 TB/s.)
__device__ __forceinline__ float head_row16_sum(float v) {
  // lanes of a DPP row (16): quad sums, then the other quad of the half (i <-> 7 - i), then the other half (i <-> 15 - i)
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

__global__ __launch_bounds__(FGS_BLOCK) void k_head_fwd(const float *__restrict__ R, int64_t ldr, int W, int64_t M,
                                                        const float *__restrict__ V, const float *__restrict__ bias,
                                                        float *__restrict__ rgb, const int64_t *__restrict__ m_dev) {
  M = fgs_rows(M, m_dev);
  const int lane = threadIdx.x & 63, l = lane & 15, g = lane >> 4;
  const int64_t wave = (int64_t)blockIdx.x * (FGS_BLOCK / FGS_WAVE) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (FGS_BLOCK / FGS_WAVE);
  float4 w[3][4];
  bool ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ok[j] = 4 * l + 64 * j < W;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      w[c][j] = ok[j] ? *reinterpret_cast<const float4 *>(V + c * W + 4 * l + 64 * j) : make_float4(0, 0, 0, 0);
  }
  const float b0 = bias[0], b1 = bias[1], b2 = bias[2];
  constexpr int PASSES = 2;
  for (int64_t mb = wave * 4; mb < M; mb += n_waves * 4 * PASSES) {
    float4 x[PASSES][4];
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
      const int64_t m = mb + u * n_waves * 4 + g;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        x[u][j] = (m < M && ok[j]) ? *reinterpret_cast<const float4 *>(R + m * ldr + 4 * l + 64 * j) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
      const int64_t m = mb + u * n_waves * 4 + g;
      float a[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        // (the order of the old kernel inside a lane's four columns; across lanes the sum order differs: float rounding only)
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          t += ((x[u][j].x * w[c][j].x + x[u][j].y * w[c][j].y) + x[u][j].z * w[c][j].z) + x[u][j].w * w[c][j].w;
        a[c] = head_row16_sum(t);
      }
      if (l == 0 && m < M) {
        rgb[3 * m + 0] = 1.f / (1.f + expf(-(a[0] + b0)));
        rgb[3 * m + 1] = 1.f / (1.f + expf(-(a[1] + b1)));
        rgb[3 * m + 2] = 1.f / (1.f + expf(-(a[2] + b2)));
      }
    }
  }
}

constexpr int HEAD_ILP = 4;
// d_out [M,3] -> dR = (d_out . V4) * (R > 0) ; dV4 += d_out^T R ; dc4 += colsum(d_out) ; dR_colsum += colsum(dR)
__global__ __launch_bounds__(FGS_BLOCK) void k_head_bwd(const float *__restrict__ R, int64_t ldr, int W, int64_t M,
                                                        const float *__restrict__ V, const float *__restrict__ d_out,
                                                        float *__restrict__ dR, float *__restrict__ dV,
                                                        float *__restrict__ dbias, float *__restrict__ dR_colsum,
                                                        float *__restrict__ scratch /* [gridDim.x][4W + 4] or null */,
                                                        const int64_t *__restrict__ m_dev) {
  M = fgs_rows(M, m_dev);
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (FGS_BLOCK / FGS_WAVE) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (FGS_BLOCK / FGS_WAVE);
  const bool col_ok = 4 * lane < W;
  float4 w[3], acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    w[c] = col_ok ? *reinterpret_cast<const float4 *>(V + c * W + 4 * lane) : make_float4(0, 0, 0, 0);
    acc[c] = make_float4(0, 0, 0, 0);
  }
  float bsum[3] = {0.f, 0.f, 0.f};
  float4 osum = make_float4(0, 0, 0, 0);
  // each wave owns a contiguous run of rows and keeps HEAD_ILP row loads in flight
  const int64_t per = (M + n_waves - 1) / n_waves;
  const int64_t m_lo = wave * per, m_hi = (m_lo + per < M) ? m_lo + per : M;
  for (int64_t mb = m_lo; mb < m_hi; mb += HEAD_ILP) {
    float4 x[HEAD_ILP];
    float dd[HEAD_ILP][3];
#pragma unroll
    for (int u = 0; u < HEAD_ILP; ++u) {
      const int64_t m = mb + u;
      const bool ok = m < m_hi;
      x[u] = (ok && col_ok) ? *reinterpret_cast<const float4 *>(R + m * ldr + 4 * lane) : make_float4(0, 0, 0, 0);
      dd[u][0] = ok ? d_out[3 * m] : 0.f;
      dd[u][1] = ok ? d_out[3 * m + 1] : 0.f;
      dd[u][2] = ok ? d_out[3 * m + 2] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < HEAD_ILP; ++u) {
      const int64_t m = mb + u;
      if (m >= m_hi) continue;
      const float d0 = dd[u][0], d1 = dd[u][1], d2 = dd[u][2];
      bsum[0] += d0; bsum[1] += d1; bsum[2] += d2;
      if (!col_ok) continue;
      float4 o;
      o.x = (x[u].x > 0.f) ? (d0 * w[0].x + d1 * w[1].x) + d2 * w[2].x : 0.f;
      o.y = (x[u].y > 0.f) ? (d0 * w[0].y + d1 * w[1].y) + d2 * w[2].y : 0.f;
      o.z = (x[u].z > 0.f) ? (d0 * w[0].z + d1 * w[1].z) + d2 * w[2].z : 0.f;
      o.w = (x[u].w > 0.f) ? (d0 * w[0].w + d1 * w[1].w) + d2 * w[2].w : 0.f;
      *reinterpret_cast<float4 *>(dR + m * ldr + 4 * lane) = o;
      osum.x += o.x; osum.y += o.y; osum.z += o.z; osum.w += o.w;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        acc[c].x = fmaf(dd[u][c], x[u].x, acc[c].x);
        acc[c].y = fmaf(dd[u][c], x[u].y, acc[c].y);
        acc[c].z = fmaf(dd[u][c], x[u].z, acc[c].z);
        acc[c].w = fmaf(dd[u][c], x[u].w, acc[c].w);
      }
    }
  }
  // block reduction of the 16 per-lane partial sums (3x4 dV + 4 colsum) over the 4 waves, then one atomic each
  __shared__ float red[FGS_BLOCK / FGS_WAVE][16][FGS_WAVE];
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    red[wv][4 * c + 0][lane] = acc[c].x; red[wv][4 * c + 1][lane] = acc[c].y;
    red[wv][4 * c + 2][lane] = acc[c].z; red[wv][4 * c + 3][lane] = acc[c].w;
  }
  red[wv][12][lane] = osum.x; red[wv][13][lane] = osum.y; red[wv][14][lane] = osum.z; red[wv][15][lane] = osum.w;
  __syncthreads();
  // consecutive threads finish consecutive output elements: one atomic wave-instruction covers 4 cache lines, not 16
  // (memory-side float atomics are priced per 64-byte line, and every block hits the same 4W floats)
  float *part = scratch ? scratch + (int64_t)blockIdx.x * (4 * W + 4) : nullptr;
  for (int idx = threadIdx.x; idx < 4 * W; idx += FGS_BLOCK) {
    const int c = idx / W, col = idx - c * W, v = 4 * c + (col & 3), l = col >> 2;
    const float s = (red[0][v][l] + red[1][v][l]) + (red[2][v][l] + red[3][v][l]);
    if (part) part[idx] = s;                    // two-stage form: k_head_bwd_reduce sums the per-block partials
    else if (c < 3) atomicAdd(dV + idx, s);
    else if (dR_colsum) atomicAdd(dR_colsum + col, s);
  }
  // every lane of a wave walked the same rows: lane l < 3 holds the wave's sum of d_out[:, l]
  if (part) {
    __shared__ float bred[FGS_BLOCK / FGS_WAVE][3];
    if (lane < 3) bred[wv][lane] = bsum[lane];
    __syncthreads();
    if (threadIdx.x < 3) part[4 * W + threadIdx.x] = (bred[0][threadIdx.x] + bred[1][threadIdx.x]) + (bred[2][threadIdx.x] + bred[3][threadIdx.x]);
  } else if (lane < 3) {
    atomicAdd(dbias + lane, bsum[lane]);
  }
}

// second stage: element e of (dV [3W] | dR_colsum [W] | dbias [3]) += sum over blocks of scratch[b][e].  Workgroup =
// 64 elements x 4 slices of the partial rows, grid.y = 8 more slices: a thread sums n_blocks / 32 partials (4 loads in
// flight), the 4 slices meet in LDS, and the 8 workgroups of an element finish with one atomic each.  With atomics in
// the first stage instead, 1024 blocks x 4W floats land on the same 64 cache lines and the memory-side atomic units
// serialise them: 62 us for a pass that moves 100 MB.
__global__ __launch_bounds__(FGS_BLOCK) void k_head_bwd_reduce(const float *__restrict__ scratch, int n_blocks, int W,
                                                               float *__restrict__ dV, float *__restrict__ dbias,
                                                               float *__restrict__ dR_colsum) {
  __shared__ float red[4][64];
  const int le = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + le;
  const int n_el = 4 * W + 3, ld = 4 * W + 4;
  const int slice = blockIdx.y * 4 + sub, n_slices = gridDim.y * 4;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < n_el) {
    int b = slice;
    for (; b + 3 * n_slices < n_blocks; b += 4 * n_slices) {
      s0 += scratch[(int64_t)b * ld + e];
      s1 += scratch[(int64_t)(b + n_slices) * ld + e];
      s2 += scratch[(int64_t)(b + 2 * n_slices) * ld + e];
      s3 += scratch[(int64_t)(b + 3 * n_slices) * ld + e];
    }
    for (; b < n_blocks; b += n_slices) s0 += scratch[(int64_t)b * ld + e];
  }
  red[sub][le] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sub != 0 || e >= n_el) return;
  const float s = (red[0][le] + red[1][le]) + (red[2][le] + red[3][le]);
  if (e < 3 * W) atomicAdd(dV + e, s);
  else if (e < 4 * W) { if (dR_colsum) atomicAdd(dR_colsum + (e - 3 * W), s); }
  else atomicAdd(dbias + (e - 4 * W), s);
}

// -------------------------------------------------------------------------------------------- per-ray compositing
struct CompositeArgs {
  int64_t n_rays;
  const int64_t *surv_off;  // [n_rays + 1]
  const float *weights, *rgb, *normal;  // per survivor
  const int64_t *step_id;
  float bg, dist;
  float *rgb_marched, *sigmoid_rgb, *pre_rgb, *pre_sig;  // [n_rays,3]
  float *normal_marched;  // [n_rays,3] or null (render_grad)
  float *depth;           // [n_rays] or null (render_depth)
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(FGS_BLOCK) void k_composite_fwd(CompositeArgs C) {
  const int64_t ray = (int64_t)blockIdx.x * (FGS_BLOCK / FGS_WAVE) + fgs_uniform((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  if (ray >= C.n_rays) return;
  const int64_t s0 = fgs_uniform(C.surv_off[ray]), s1 = fgs_uniform(C.surv_off[ray + 1]);
  float acc[3] = {0, 0, 0}, sig[3] = {0, 0, 0}, nrm[3] = {0, 0, 0}, wsum = 0.f, dep = 0.f;
  for (int64_t i = s0 + lane; i < s1; i += FGS_WAVE) {
    const float w = C.weights[i];
    wsum += w;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = C.rgb[3 * i + c];
      acc[c] += w * v;
      sig[c] += w * (1.f / (1.f + expf(-v)));  // sigmoid_rgb = sigmoid(rgb): the reference's second sigmoid (:886)
      if (C.normal_marched) nrm[c] += w * C.normal[3 * i + c];
    }
    if (C.depth) dep += (w * (float)C.step_id[i]) * C.dist;
  }
  wsum = wave_sum(wsum);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    acc[c] = wave_sum(acc[c]);
    sig[c] = wave_sum(sig[c]);
    if (C.normal_marched) nrm[c] = wave_sum(nrm[c]);
  }
  if (C.depth) dep = wave_sum(dep);
  if (lane == 0) {
    const float bgterm = (1.f - wsum) * C.bg;  // model/nerf.py:899,902
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float a = acc[c] + bgterm, b = sig[c] + bgterm;
      C.pre_rgb[3 * ray + c] = a;
      C.pre_sig[3 * ray + c] = b;
      C.rgb_marched[3 * ray + c] = fminf(fmaxf(a, 0.f), 1.f);
      C.sigmoid_rgb[3 * ray + c] = fminf(fmaxf(b, 0.f), 1.f);
      if (C.normal_marched) C.normal_marched[3 * ray + c] = nrm[c];
    }
    if (C.depth) C.depth[ray] = dep;
  }
}

struct CompositeBwdArgs {
  int64_t M;
  const int64_t *m_dev;
  const int64_t *ray_id;
  const float *weights, *rgb;
  const float *pre_rgb, *pre_sig;                   // [n_rays,3]
  const float *g_rgb_marched, *g_sigmoid_rgb;       // [n_rays,3] or null
  const float *g_raw_rgb;                           // [M,3] or null
  const float *g_weights_direct;                    // [M] or null
  float bg;
  float *d_out;  // [M,3] gradient w.r.t. the pre-sigmoid head output
  float *d_w;    // [M]
};

__global__ __launch_bounds__(FGS_BLOCK) void k_composite_bwd(CompositeBwdArgs C) {
  C.M = fgs_rows(C.M, C.m_dev);
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= C.M) return;
  const int64_t r = C.ray_id[m];
  const float w = C.weights[m];
  float dw = C.g_weights_direct ? C.g_weights_direct[m] : 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = C.rgb[3 * m + c];
    const float sg = 1.f / (1.f + expf(-v));
    float g1 = 0.f, g2 = 0.f;  // clamp(0,1) passes the gradient on the closed interval
    if (C.g_rgb_marched) {
      const float pre = C.pre_rgb[3 * r + c];
      g1 = (pre >= 0.f && pre <= 1.f) ? C.g_rgb_marched[3 * r + c] : 0.f;
    }
    if (C.g_sigmoid_rgb) {
      const float pre = C.pre_sig[3 * r + c];
      g2 = (pre >= 0.f && pre <= 1.f) ? C.g_sigmoid_rgb[3 * r + c] : 0.f;
    }
    dw += v * g1 + sg * g2 - C.bg * (g1 + g2);
    float d_rgb = w * g1 + (w * g2) * (sg * (1.f - sg));
    if (C.g_raw_rgb) d_rgb += C.g_raw_rgb[3 * m + c];
    C.d_out[3 * m + c] = d_rgb * (v * (1.f - v));  // rgb = sigmoid(out)
  }
  C.d_w[m] = dw;
}

SceneGeom geom_of(const float *lo, const float *hi, int X, int Y, int Z, float voxel_size) {
  SceneGeom g;
  for (int c = 0; c < 3; ++c) { g.lo[c] = lo[c]; g.hi[c] = hi[c]; }
  g.X = X; g.Y = Y; g.Z = Z; g.voxel_size = voxel_size;
  return g;
}

int fill_layout(const int *li, const float *disp, FeatLayout *L, int compact = 0) {
  // li: k0_dim, n_posfreq, n_viewfreq, n_reffreq, use_viewdir, center_sdf, use_grad_norm, K, ldx0, off_ref, ldz
  L->k0_dim = li[0]; L->n_posfreq = li[1]; L->n_viewfreq = li[2]; L->n_reffreq = li[3];
  L->use_viewdir = li[4]; L->center_sdf = li[5]; L->use_grad_norm = li[6]; L->K = li[7];
  L->ldx0 = li[8]; L->off_ref = li[9]; L->ldz = li[10];
  L->coarse = 0;
  if (L->K < 0 || L->K > 5) return fgs_set_error(FGS_E_RANGE, "feature layout: K=%d displacements (0..5 supported)", L->K);
  for (int i = 0; i < MAXK; ++i) L->disp[i] = (i < L->K && disp) ? disp[i] : 0.f;
  // column order of torch.cat([k0, xyz_emb, viewdirs_emb, sdf, all_feat, all_grad, gradient]) (model/nerf.py:874)
  int c = 0;
  L->off_k0 = c; c += L->k0_dim;
  L->off_xyz = c; c += 3 + 6 * L->n_posfreq;
  L->off_view = c; c += L->use_viewdir ? 3 + 6 * L->n_viewfreq : 0;
  L->off_sdf = c; c += L->center_sdf ? 1 : 0;
  L->off_feat = c; c += 6 * L->K;
  L->off_hgrad = c; c += 3 * L->K;
  L->off_grad = c; c += 3;
  L->x0_cols = c;
  L->dx_gap = compact ? L->off_sdf - L->off_xyz : 0;
  L->dx_ld = compact ? (L->x0_cols - L->dx_gap + 3) / 4 * 4 : L->ldx0;
  L->z_cols = L->off_ref + 3 + 6 * L->n_reffreq;
  if (L->ldx0 < L->x0_cols || L->ldz < L->z_cols || (L->ldx0 & 3) || (L->ldz & 3))
    return fgs_set_error(FGS_E_INVALID, "feature layout: ldx0=%d (need >= %d), ldz=%d (need >= %d), multiples of 4",
                         L->ldx0, L->x0_cols, L->ldz, L->z_cols);
  return 0;
}

int fill_layout_coarse(const int *li, FeatLayout *L, int compact = 0) {
  // li: k0_dim, n_posfreq, n_viewfreq, n_reffreq, use_viewdir, ldx0
  L->k0_dim = li[0]; L->n_posfreq = li[1]; L->n_viewfreq = li[2]; L->n_reffreq = li[3]; L->use_viewdir = li[4];
  L->center_sdf = 0; L->use_grad_norm = 0; L->K = 0; L->ldx0 = li[5]; L->coarse = 1;
  for (int i = 0; i < MAXK; ++i) L->disp[i] = 0.f;
  // column order of torch.cat([k0, xyz_emb, reflect_emb, normal, viewdirs_emb]) (model/nerf.py:1007)
  int c = 0;
  L->off_k0 = c; c += L->k0_dim;
  L->off_xyz = c; c += 3 + 6 * L->n_posfreq;
  L->off_ref = c; c += 3 + 6 * L->n_reffreq;
  L->off_grad = c; c += 3;
  L->off_view = c; c += L->use_viewdir ? 3 + 6 * L->n_viewfreq : 0;
  L->off_sdf = L->off_feat = L->off_hgrad = c;
  L->x0_cols = c;
  // compact dX0 (fgs_dyn_t.dx0_compact): [k0 | reflect_emb | normal] -- the xyz encoding between k0 and the reflection block and
  // the view-direction encoding at the end are functions of the fixed ray inputs
  L->dx_gap = compact ? L->off_ref - L->off_xyz : 0;
  L->dx_ld = compact ? (L->off_view - L->dx_gap + 3) / 4 * 4 : L->ldx0;
  L->ldz = L->ldx0; L->z_cols = L->ldx0;
  if (L->ldx0 < L->x0_cols || (L->ldx0 & 3))
    return fgs_set_error(FGS_E_INVALID, "coarse feature layout: ldx0=%d (need >= %d, multiple of 4)", L->ldx0, L->x0_cols);
  return 0;
}

}  // namespace

// Coarse-stage operand rows [k0, xyz_emb, reflect_emb, normal, viewdirs_emb] (model/nerf.py:992-1009).
// layout_i: 6 ints (see fill_layout_coarse).
FGS_API int fgs_feat_coarse_fwd(int64_t M, const int64_t *ray_id, const float *pts, const float *gradient,
                                const float *viewdirs, const float *xyz_min_host, const float *xyz_max_host, int X, int Y,
                                int Z, const int *layout_i, const float *k0_grid, int64_t ksC, int64_t ksX, int64_t ksY,
                                int64_t ksZ, float *X0, float *normal_out, const fgs_dyn_t *dyn, fgs_stream_t stream) {
  FGS_REQUIRE(M >= 0 && M < ((int64_t)1 << 31), FGS_E_RANGE, "fgs_feat_coarse_fwd: M=%lld", (long long)M);
  if (M == 0) return 0;
  FGS_REQUIRE(ray_id && pts && gradient && viewdirs && xyz_min_host && xyz_max_host && layout_i && k0_grid && X0 && normal_out,
              FGS_E_INVALID, "fgs_feat_coarse_fwd: null pointer");
  SurvArgs S;
  S.M = M; S.m_dev = fgs_dyn_rows(dyn); S.ray_id = ray_id; S.pts = pts; S.sdf = nullptr; S.gradient = gradient; S.viewdirs = viewdirs;
  S.geom = geom_of(xyz_min_host, xyz_max_host, X, Y, Z, 0.f);
  if (int e = fill_layout_coarse(layout_i, &S.L)) return e;
  hipStream_t st = fgs_s(stream);
  const GridDesc kd{S.L.k0_dim, X, Y, Z, ksC, ksX, ksY, ksZ};
  hipLaunchKernelGGL(k_feat_k0_fwd, dim3(fgs_blocks(M * S.L.k0_dim)), dim3(FGS_BLOCK), 0, st, S, k0_grid, kd, X0);
  FGS_LAUNCH_OK("fgs_feat_coarse_fwd/k0");
  hipLaunchKernelGGL(k_feat_enc_fwd, dim3(fgs_blocks(M, ENC_SV)), dim3(FGS_BLOCK), 0, st, S, X0, X0, normal_out);
  FGS_LAUNCH_OK("fgs_feat_coarse_fwd/enc");
  return 0;
}

FGS_API int fgs_feat_coarse_bwd(int64_t M, const int64_t *ray_id, const float *pts, const float *gradient,
                                const float *viewdirs, const float *xyz_min_host, const float *xyz_max_host, int X, int Y,
                                int Z, const int *layout_i, const float *X0, const float *dX0, const float *g_normal,
                                float *k0_grad_grid, int64_t ksC, int64_t ksX, int64_t ksY, int64_t ksZ, float *g_gradient,
                                const fgs_dyn_t *dyn, fgs_stream_t stream) {
  FGS_REQUIRE(M >= 0 && M < ((int64_t)1 << 31), FGS_E_RANGE, "fgs_feat_coarse_bwd: M=%lld", (long long)M);
  if (M == 0) return 0;
  FGS_REQUIRE(ray_id && pts && gradient && viewdirs && xyz_min_host && xyz_max_host && layout_i && X0 && dX0 && k0_grad_grid &&
                  g_gradient, FGS_E_INVALID, "fgs_feat_coarse_bwd: null pointer");
  SurvArgs S;
  S.M = M; S.m_dev = fgs_dyn_rows(dyn); S.ray_id = ray_id; S.pts = pts; S.sdf = nullptr; S.gradient = gradient; S.viewdirs = viewdirs;
  S.geom = geom_of(xyz_min_host, xyz_max_host, X, Y, Z, 0.f);
  if (int e = fill_layout_coarse(layout_i, &S.L, fgs_dyn_compact(dyn))) return e;
  hipStream_t st = fgs_s(stream);
  const GridDesc kd{S.L.k0_dim, X, Y, Z, ksC, ksX, ksY, ksZ};
  hipLaunchKernelGGL(k_feat_k0_bwd, dim3(fgs_blocks(2 * ((M + k0_run() - 1) / k0_run()) * S.L.k0_dim)), dim3(FGS_BLOCK), 0, st, S, k0_grad_grid, kd, dX0, k0_run());
  FGS_LAUNCH_OK("fgs_feat_coarse_bwd/k0");
#define FGS_ENC_BWD(F) hipLaunchKernelGGL(k_feat_enc_bwd<F>, dim3(fgs_blocks(M)), dim3(FGS_BLOCK), 0, st, S, X0, dX0, dX0, g_normal, (float *)nullptr, g_gradient)
  switch (S.L.n_reffreq) {      // the shipped configs' frequency counts (config/shiny_blender.py: 3 / 5 / 8) unrolled
    case 3: FGS_ENC_BWD(3); break;
    case 5: FGS_ENC_BWD(5); break;
    case 8: FGS_ENC_BWD(8); break;
    default: FGS_ENC_BWD(0); break;
  }
#undef FGS_ENC_BWD
  FGS_LAUNCH_OK("fgs_feat_coarse_bwd/enc");
  return 0;
}

// layout_i: 11 ints (see fill_layout); displace_host: K floats.  Returns x0_cols through *x0_cols_out when non-null.
FGS_API int fgs_feat_fine_fwd(int64_t M, const int64_t *ray_id, const float *pts, const float *sdf, const float *gradient,
                              const float *viewdirs, const float *xyz_min_host, const float *xyz_max_host, int X, int Y,
                              int Z, float voxel_size, const int *layout_i, const float *displace_host,
                              const float *sdf_grid, const float *k0_grid, int64_t ksC, int64_t ksX, int64_t ksY,
                              int64_t ksZ, float *X0, float *Zbuf, float *normal_out, const fgs_dyn_t *dyn, fgs_stream_t stream) {
  FGS_REQUIRE(M >= 0 && M < ((int64_t)1 << 31), FGS_E_RANGE, "fgs_feat_fine_fwd: M=%lld", (long long)M);
  if (M == 0) return 0;
  FGS_REQUIRE(ray_id && pts && sdf && gradient && viewdirs && xyz_min_host && xyz_max_host && layout_i && sdf_grid &&
                  k0_grid && X0 && Zbuf && normal_out, FGS_E_INVALID, "fgs_feat_fine_fwd: null pointer");
  SurvArgs S;
  S.M = M; S.m_dev = fgs_dyn_rows(dyn); S.ray_id = ray_id; S.pts = pts; S.sdf = sdf; S.gradient = gradient; S.viewdirs = viewdirs;
  S.geom = geom_of(xyz_min_host, xyz_max_host, X, Y, Z, voxel_size);
  if (int e = fill_layout(layout_i, displace_host, &S.L)) return e;
  hipStream_t st = fgs_s(stream);
  const GridDesc kd{S.L.k0_dim, X, Y, Z, ksC, ksX, ksY, ksZ};
  hipLaunchKernelGGL(k_feat_k0_fwd, dim3(fgs_blocks(M * S.L.k0_dim)), dim3(FGS_BLOCK), 0, st, S, k0_grid, kd, X0);
  FGS_LAUNCH_OK("fgs_feat_fine_fwd/k0");
  if (S.L.K > 0) {
    hipLaunchKernelGGL(k_feat_taps_fwd, dim3(fgs_blocks(M * 32)), dim3(FGS_BLOCK), 0, st, S, sdf_grid, X0);
    FGS_LAUNCH_OK("fgs_feat_fine_fwd/taps");
  }
  hipLaunchKernelGGL(k_feat_enc_fwd, dim3(fgs_blocks(M, ENC_SV)), dim3(FGS_BLOCK), 0, st, S, X0, Zbuf, normal_out);
  FGS_LAUNCH_OK("fgs_feat_fine_fwd/enc");
  return 0;
}

FGS_API int fgs_feat_fine_bwd(int64_t M, const int64_t *ray_id, const float *pts, const float *sdf, const float *gradient,
                              const float *viewdirs, const float *xyz_min_host, const float *xyz_max_host, int X, int Y,
                              int Z, float voxel_size, const int *layout_i, const float *displace_host, const float *X0,
                              const float *Zbuf, const float *dX0, const float *dZ, const float *g_normal,
                              float *sdf_grad_grid, float *k0_grad_grid, int64_t ksC, int64_t ksX, int64_t ksY, int64_t ksZ,
                              float *g_sdf, float *g_gradient, const fgs_dyn_t *dyn, fgs_stream_t stream) {
  FGS_REQUIRE(M >= 0 && M < ((int64_t)1 << 31), FGS_E_RANGE, "fgs_feat_fine_bwd: M=%lld", (long long)M);
  if (M == 0) return 0;
  // k0_grad_grid == NULL: only the encoding part (g_sdf / g_gradient);  g_sdf == g_gradient == NULL: only the k0 scatter -- the two
  // kernels are independent, and the training step issues them on either side of its weight-gradient fork (fused.py)
  FGS_REQUIRE(ray_id && pts && sdf && gradient && viewdirs && xyz_min_host && xyz_max_host && layout_i && X0 && Zbuf && dX0 &&
                  dZ && (k0_grad_grid || (g_sdf && g_gradient)) && (!g_sdf == !g_gradient), FGS_E_INVALID,
              "fgs_feat_fine_bwd: null pointer");
  SurvArgs S;
  S.M = M; S.m_dev = fgs_dyn_rows(dyn); S.ray_id = ray_id; S.pts = pts; S.sdf = sdf; S.gradient = gradient; S.viewdirs = viewdirs;
  S.geom = geom_of(xyz_min_host, xyz_max_host, X, Y, Z, voxel_size);
  if (int e = fill_layout(layout_i, displace_host, &S.L, fgs_dyn_compact(dyn))) return e;
  hipStream_t st = fgs_s(stream);
  const GridDesc kd{S.L.k0_dim, X, Y, Z, ksC, ksX, ksY, ksZ};
  if (k0_grad_grid) {
    hipLaunchKernelGGL(k_feat_k0_bwd, dim3(fgs_blocks(2 * ((M + k0_run() - 1) / k0_run()) * S.L.k0_dim)), dim3(FGS_BLOCK), 0, st, S, k0_grad_grid, kd, dX0, k0_run());
    FGS_LAUNCH_OK("fgs_feat_fine_bwd/k0");
  }
  if (!g_gradient) return 0;
#define FGS_ENC_BWD(F) hipLaunchKernelGGL(k_feat_enc_bwd<F>, dim3(fgs_blocks(M)), dim3(FGS_BLOCK), 0, st, S, Zbuf, dX0, dZ, g_normal, g_sdf, g_gradient)
  switch (S.L.n_reffreq) {      // the shipped configs' frequency counts (config/shiny_blender.py: 3 / 5 / 8) unrolled
    case 3: FGS_ENC_BWD(3); break;
    case 5: FGS_ENC_BWD(5); break;
    case 8: FGS_ENC_BWD(8); break;
    default: FGS_ENC_BWD(0); break;
  }
#undef FGS_ENC_BWD
  FGS_LAUNCH_OK("fgs_feat_fine_bwd/enc");
  (void)sdf_grad_grid;  // the sdf.grad scatter of the survivors is fgs_sdf_scatter_surv (after fgs_march_fine_bwd)
  return 0;
}

FGS_API int fgs_sdf_scatter_surv(int64_t M, const float *pts, const float *xyz_min_host, const float *xyz_max_host, int X,
                                 int Y, int Z, float voxel_size, const int *layout_i, const float *displace_host,
                                 const float *X0, const float *dX0, const float *tot_sdf, const float *tot_grad,
                                 float *sdf_grad_grid, const fgs_dyn_t *dyn, fgs_stream_t stream) {
  FGS_REQUIRE(M >= 0 && M < ((int64_t)1 << 31), FGS_E_RANGE, "fgs_sdf_scatter_surv: M=%lld", (long long)M);
  if (M == 0) return 0;
  FGS_REQUIRE(pts && xyz_min_host && xyz_max_host && layout_i && X0 && dX0 && sdf_grad_grid && (!tot_sdf == !tot_grad),
              FGS_E_INVALID, "fgs_sdf_scatter_surv: null pointer");
  SurvArgs S;
  S.M = M; S.m_dev = fgs_dyn_rows(dyn); S.ray_id = nullptr; S.pts = pts; S.sdf = nullptr; S.gradient = nullptr; S.viewdirs = nullptr;
  S.geom = geom_of(xyz_min_host, xyz_max_host, X, Y, Z, voxel_size);
  if (int e = fill_layout(layout_i, displace_host, &S.L, fgs_dyn_compact(dyn))) return e;
  static const int prio = fgs_env_int("FGS_PRIO_TAPS_BWD", 0);
  hipLaunchKernelGGL(k_feat_taps_bwd, dim3(fgs_blocks((M + TAPS_GROUP - 1) / TAPS_GROUP * 32)), dim3(FGS_BLOCK), 0, fgs_s(stream), S, X0, dX0, tot_sdf,
                     tot_grad, sdf_grad_grid, prio);
  FGS_LAUNCH_OK("fgs_sdf_scatter_surv");
  return 0;
}

FGS_API int fgs_head_fwd(const float *R, int64_t ldr, int W, int64_t M, const float *V, const float *bias, float *rgb,
                         const fgs_dyn_t *dyn, fgs_stream_t stream) {
  FGS_REQUIRE(M >= 0 && M < ((int64_t)1 << 31) && W > 0 && W <= 256 && (W & 3) == 0 && (ldr & 3) == 0, FGS_E_RANGE,
              "fgs_head_fwd: M=%lld W=%d ldr=%lld (W <= 256, multiples of 4)", (long long)M, W, (long long)ldr);
  if (M == 0) return 0;
  FGS_REQUIRE(R && V && bias && rgb, FGS_E_INVALID, "fgs_head_fwd: null pointer");
  const int64_t want = (M + 31) / 32;            // a workgroup takes 4 waves x 4 rows x 2 passes per trip
  const unsigned blocks = (unsigned)(want < 2048 ? want : 2048);     // (512 .. 8192 measured alike: 5.5 .. 6.4 TB/s alone on the chip)
  hipLaunchKernelGGL(k_head_fwd, dim3(blocks), dim3(FGS_BLOCK), 0, fgs_s(stream), R, ldr, W, M, V, bias, rgb, fgs_dyn_rows(dyn));
  FGS_LAUNCH_OK("fgs_head_fwd");
  return 0;
}

FGS_API int64_t fgs_head_bwd_scratch_floats(int W) { return (int64_t)1024 * (4 * (int64_t)W + 4); }

FGS_API int fgs_head_bwd(const float *R, int64_t ldr, int W, int64_t M, const float *V, const float *d_out, float *dR,
                         float *dV, float *dbias, float *dR_colsum, float *scratch, const fgs_dyn_t *dyn, fgs_stream_t stream) {
  FGS_REQUIRE(M >= 0 && M < ((int64_t)1 << 31) && W > 0 && W <= 256 && (W & 3) == 0 && (ldr & 3) == 0, FGS_E_RANGE,
              "fgs_head_bwd: M=%lld W=%d ldr=%lld", (long long)M, W, (long long)ldr);
  if (M == 0) return 0;
  FGS_REQUIRE(R && V && d_out && dR && dV && dbias, FGS_E_INVALID, "fgs_head_bwd: null pointer");
  // >= 16 rows per wave, at most 4 blocks per CU: every block ends with ~4W atomics on the same few cache lines
  const int64_t want = (M + 63) / 64;
  static const int64_t cap = getenv("FGS_HEAD_BLOCKS") ? atoll(getenv("FGS_HEAD_BLOCKS")) : 1024;
  unsigned blocks = (unsigned)(want < cap ? want : cap);
  if (scratch && blocks > 1024) blocks = 1024;      // fgs_head_bwd_scratch_floats() sizes the scratch for 1024 blocks
  hipLaunchKernelGGL(k_head_bwd, dim3(blocks), dim3(FGS_BLOCK), 0, fgs_s(stream), R, ldr, W, M, V, d_out, dR, dV, dbias,
                     dR_colsum, scratch, fgs_dyn_rows(dyn));
  FGS_LAUNCH_OK("fgs_head_bwd");
  if (scratch) {
    hipLaunchKernelGGL(k_head_bwd_reduce, dim3((4 * W + 3 + 63) / 64, 8), dim3(FGS_BLOCK), 0, fgs_s(stream), scratch,
                       (int)blocks, W, dV, dbias, dR_colsum);
    FGS_LAUNCH_OK("fgs_head_bwd (reduce)");
  }
  return 0;
}

FGS_API int fgs_composite_fwd(int64_t n_rays, const int64_t *surv_off, const float *weights, const float *rgb,
                              const float *normal, const int64_t *step_id, float bg, float dist, float *rgb_marched,
                              float *sigmoid_rgb, float *pre_rgb, float *pre_sig, float *normal_marched, float *depth,
                              fgs_stream_t stream) {
  FGS_REQUIRE(n_rays >= 0 && n_rays < ((int64_t)1 << 31), FGS_E_RANGE, "fgs_composite_fwd: n_rays=%lld", (long long)n_rays);
  if (n_rays == 0) return 0;
  FGS_REQUIRE(surv_off && rgb_marched && sigmoid_rgb && pre_rgb && pre_sig, FGS_E_INVALID, "fgs_composite_fwd: null pointer");
  CompositeArgs C;
  C.n_rays = n_rays; C.surv_off = surv_off; C.weights = weights; C.rgb = rgb; C.normal = normal; C.step_id = step_id;
  C.bg = bg; C.dist = dist; C.rgb_marched = rgb_marched; C.sigmoid_rgb = sigmoid_rgb; C.pre_rgb = pre_rgb; C.pre_sig = pre_sig;
  C.normal_marched = normal_marched; C.depth = depth;
  hipLaunchKernelGGL(k_composite_fwd, dim3(fgs_blocks(n_rays * FGS_WAVE)), dim3(FGS_BLOCK), 0, fgs_s(stream), C);
  FGS_LAUNCH_OK("fgs_composite_fwd");
  return 0;
}

FGS_API int fgs_composite_bwd(int64_t M, const int64_t *ray_id, const float *weights, const float *rgb, const float *pre_rgb,
                              const float *pre_sig, const float *g_rgb_marched, const float *g_sigmoid_rgb,
                              const float *g_raw_rgb, const float *g_weights_direct, float bg, float *d_out, float *d_w,
                              const fgs_dyn_t *dyn, fgs_stream_t stream) {
  FGS_REQUIRE(M >= 0 && M < ((int64_t)1 << 31), FGS_E_RANGE, "fgs_composite_bwd: M=%lld", (long long)M);
  if (M == 0) return 0;
  FGS_REQUIRE(ray_id && weights && rgb && pre_rgb && pre_sig && d_out && d_w, FGS_E_INVALID, "fgs_composite_bwd: null pointer");
  CompositeBwdArgs C;
  C.M = M; C.m_dev = fgs_dyn_rows(dyn); C.ray_id = ray_id; C.weights = weights; C.rgb = rgb; C.pre_rgb = pre_rgb; C.pre_sig = pre_sig;
  C.g_rgb_marched = g_rgb_marched; C.g_sigmoid_rgb = g_sigmoid_rgb; C.g_raw_rgb = g_raw_rgb;
  C.g_weights_direct = g_weights_direct; C.bg = bg; C.d_out = d_out; C.d_w = d_w;
  hipLaunchKernelGGL(k_composite_bwd, dim3(fgs_blocks(M)), dim3(FGS_BLOCK), 0, fgs_s(stream), C);
  FGS_LAUNCH_OK("fgs_composite_bwd");
  return 0;
}
