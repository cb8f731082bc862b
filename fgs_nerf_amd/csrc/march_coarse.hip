// march_coarse.hip -- the fused front half of nerf.forward_coarse (model/nerf.py:943-990), one wavefront per ray.
//
// Differences from the fine stage (march.hip), all taken from the reference:
//   * the SDF value is the trilinear lookup of the SMOOTHED grid (smooth_conv(sdf.grid), :969-971), the gradient is the
//     trilinear lookup of the central-difference volume of the UN-smoothed grid (:972-973) -- both dense volumes are
//     produced by csrc/dense.hip once per forward;
//   * optional voxel-increment mask (MaskGrid nearest-voxel lookup, :962-967) besides the mask cache (:952-959);
//   * no `alpha > thres` compaction: Alphas2Weights runs on every sample (early termination at T < 1e-3), the
//     `weights > thres` mask selects the kept samples, and Alphas2Weights runs AGAIN on the kept list (:978-990).
//     Both exact sequential chains are replayed per 64-step chunk; kept samples behind the second chain's terminating
//     sample keep weight 0 (the reference's zero-initialised tail) but stay in the output lists.
// Records are written for the kept samples only; the backward walks the second chain's [i_start, i_end) prefix.
#include "fgs_march.h"

namespace {

struct IncMask {  // grid.MaskGrid buffers (model/grid.py:270-273), scale / shift by value
  const uint8_t *world;
  int sx, sy, sz;
  float scale[3], shift[3];
};

struct CoarseArgs {
  const float *rays_o, *rays_d, *viewdirs;
  int64_t n_rays;
  SceneGeom geom;
  float near, far, stepdist;
  const float *sdf_smooth;  // [X,Y,Z]
  const float *gradvol;     // [3,X,Y,Z]
  const float4 *vol4;       // optional [X,Y,Z] x {sdf_smooth, g_x, g_y, g_z}: one 16-byte load per trilinear corner
  float dist, inv_s, thres;
  const float *inv_s_dev;
  const float *mask_grid;
  SceneGeom mask_geom;
  float mask_thres;
  IncMask inc;
  int max_steps;
  int *a_step;
  float *a_alpha, *a_T, *a_weight, *a_sdf, *a_grad;
  int *a_surv, *surv_slot;
  int64_t *n_alive, *n_surv, *n_inbbox;
  float *alphainv_last;
};

__device__ __forceinline__ GridDesc grad_desc(const SceneGeom &s) {
  return GridDesc{3, s.X, s.Y, s.Z, (int64_t)s.X * s.Y * s.Z, (int64_t)s.Y * s.Z, (int64_t)s.Z, 1};
}

// one step of the reference's transmittance recurrence (render_utils_kernel.cu:596-597)
__device__ __forceinline__ bool chain_step(float &T_cum, float alpha_j) {
  T_cum = (float)((double)T_cum * (1. - (double)alpha_j));
  return fgs_uniform((double)T_cum < 1e-3 ? 1 : 0) != 0;
}

__global__ __launch_bounds__(FGS_BLOCK) void k_march_coarse_fwd(CoarseArgs A) {
  if (A.inv_s_dev) A.inv_s = *A.inv_s_dev;
  const int64_t ray = (int64_t)blockIdx.x * (FGS_BLOCK / FGS_WAVE) + fgs_uniform((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  if (ray >= A.n_rays) return;
  const float o[3] = {A.rays_o[3 * ray], A.rays_o[3 * ray + 1], A.rays_o[3 * ray + 2]};
  const float d[3] = {A.rays_d[3 * ray], A.rays_d[3 * ray + 1], A.rays_d[3 * ray + 2]};
  const float vx = A.viewdirs[3 * ray], vy = A.viewdirs[3 * ray + 1], vz = A.viewdirs[3 * ray + 2];
  const RaySetup rs = ray_setup(o, d, A.geom, A.near, A.far, A.stepdist);
  const int n_steps = (int)fgs_uniform((int64_t)(rs.n_steps < A.max_steps ? rs.n_steps : A.max_steps));
  const int64_t rec0 = ray * A.max_steps;
  const GridDesc sd = fgs_sdf_desc(A.geom), gd3 = grad_desc(A.geom);

  float T1 = 1.f, T2 = 1.f;
  bool term1 = false, term2 = false;
  int kept_base = 0, active_total = 0, inb_count = 0;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

  for (int base = 0; base < n_steps && !term1; base += FGS_WAVE) {
    const int s = base + lane;
    const float dist_s = A.stepdist * (float)s;
    const float px = fmaf(rs.dir[0], dist_s, rs.start[0]);
    const float py = fmaf(rs.dir[1], dist_s, rs.start[1]);
    const float pz = fmaf(rs.dir[2], dist_s, rs.start[2]);
    const bool outb = (A.geom.lo[0] > px) | (A.geom.lo[1] > py) | (A.geom.lo[2] > pz) | (A.geom.hi[0] < px) |
                      (A.geom.hi[1] < py) | (A.geom.hi[2] < pz);
    bool in = (s < n_steps) && !outb;
    inb_count += __popcll(__ballot(in));
    if (A.mask_grid && in) {
      const GridDesc md = fgs_sdf_desc(A.mask_geom);
      const PointIdx mp = fgs_point_to_index(px, py, pz, A.mask_geom.lo, A.mask_geom.hi, md);
      in = fgs_tri_sample(A.mask_grid, md, 0, fgs_tri_setup(mp.fx, mp.fy, mp.fz)) >= A.mask_thres;
    }
    if (A.inc.world && in) {  // render_utils_kernel.cu:385-390
      const int i = (int)roundf(fmaf(px, A.inc.scale[0], A.inc.shift[0]));
      const int j = (int)roundf(fmaf(py, A.inc.scale[1], A.inc.shift[1]));
      const int k = (int)roundf(fmaf(pz, A.inc.scale[2], A.inc.shift[2]));
      in = fgs_in(i, A.inc.sx) && fgs_in(j, A.inc.sy) && fgs_in(k, A.inc.sz) &&
           A.inc.world[((int64_t)i * A.inc.sy + j) * A.inc.sz + k] != 0;
    }
    float alpha = 0.f, sdf = 0.f, g[3] = {0.f, 0.f, 0.f};
    if (in) {
      const PointIdx p = fgs_point_to_index(px, py, pz, A.geom.lo, A.geom.hi, sd);
      const TriCorners t = fgs_tri_setup(p.fx, p.fy, p.fz);
      if (A.vol4) {
        // the four trilinear sums of fgs_tri_sample, corner by corner in the same order (bit-identical), from the
        // voxel-interleaved volume: 8 x 16 bytes instead of 32 x 4
        float4 v[8];
        float w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int x = t.x0 + (k >> 2), y = t.y0 + ((k >> 1) & 1), z = t.z0 + (k & 1);
          const bool ok = fgs_in(x, (int)sd.X) && fgs_in(y, (int)sd.Y) && fgs_in(z, (int)sd.Z);
          const int xc = min(max(x, 0), (int)sd.X - 1), yc = min(max(y, 0), (int)sd.Y - 1), zc = min(max(z, 0), (int)sd.Z - 1);
          v[k] = A.vol4[((int64_t)xc * sd.Y + yc) * sd.Z + zc];
          w[k] = ok ? t.w[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          sdf = fmaf(v[k].x, w[k], sdf);
          g[0] = fmaf(v[k].y, w[k], g[0]);
          g[1] = fmaf(v[k].z, w[k], g[1]);
          g[2] = fmaf(v[k].w, w[k], g[2]);
        }
      } else {
        sdf = fgs_tri_sample(A.sdf_smooth, sd, 0, t);
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c] = fgs_tri_sample(A.gradvol, gd3, c, t);
      }
      alpha = neus_alpha(sdf, g[0], g[1], g[2], vx, vy, vz, A.dist, A.inv_s);
    }
    // first Alphas2Weights (model/nerf.py:978): every sample, early termination
    unsigned long long bal = __ballot(in);
    float T_first = 1.f;
    int last1 = 64;
    while (bal) {
      const int j = __builtin_ctzll(bal);
      const float aj = fgs_bcast_lane(alpha, j);
      if (lane == j) T_first = T1;
      if (chain_step(T1, aj)) {
        term1 = true;
        last1 = j;
        break;
      }
      bal &= bal - 1;
    }
    const bool kept = in && lane <= last1 && (T_first * alpha > A.thres);  // mask = weights > fast_color_thres (:982)
    // second Alphas2Weights on the kept list (:990)
    const unsigned long long kept_bal = __ballot(kept);
    float T_second = 1.f;  // T is ones-initialised, weight zero-initialised (render_utils_kernel.cu:624-625)
    bool active = false;
    if (!term2) {
      unsigned long long b2 = kept_bal;
      while (b2) {
        const int j = __builtin_ctzll(b2);
        const float aj = fgs_bcast_lane(alpha, j);
        if (lane == j) {
          T_second = T2;
          active = true;
        }
        if (chain_step(T2, aj)) {
          term2 = true;
          break;
        }
        b2 &= b2 - 1;
      }
    }
    const float w2 = active ? T_second * alpha : 0.f;
    if (kept) {
      const int k_local = kept_base + __popcll(kept_bal & lt_mask);
      const int64_t rec = rec0 + k_local;
      A.a_step[rec] = s;
      A.a_alpha[rec] = alpha;
      A.a_T[rec] = T_second;
      A.a_weight[rec] = w2;
      A.a_sdf[rec] = sdf;
      A.a_grad[3 * rec + 0] = g[0];
      A.a_grad[3 * rec + 1] = g[1];
      A.a_grad[3 * rec + 2] = g[2];
      A.a_surv[rec] = k_local;          // every kept sample is a survivor
      A.surv_slot[rec0 + k_local] = k_local;
    }
    kept_base += __popcll(kept_bal);
    active_total += __popcll(__ballot(active));
  }
  if (lane == 0) {
    A.n_alive[ray] = active_total;   // the second chain's [i_start, i_end) prefix of the kept list
    A.n_surv[ray] = kept_base;
    A.n_inbbox[ray] = inb_count;
    A.alphainv_last[ray] = T2;
  }
}

struct CoarseBwdArgs {
  const float *rays_o, *rays_d, *viewdirs;
  int64_t n_rays;
  SceneGeom geom;
  float near, far, stepdist, dist, inv_s;
  const float *inv_s_dev;
  int max_steps;
  const int *a_step;
  const float *a_alpha, *a_T, *a_weight, *a_sdf, *a_grad;
  const int64_t *n_alive, *n_surv, *surv_off;
  const float *alphainv_last;
  const float *g_weights, *g_last, *g_gradient;  // [M_s], [n_rays] or null, [M_s,3] or null
  float *d_grid4;   // [X,Y,Z,4] voxel-interleaved (d smoothed-sdf, d gradient-volume xyz), accumulated with atomics
  float *g_inv_s;   // device float accumulating d loss / d inv_s (s_learn), or null
};

__global__ __launch_bounds__(FGS_BLOCK) void k_march_coarse_bwd(CoarseBwdArgs A) {
  if (A.inv_s_dev) A.inv_s = *A.inv_s_dev;
  const int64_t ray = (int64_t)blockIdx.x * (FGS_BLOCK / FGS_WAVE) + fgs_uniform((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  if (ray >= A.n_rays) return;
  const int n_active = (int)fgs_uniform(A.n_alive[ray]);
  int n_kept = (int)fgs_uniform(A.n_surv[ray]);
  const int64_t rec0 = ray * A.max_steps, s_off = fgs_uniform(A.surv_off[ray]);
  // (fgs_count_guard cuts the offsets at the buffers' capacity when a step's survivor list did not fit: never walk past
  // this ray's slots -- that step's update is skipped anyway)
  const int64_t n_slots = fgs_uniform(A.surv_off[ray + 1]) - s_off;
  if ((int64_t)n_kept > n_slots) n_kept = (int)n_slots;
  if (n_kept <= 0) return;
  const float o[3] = {A.rays_o[3 * ray], A.rays_o[3 * ray + 1], A.rays_o[3 * ray + 2]};
  const float d[3] = {A.rays_d[3 * ray], A.rays_d[3 * ray + 1], A.rays_d[3 * ray + 2]};
  const float vx = A.viewdirs[3 * ray], vy = A.viewdirs[3 * ray + 1], vz = A.viewdirs[3 * ray + 2];
  const RaySetup rs = ray_setup(o, d, A.geom, A.near, A.far, A.stepdist);
  const GridDesc sd = fgs_sdf_desc(A.geom);

  float back_cum = (A.g_last ? A.g_last[ray] : 0.f) * A.alphainv_last[ray];
  float acc_inv_s = 0.f;
  for (int top = n_kept; top > 0; top -= FGS_WAVE) {
    const int cnt = top < FGS_WAVE ? top : FGS_WAVE;
    const int k_local = top - 1 - lane;
    const bool act = lane < cnt;
    const bool in_chain = act && k_local < n_active;   // inside the second chain's [i_start, i_end)
    const int64_t rec = rec0 + (act ? k_local : 0);
    float gw = 0.f, w = 0.f, tt = 0.f, alpha = 0.f;
    if (in_chain) {
      gw = A.g_weights[s_off + k_local];
      w = A.a_weight[rec];
      tt = A.a_T[rec];
      alpha = A.a_alpha[rec];
    }
    float my_back = 0.f;
    for (int j = 0; j < cnt; ++j) {  // lanes outside the chain carry gw = w = 0: no effect on the recurrence
      if (lane == j) my_back = back_cum;
      back_cum = fmaf(fgs_bcast_lane(gw, j), fgs_bcast_lane(w, j), back_cum);
    }
    float gq[4] = {0.f, 0.f, 0.f, 0.f};   // (d sdf, d gradient xyz) of this lane's sample
    if (in_chain) {
      const double den = (double)(1.f - alpha) + 1e-10;
      const float g_alpha = (float)((double)(gw * tt) - (double)my_back / den);
      const AlphaGrad ag = neus_alpha_bwd(g_alpha, A.a_sdf[rec], A.a_grad[3 * rec], A.a_grad[3 * rec + 1],
                                          A.a_grad[3 * rec + 2], vx, vy, vz, A.dist, A.inv_s);
      gq[0] = ag.d_sdf; gq[1] = ag.dgx; gq[2] = ag.dgy; gq[3] = ag.dgz;
      acc_inv_s += ag.d_inv_s;
    }
    if (act && A.g_gradient) {
#pragma unroll
      for (int c = 0; c < 3; ++c) gq[1 + c] += A.g_gradient[3 * (s_off + k_local) + c];
    }
    const bool has = act && !(gq[0] == 0.f && gq[1] == 0.f && gq[2] == 0.f && gq[3] == 0.f);
    float fx = 0.f, fy = 0.f, fz = 0.f;
    if (has) {
      const float dist_s = A.stepdist * (float)A.a_step[rec];
      const PointIdx p = fgs_point_to_index(fmaf(rs.dir[0], dist_s, rs.start[0]), fmaf(rs.dir[1], dist_s, rs.start[1]),
                                            fmaf(rs.dir[2], dist_s, rs.start[2]), A.geom.lo, A.geom.hi, sd);
      fx = p.fx; fy = p.fy; fz = p.fz;
    }
    // Lane-transposed scatter into the voxel-interleaved buffer d4[X][Y][Z][4] (channel 0: d smoothed-sdf, 1..3: d gradient
    // volume).  Memory-side float atomics are priced per 64-byte line per wave-instruction: with lane = sample, each of the
    // 32 (corner, channel) instructions touched up to 64 lines (measured 165 us).  Here one instruction serves TWO
    // samples, lane = (sample half, corner, channel): the 8 floats of a (z, z+1) corner pair x 4 channels are contiguous,
    // so a sample costs ~5 line requests instead of 32.  Same products w[k] * g as fgs_tri_scatter.
    unsigned long long pend = __ballot(has);
    const int half = lane >> 5, k = (lane >> 2) & 7, ch = lane & 3;
    while (pend) {
      const int j0 = __builtin_ctzll(pend);
      pend &= pend - 1;
      int j1 = -1;
      if (pend) {
        j1 = __builtin_ctzll(pend);
        pend &= pend - 1;
      }
      const int src = half ? (j1 < 0 ? j0 : j1) : j0;
      const float sfx = __shfl(fx, src), sfy = __shfl(fy, src), sfz = __shfl(fz, src);
      const float g0 = __shfl(gq[0], src), g1 = __shfl(gq[1], src), g2 = __shfl(gq[2], src), g3 = __shfl(gq[3], src);
      if (half && j1 < 0) continue;   // odd count: the upper half idles in the last round (uniform per half-wave)
      const float go = ch == 0 ? g0 : (ch == 1 ? g1 : (ch == 2 ? g2 : g3));
      // this lane's corner weight, the expression fgs_tri_setup uses for w[k]: (z-term * y-term) * x-term
      const float flx = fgs_safe_floor(sfx), fly = fgs_safe_floor(sfy), flz = fgs_safe_floor(sfz);
      const float wx = (k >> 2) ? sfx - flx : (flx + 1.f) - sfx;
      const float wy = ((k >> 1) & 1) ? sfy - fly : (fly + 1.f) - sfy;
      const float wz = (k & 1) ? sfz - flz : (flz + 1.f) - sfz;
      const float wk = (wz * wy) * wx;
      const int x = (int)flx + (k >> 2), y = (int)fly + ((k >> 1) & 1), z = (int)flz + (k & 1);
      if (go != 0.f && fgs_in(x, A.geom.X) && fgs_in(y, A.geom.Y) && fgs_in(z, A.geom.Z))
        atomicAdd(A.d_grid4 + ((((int64_t)x * A.geom.Y + y) * A.geom.Z + z) << 2) + ch, wk * go);
    }
  }
  if (A.g_inv_s) fgs_wave_atomic_sum(acc_inv_s, A.g_inv_s);
}

SceneGeom geom_make(const float *lo, const float *hi, int X, int Y, int Z, float voxel_size) {
  SceneGeom g;
  for (int c = 0; c < 3; ++c) { g.lo[c] = lo[c]; g.hi[c] = hi[c]; }
  g.X = X; g.Y = Y; g.Z = Z; g.voxel_size = voxel_size;
  return g;
}

}  // namespace

FGS_API int fgs_march_coarse_fwd(const float *rays_o, const float *rays_d, const float *viewdirs, int64_t n_rays,
                                 const float *xyz_min_host, const float *xyz_max_host, int X, int Y, int Z, float near,
                                 float far, float stepdist, const float *sdf_smooth, const float *gradvol, const float *vol4,
                                 float dist, float inv_s, float thres, const float *mask_grid, const float *mask_min_host,
                                 const float *mask_max_host, int mX, int mY, int mZ, float mask_thres,
                                 const uint8_t *inc_world, int iX, int iY, int iZ, const float *inc_scale_host,
                                 const float *inc_shift_host, int max_steps, int *a_step, float *a_alpha, float *a_T,
                                 float *a_weight, float *a_sdf, float *a_grad, int *a_surv, int *surv_slot, int64_t *n_alive,
                                 int64_t *n_surv, int64_t *n_inbbox, float *alphainv_last, const fgs_dyn_t *dyn, fgs_stream_t stream) {
  FGS_REQUIRE(n_rays >= 0 && n_rays < ((int64_t)1 << 31), FGS_E_RANGE, "fgs_march_coarse_fwd: n_rays=%lld", (long long)n_rays);
  if (n_rays == 0) return 0;
  FGS_REQUIRE(rays_o && rays_d && viewdirs && xyz_min_host && xyz_max_host && sdf_smooth && gradvol && a_step && a_alpha &&
                  a_T && a_weight && a_sdf && a_grad && a_surv && surv_slot && n_alive && n_surv && n_inbbox && alphainv_last,
              FGS_E_INVALID, "fgs_march_coarse_fwd: null pointer");
  FGS_REQUIRE(X > 1 && Y > 1 && Z > 1 && max_steps > 0 && stepdist > 0.f && thres > 0.f, FGS_E_INVALID,
              "fgs_march_coarse_fwd: bad geometry or fast_color_thres <= 0");
  CoarseArgs A;
  A.rays_o = rays_o; A.rays_d = rays_d; A.viewdirs = viewdirs; A.n_rays = n_rays;
  A.geom = geom_make(xyz_min_host, xyz_max_host, X, Y, Z, 0.f);
  A.near = near; A.far = far; A.stepdist = stepdist; A.sdf_smooth = sdf_smooth; A.gradvol = gradvol;
  FGS_REQUIRE((reinterpret_cast<uintptr_t>(vol4) & 15) == 0, FGS_E_INVALID, "fgs_march_coarse_fwd: vol4 must be 16-byte aligned");
  A.vol4 = reinterpret_cast<const float4 *>(vol4);
  A.dist = dist; A.inv_s = inv_s; A.inv_s_dev = fgs_dyn_inv_s(dyn); A.thres = thres;
  A.mask_grid = mask_grid; A.mask_geom = A.geom; A.mask_thres = mask_thres;
  if (mask_grid) {
    FGS_REQUIRE(mask_min_host && mask_max_host && mX > 1 && mY > 1 && mZ > 1, FGS_E_INVALID, "fgs_march_coarse_fwd: bad mask cache");
    A.mask_geom = geom_make(mask_min_host, mask_max_host, mX, mY, mZ, 0.f);
  }
  A.inc.world = inc_world; A.inc.sx = iX; A.inc.sy = iY; A.inc.sz = iZ;
  for (int c = 0; c < 3; ++c) { A.inc.scale[c] = 0.f; A.inc.shift[c] = 0.f; }
  if (inc_world) {
    FGS_REQUIRE(inc_scale_host && inc_shift_host && iX > 0 && iY > 0 && iZ > 0, FGS_E_INVALID, "fgs_march_coarse_fwd: bad inc mask");
    for (int c = 0; c < 3; ++c) { A.inc.scale[c] = inc_scale_host[c]; A.inc.shift[c] = inc_shift_host[c]; }
  }
  A.max_steps = max_steps;
  A.a_step = a_step; A.a_alpha = a_alpha; A.a_T = a_T; A.a_weight = a_weight; A.a_sdf = a_sdf; A.a_grad = a_grad;
  A.a_surv = a_surv; A.surv_slot = surv_slot; A.n_alive = n_alive; A.n_surv = n_surv; A.n_inbbox = n_inbbox;
  A.alphainv_last = alphainv_last;
  hipLaunchKernelGGL(k_march_coarse_fwd, dim3(fgs_blocks(n_rays * FGS_WAVE)), dim3(FGS_BLOCK), 0, fgs_s(stream), A);
  FGS_LAUNCH_OK("fgs_march_coarse_fwd");
  return 0;
}

FGS_API int fgs_march_coarse_bwd(const float *rays_o, const float *rays_d, const float *viewdirs, int64_t n_rays,
                                 const float *xyz_min_host, const float *xyz_max_host, int X, int Y, int Z, float near,
                                 float far, float stepdist, float dist, float inv_s, int max_steps, const int *a_step,
                                 const float *a_alpha, const float *a_T, const float *a_weight, const float *a_sdf,
                                 const float *a_grad, const int64_t *n_alive, const int64_t *n_surv, const int64_t *surv_off,
                                 const float *alphainv_last, const float *g_weights, const float *g_last,
                                 const float *g_gradient, float *d_grid4, float *g_inv_s, const fgs_dyn_t *dyn,
                                 fgs_stream_t stream) {
  FGS_REQUIRE(n_rays >= 0 && n_rays < ((int64_t)1 << 31), FGS_E_RANGE, "fgs_march_coarse_bwd: n_rays=%lld", (long long)n_rays);
  if (n_rays == 0) return 0;
  FGS_REQUIRE(rays_o && rays_d && viewdirs && xyz_min_host && xyz_max_host && a_step && a_alpha && a_T && a_weight && a_sdf &&
                  a_grad && n_alive && n_surv && surv_off && alphainv_last && g_weights && d_grid4,
              FGS_E_INVALID, "fgs_march_coarse_bwd: null pointer");
  CoarseBwdArgs A;
  A.rays_o = rays_o; A.rays_d = rays_d; A.viewdirs = viewdirs; A.n_rays = n_rays;
  A.geom = geom_make(xyz_min_host, xyz_max_host, X, Y, Z, 0.f);
  A.near = near; A.far = far; A.stepdist = stepdist; A.dist = dist; A.inv_s = inv_s; A.inv_s_dev = fgs_dyn_inv_s(dyn); A.max_steps = max_steps;
  A.a_step = a_step; A.a_alpha = a_alpha; A.a_T = a_T; A.a_weight = a_weight; A.a_sdf = a_sdf; A.a_grad = a_grad;
  A.n_alive = n_alive; A.n_surv = n_surv; A.surv_off = surv_off; A.alphainv_last = alphainv_last;
  A.g_weights = g_weights; A.g_last = g_last; A.g_gradient = g_gradient;
  A.d_grid4 = d_grid4;
  A.g_inv_s = g_inv_s;
  hipLaunchKernelGGL(k_march_coarse_bwd, dim3(fgs_blocks(n_rays * FGS_WAVE)), dim3(FGS_BLOCK), 0, fgs_s(stream), A);
  FGS_LAUNCH_OK("fgs_march_coarse_bwd");
  return 0;
}
