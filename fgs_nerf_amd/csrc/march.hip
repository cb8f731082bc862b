// march.hip -- the fused front half of nerf.forward_fine (model/nerf.py:776-833), one wavefront per ray.
//
// Reference chain being fused: sample_pts_on_rays (6 launches + 2 cumsums + .item()) -> 3 boolean compactions ->
// optional mask cache -> grid_sample (sdf) + 6-tap grid_sample (gradient) over ALL in-bbox samples -> NeuS alpha
// (~12 elementwise launches) -> alpha > thres compaction (7 arrays) -> alpha2weight (1 thread per ray) ->
// weights > thres compaction (8 arrays).
//
// Here a 64-lane wavefront owns a ray and walks it 64 steps at a time: lane l evaluates step base+l (point, bbox test,
// mask cache, SDF trilerp + 6 clamped taps, alpha).  The transmittance recurrence over the lanes with alpha > thres is
// replayed in the reference's exact sequential double->float order (v_readlane broadcast, as csrc/compositing.hip),
// a ballot of "T < 1e-3" ends the ray (samples behind an opaque surface are never evaluated), and the two threshold
// compactions become ballot/popcount ranks.  Nothing of size M_total is ever written: the kernel emits only the
// "alive" records (alpha > thres, up to and including the terminating sample), which is exactly the segment
// [i_start, i_end) that the reference's alpha2weight backward walks, plus per-ray counts.
//
// The backward kernel walks the same records per ray: alpha2weight backward (exact reverse chain), NeuS-alpha
// backward, then scatters d sdf (8 corners) and d gradient (6 taps x 8 corners) into sdf.grad with fp32 atomics.
#include "fgs_march.h"

namespace {

struct MarchArgs {
  const float *rays_o, *rays_d, *viewdirs;
  int64_t n_rays;
  SceneGeom geom;
  float near, far, stepdist;
  const float *sdf;
  float dist;    // stepsize * voxel_size (fp32), model/nerf.py:795
  float inv_s;   // 1 / s_val (fp32 division), model/nerf.py:522
  const float *inv_s_dev;   // fgs_dyn_t.inv_s: read inv_s from the device (schedule table of a captured step)
  float thres;   // fast_color_thres
  // optional mask cache (model/nerf.py:1192-1209): max-pooled sdf_mask grid with its own bbox
  const float *mask_grid;
  SceneGeom mask_geom;
  float mask_thres;
  int max_steps;  // per-ray stride of the record arrays (>= any ray's n_steps)
  // alive records [n_rays * max_steps]
  int *a_step;
  float *a_alpha, *a_T, *a_weight, *a_sdf, *a_grad;  // a_grad: 3 per record
  int *a_surv;     // rank among this ray's survivors, or -1
  int *surv_slot;  // [n_rays * max_steps]: k-th survivor of the ray -> local alive index
  // per ray
  int64_t *n_alive, *n_surv, *n_inbbox;
  float *alphainv_last;
};

template <bool COUNT_ONLY>
__global__ __launch_bounds__(FGS_BLOCK) void k_march_fine_fwd(MarchArgs A) {
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

  float T_cum = 1.f;
  int alive_base = 0, surv_base = 0, inb_count = 0;
  bool terminated = false;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

  for (int base = 0; base < n_steps && !terminated; base += FGS_WAVE) {
    const int s = base + lane;
    const bool valid = s < n_steps;
    const float dist_s = A.stepdist * (float)s;
    const float px = fmaf(rs.dir[0], dist_s, rs.start[0]);
    const float py = fmaf(rs.dir[1], dist_s, rs.start[1]);
    const float pz = fmaf(rs.dir[2], dist_s, rs.start[2]);
    const bool outb = (A.geom.lo[0] > px) | (A.geom.lo[1] > py) | (A.geom.lo[2] > pz) | (A.geom.hi[0] < px) |
                      (A.geom.hi[1] < py) | (A.geom.hi[2] < pz);
    bool in = valid && !outb;
    inb_count += __popcll(__ballot(in));
    if (A.mask_grid && in) {
      const GridDesc md = fgs_sdf_desc(A.mask_geom);
      const PointIdx mp = fgs_point_to_index(px, py, pz, A.mask_geom.lo, A.mask_geom.hi, md);
      in = fgs_tri_sample(A.mask_grid, md, 0, fgs_tri_setup(mp.fx, mp.fy, mp.fz)) >= A.mask_thres;
    }
    float alpha = 0.f;
    SdfSample sv = {0.f, 0.f, 0.f, 0.f};
    if (in) {
      sv = fgs_sdf_value_grad_nb(A.sdf, A.geom, px, py, pz);   // bit-identical to fgs_sdf_value_grad, 12 loads not 56
      alpha = neus_alpha(sv.sdf, sv.gx, sv.gy, sv.gz, vx, vy, vz, A.dist, A.inv_s);
    }
    const bool m1 = in && (A.thres > 0.f ? alpha > A.thres : true);
    if (COUNT_ONLY) {  // length of the reference's `alpha > thres` list for this ray (no termination): see fgs_march_count
      alive_base += __popcll(__ballot(m1));
      continue;
    }
    unsigned long long bal = __ballot(m1);
    // exact sequential transmittance chain over the m1 lanes (render_utils_kernel.cu:591-601)
    float my_T = 1.f;
    int last = 64;
    while (bal) {
      const int j = __builtin_ctzll(bal);
      const float aj = fgs_bcast_lane(alpha, j);
      if (lane == j) my_T = T_cum;
      T_cum = (float)((double)T_cum * (1. - (double)aj));
      if (fgs_uniform((double)T_cum < 1e-3 ? 1 : 0)) {
        terminated = true;
        last = j;
        break;
      }
      bal &= bal - 1;
    }
    const bool alive = m1 && lane <= last;
    const float w = my_T * alpha;
    const bool surv = alive && (A.thres > 0.f ? w > A.thres : true);
    const unsigned long long alive_bal = __ballot(alive), surv_bal = __ballot(surv);
    if (alive) {
      const int a_local = alive_base + __popcll(alive_bal & lt_mask);
      const int64_t rec = rec0 + a_local;
      A.a_step[rec] = s;
      A.a_alpha[rec] = alpha;
      A.a_T[rec] = my_T;
      A.a_weight[rec] = w;
      A.a_sdf[rec] = sv.sdf;
      A.a_grad[3 * rec + 0] = sv.gx;
      A.a_grad[3 * rec + 1] = sv.gy;
      A.a_grad[3 * rec + 2] = sv.gz;
      int rank = -1;
      if (surv) {
        rank = surv_base + __popcll(surv_bal & lt_mask);
        A.surv_slot[rec0 + rank] = a_local;
      }
      A.a_surv[rec] = rank;
    }
    alive_base += __popcll(alive_bal);
    surv_base += __popcll(surv_bal);
  }
  if (lane == 0) {
    A.n_alive[ray] = alive_base;
    A.n_inbbox[ray] = inb_count;
    if (!COUNT_ONLY) {
      A.n_surv[ray] = surv_base;
      A.alphainv_last[ray] = T_cum;
    }
  }
}

// ---- survivor list: position t in [0, M_s) -> (ray, record) and the per-survivor arrays of the result dict --------
struct CompactArgs {
  int64_t n_rays, n_surv_total;
  const int64_t *m_dev;     // fgs_dyn_t.row_count: n_surv_total is then the capacity of the output arrays
  const int64_t *surv_off;  // [n_rays + 1] exclusive scan of n_surv
  int max_steps;
  const int *surv_slot, *a_step;
  const float *a_alpha, *a_weight, *a_sdf, *a_grad;
  const float *rays_o, *rays_d;
  SceneGeom geom;
  float near, far, stepdist;
  // outputs [M_s]
  int64_t *ray_id, *step_id;
  int *rec_idx;
  float *weights, *alpha, *sdf, *gradient, *pts;
};

__global__ __launch_bounds__(FGS_BLOCK) void k_surv_compact(CompactArgs C) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= fgs_rows(C.n_surv_total, C.m_dev)) return;
  int64_t lo = 0, hi = C.n_rays;  // largest r with surv_off[r] <= t and a non-empty segment
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (C.surv_off[mid] <= t) lo = mid; else hi = mid;
  }
  const int64_t r = lo;
  const int k = (int)(t - C.surv_off[r]);
  const int64_t rec = r * C.max_steps + C.surv_slot[r * C.max_steps + k];
  const int s = C.a_step[rec];
  C.ray_id[t] = r;
  C.step_id[t] = s;
  C.rec_idx[t] = (int)(rec - r * C.max_steps);
  C.weights[t] = C.a_weight[rec];
  C.alpha[t] = C.a_alpha[rec];
  C.sdf[t] = C.a_sdf[rec];
  C.gradient[3 * t + 0] = C.a_grad[3 * rec + 0];
  C.gradient[3 * t + 1] = C.a_grad[3 * rec + 1];
  C.gradient[3 * t + 2] = C.a_grad[3 * rec + 2];
  const float o[3] = {C.rays_o[3 * r], C.rays_o[3 * r + 1], C.rays_o[3 * r + 2]};
  const float d[3] = {C.rays_d[3 * r], C.rays_d[3 * r + 1], C.rays_d[3 * r + 2]};
  const RaySetup rs = ray_setup(o, d, C.geom, C.near, C.far, C.stepdist);
  const float dist_s = C.stepdist * (float)s;
#pragma unroll
  for (int c = 0; c < 3; ++c) C.pts[3 * t + c] = fmaf(rs.dir[c], dist_s, rs.start[c]);
}

// ---- backward over the alive records of each ray ------------------------------------------------------------------
struct MarchBwdArgs {
  const float *rays_o, *rays_d, *viewdirs;
  int64_t n_rays;
  SceneGeom geom;
  float near, far, stepdist, dist, inv_s;
  const float *inv_s_dev;
  int max_steps;
  const int *a_step, *a_surv;
  const float *a_alpha, *a_T, *a_weight, *a_sdf, *a_grad;
  const int64_t *n_alive, *surv_off;
  const float *alphainv_last;
  // incoming gradients
  const float *g_weights;   // [M_s] d loss / d weights (all paths into the weights)
  const float *g_last;      // [n_rays] d loss / d alphainv_last, or null
  const float *g_sdf;       // [M_s] from the feature path (center sdf), or null
  const float *g_gradient;  // [M_s,3] from the feature path (normal, reflection, gradient feature), or null
  float *grad_sdf_grid;     // [X,Y,Z] accumulated with atomics
  float *tot_sdf, *tot_grad;  // [M_s], [M_s,3]: if non-null, survivors' totals are written here instead of scattered
  int prio;                   // s_setprio level (FGS_PRIO_MARCH_BWD): the kernel runs beside k_mlp_wgrad, see fused.py _wgrad
  float *g_inv_s;             // device float accumulating d loss / d inv_s (s_learn, model/nerf.py:512-522), or null
};

__global__ __launch_bounds__(FGS_BLOCK) void k_march_fine_bwd(MarchBwdArgs A) {
  fgs_setprio(A.prio);
  if (A.inv_s_dev) A.inv_s = *A.inv_s_dev;
  const int64_t ray = (int64_t)blockIdx.x * (FGS_BLOCK / FGS_WAVE) + fgs_uniform((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  if (ray >= A.n_rays) return;
  const int n_alive = (int)fgs_uniform(A.n_alive[ray]);
  if (n_alive == 0) return;
  const int64_t rec0 = ray * A.max_steps;
  const int64_t s_off = fgs_uniform(A.surv_off[ray]);
  // survivors this ray has in the survivor arrays: all of them, unless the step overflowed its capacity and the guard cut the
  // offsets (fgs_count_guard) -- then the ranks beyond the cut have no slot and must not be touched
  const int n_slots = (int)(fgs_uniform(A.surv_off[ray + 1]) - s_off);
  const float o[3] = {A.rays_o[3 * ray], A.rays_o[3 * ray + 1], A.rays_o[3 * ray + 2]};
  const float d[3] = {A.rays_d[3 * ray], A.rays_d[3 * ray + 1], A.rays_d[3 * ray + 2]};
  const float vx = A.viewdirs[3 * ray], vy = A.viewdirs[3 * ray + 1], vz = A.viewdirs[3 * ray + 2];
  const RaySetup rs = ray_setup(o, d, A.geom, A.near, A.far, A.stepdist);
  const GridDesc gd = fgs_sdf_desc(A.geom);

  // alpha2weight backward, back to front (render_utils_kernel.cu:671-675)
  float back_cum = (A.g_last ? A.g_last[ray] : 0.f) * A.alphainv_last[ray];
  float acc_inv_s = 0.f;
  for (int top = n_alive; top > 0; top -= FGS_WAVE) {
    const int cnt = top < FGS_WAVE ? top : FGS_WAVE;
    const int a_local = top - 1 - lane;
    const bool act = lane < cnt;
    const int64_t rec = rec0 + (act ? a_local : 0);
    float gw = 0.f, w = 0.f, tt = 0.f, alpha = 0.f;
    int rank = -1;
    if (act) {
      rank = A.a_surv[rec];
      if (rank >= n_slots) rank = -1;
      gw = (rank >= 0) ? A.g_weights[s_off + rank] : 0.f;   // non-survivors were dropped before any consumer
      w = A.a_weight[rec];
      tt = A.a_T[rec];
      alpha = A.a_alpha[rec];
    }
    float my_back = 0.f;
    for (int j = 0; j < cnt; ++j) {
      if (lane == j) my_back = back_cum;
      back_cum = fmaf(fgs_bcast_lane(gw, j), fgs_bcast_lane(w, j), back_cum);
    }
    if (!act) continue;
    const double den = (double)(1.f - alpha) + 1e-10;
    const float g_alpha = (float)((double)(gw * tt) - (double)my_back / den);

    // NeuS alpha backward (autograd of model/nerf.py:525-543)
    const float sdf = A.a_sdf[rec];
    const float gx = A.a_grad[3 * rec], gy = A.a_grad[3 * rec + 1], gz = A.a_grad[3 * rec + 2];
    const float true_cos = (vx * gx + vy * gy) + vz * gz;
    const float iter_cos = -(fmaxf(-true_cos, 0.f));
    const float half = iter_cos * A.dist * 0.5f;
    const float pc = sigmoidf_((sdf - half) * A.inv_s), nc = sigmoidf_((sdf + half) * A.inv_s);
    const float num = (pc - nc) + 1e-5f, dn = pc + 1e-5f, q = num / dn;
    float d_sdf = 0.f, d_cos = 0.f;
    if (q >= 0.f && q <= 1.f) {  // clip passes the gradient on the closed interval
      const float d_p = g_alpha / dn;
      const float d_c = -g_alpha * num / (dn * dn);
      const float d_prev = (d_p + d_c) * (pc * (1.f - pc));
      const float d_next = -d_p * (nc * (1.f - nc));
      d_sdf = (d_prev + d_next) * A.inv_s;
      acc_inv_s += d_prev * (sdf - half) + d_next * (sdf + half);
      const float d_half = (d_next - d_prev) * A.inv_s;
      const float d_iter = d_half * A.dist * 0.5f;
      d_cos = (true_cos < 0.f) ? d_iter : 0.f;  // iter_cos = cos where cos < 0, else 0
    }
    float dgx = d_cos * vx, dgy = d_cos * vy, dgz = d_cos * vz;
    if (rank >= 0) {
      if (A.g_sdf) d_sdf += A.g_sdf[s_off + rank];
      if (A.g_gradient) {
        dgx += A.g_gradient[3 * (s_off + rank) + 0];
        dgy += A.g_gradient[3 * (s_off + rank) + 1];
        dgz += A.g_gradient[3 * (s_off + rank) + 2];
      }
      if (A.tot_sdf) {  // survivors: hand the totals to fgs_sdf_scatter_surv, which combines them with the 24 taps on chip
        A.tot_sdf[s_off + rank] = d_sdf;
        A.tot_grad[3 * (s_off + rank) + 0] = dgx;
        A.tot_grad[3 * (s_off + rank) + 1] = dgy;
        A.tot_grad[3 * (s_off + rank) + 2] = dgz;
        continue;
      }
    }
    // scatter: sdf trilerp corners, then the six +/-1 voxel taps of the finite-difference gradient
    const float dist_s = A.stepdist * (float)A.a_step[rec];
    const float px = fmaf(rs.dir[0], dist_s, rs.start[0]);
    const float py = fmaf(rs.dir[1], dist_s, rs.start[1]);
    const float pz = fmaf(rs.dir[2], dist_s, rs.start[2]);
    const PointIdx p = fgs_point_to_index(px, py, pz, A.geom.lo, A.geom.hi, gd);
    if (d_sdf != 0.f) fgs_tri_scatter(A.grad_sdf_grid, gd, 0, fgs_tri_setup(p.fx, p.fy, p.fz), d_sdf);
    const float dg_zyx[3] = {dgz, dgy, dgx};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (dg_zyx[a] == 0.f) continue;
      const TapPoint tm = fgs_tap_point(p, gd, 2 * a, 1.0f), tp = fgs_tap_point(p, gd, 2 * a + 1, 1.0f);
      const float coef = (dg_zyx[a] / A.geom.voxel_size) / (tp.clamped - tm.clamped);
      fgs_tri_scatter(A.grad_sdf_grid, gd, 0, fgs_tri_setup(tp.fx, tp.fy, tp.fz), coef);
      fgs_tri_scatter(A.grad_sdf_grid, gd, 0, fgs_tri_setup(tm.fx, tm.fy, tm.fz), -coef);
    }
  }
  if (A.g_inv_s) fgs_wave_atomic_sum(acc_inv_s, A.g_inv_s);
}

SceneGeom make_geom(const float *lo, const float *hi, int X, int Y, int Z, float voxel_size) {
  SceneGeom g;
  for (int c = 0; c < 3; ++c) { g.lo[c] = lo[c]; g.hi[c] = hi[c]; }
  g.X = X; g.Y = Y; g.Z = Z; g.voxel_size = voxel_size;
  return g;
}

}  // namespace

FGS_API int fgs_march_fine_fwd(const float *rays_o, const float *rays_d, const float *viewdirs, int64_t n_rays,
                               const float *xyz_min_host, const float *xyz_max_host, int X, int Y, int Z, float voxel_size,
                               float near, float far, float stepdist, const float *sdf, float dist, float inv_s, float thres,
                               const float *mask_grid, const float *mask_min_host, const float *mask_max_host, int mX, int mY,
                               int mZ, float mask_thres, int max_steps, int *a_step, float *a_alpha, float *a_T,
                               float *a_weight, float *a_sdf, float *a_grad, int *a_surv, int *surv_slot, int64_t *n_alive,
                               int64_t *n_surv, int64_t *n_inbbox, float *alphainv_last, const fgs_dyn_t *dyn, fgs_stream_t stream) {
  FGS_REQUIRE(n_rays >= 0 && n_rays < ((int64_t)1 << 31), FGS_E_RANGE, "fgs_march_fine_fwd: n_rays=%lld", (long long)n_rays);
  if (n_rays == 0) return 0;
  FGS_REQUIRE(rays_o && rays_d && viewdirs && xyz_min_host && xyz_max_host && sdf && a_step && a_alpha && a_T && a_weight &&
                  a_sdf && a_grad && a_surv && surv_slot && n_alive && n_surv && n_inbbox && alphainv_last,
              FGS_E_INVALID, "fgs_march_fine_fwd: null pointer");
  FGS_REQUIRE(X > 1 && Y > 1 && Z > 1 && max_steps > 0 && stepdist > 0.f, FGS_E_INVALID, "fgs_march_fine_fwd: bad geometry");
  FGS_REQUIRE(n_rays * (int64_t)max_steps < ((int64_t)1 << 40), FGS_E_RANGE, "fgs_march_fine_fwd: record arrays too large");
  MarchArgs A;
  A.rays_o = rays_o; A.rays_d = rays_d; A.viewdirs = viewdirs; A.n_rays = n_rays;
  A.geom = make_geom(xyz_min_host, xyz_max_host, X, Y, Z, voxel_size);
  A.near = near; A.far = far; A.stepdist = stepdist; A.sdf = sdf; A.dist = dist; A.inv_s = inv_s; A.inv_s_dev = fgs_dyn_inv_s(dyn); A.thres = thres;
  A.mask_grid = mask_grid;
  A.mask_geom = A.geom;
  A.mask_thres = mask_thres;
  if (mask_grid) {
    FGS_REQUIRE(mask_min_host && mask_max_host && mX > 1 && mY > 1 && mZ > 1, FGS_E_INVALID, "fgs_march_fine_fwd: bad mask cache");
    A.mask_geom = make_geom(mask_min_host, mask_max_host, mX, mY, mZ, 0.f);
  }
  A.max_steps = max_steps;
  A.a_step = a_step; A.a_alpha = a_alpha; A.a_T = a_T; A.a_weight = a_weight; A.a_sdf = a_sdf; A.a_grad = a_grad;
  A.a_surv = a_surv; A.surv_slot = surv_slot; A.n_alive = n_alive; A.n_surv = n_surv; A.n_inbbox = n_inbbox;
  A.alphainv_last = alphainv_last;
  hipLaunchKernelGGL(k_march_fine_fwd<false>, dim3(fgs_blocks(n_rays * FGS_WAVE)), dim3(FGS_BLOCK), 0, fgs_s(stream), A);
  FGS_LAUNCH_OK("fgs_march_fine_fwd");
  return 0;
}

// Per-ray length of the reference's `alpha > fast_color_thres` list (model/nerf.py:802-810) WITHOUT early termination,
// and the in-bbox sample count: what the result-dict entry 'mask' (a bool per such sample) is sized by.  The fused
// forward never visits the samples behind the terminating one; this runs only when a caller asks for 'mask'.
FGS_API int fgs_march_count(const float *rays_o, const float *rays_d, const float *viewdirs, int64_t n_rays,
                            const float *xyz_min_host, const float *xyz_max_host, int X, int Y, int Z, float voxel_size,
                            float near, float far, float stepdist, const float *sdf, float dist, float inv_s, float thres,
                            const float *mask_grid, const float *mask_min_host, const float *mask_max_host, int mX, int mY,
                            int mZ, float mask_thres, int max_steps, int64_t *n_m1, int64_t *n_inbbox,
                            const fgs_dyn_t *dyn, fgs_stream_t stream) {
  FGS_REQUIRE(n_rays >= 0 && n_rays < ((int64_t)1 << 31), FGS_E_RANGE, "fgs_march_count: n_rays=%lld", (long long)n_rays);
  if (n_rays == 0) return 0;
  FGS_REQUIRE(rays_o && rays_d && viewdirs && xyz_min_host && xyz_max_host && sdf && n_m1 && n_inbbox, FGS_E_INVALID,
              "fgs_march_count: null pointer");
  FGS_REQUIRE(X > 1 && Y > 1 && Z > 1 && max_steps > 0 && stepdist > 0.f, FGS_E_INVALID, "fgs_march_count: bad geometry");
  MarchArgs A = {};
  A.rays_o = rays_o; A.rays_d = rays_d; A.viewdirs = viewdirs; A.n_rays = n_rays;
  A.geom = make_geom(xyz_min_host, xyz_max_host, X, Y, Z, voxel_size);
  A.near = near; A.far = far; A.stepdist = stepdist; A.sdf = sdf; A.dist = dist; A.inv_s = inv_s; A.inv_s_dev = fgs_dyn_inv_s(dyn); A.thres = thres;
  A.mask_grid = mask_grid;
  A.mask_geom = A.geom;
  A.mask_thres = mask_thres;
  if (mask_grid) {
    FGS_REQUIRE(mask_min_host && mask_max_host && mX > 1 && mY > 1 && mZ > 1, FGS_E_INVALID, "fgs_march_count: bad mask cache");
    A.mask_geom = make_geom(mask_min_host, mask_max_host, mX, mY, mZ, 0.f);
  }
  A.max_steps = max_steps;
  A.n_alive = n_m1; A.n_inbbox = n_inbbox;
  hipLaunchKernelGGL(k_march_fine_fwd<true>, dim3(fgs_blocks(n_rays * FGS_WAVE)), dim3(FGS_BLOCK), 0, fgs_s(stream), A);
  FGS_LAUNCH_OK("fgs_march_count");
  return 0;
}

FGS_API int fgs_surv_compact(int64_t n_rays, int64_t n_surv_total, const int64_t *surv_off, int max_steps,
                             const int *surv_slot, const int *a_step, const float *a_alpha, const float *a_weight,
                             const float *a_sdf, const float *a_grad, const float *rays_o, const float *rays_d,
                             const float *xyz_min_host, const float *xyz_max_host, int X, int Y, int Z, float near, float far,
                             float stepdist, int64_t *ray_id, int64_t *step_id, int *rec_idx, float *weights, float *alpha,
                             float *sdf, float *gradient, float *pts, const fgs_dyn_t *dyn, fgs_stream_t stream) {
  FGS_REQUIRE(n_rays > 0 && n_surv_total >= 0 && n_surv_total < ((int64_t)1 << 40), FGS_E_RANGE, "fgs_surv_compact: sizes");
  if (n_surv_total == 0) return 0;
  FGS_REQUIRE(surv_off && surv_slot && a_step && a_alpha && a_weight && a_sdf && a_grad && rays_o && rays_d && xyz_min_host &&
                  xyz_max_host && ray_id && step_id && rec_idx && weights && alpha && sdf && gradient && pts,
              FGS_E_INVALID, "fgs_surv_compact: null pointer");
  CompactArgs C;
  C.n_rays = n_rays; C.n_surv_total = n_surv_total; C.m_dev = fgs_dyn_rows(dyn); C.surv_off = surv_off; C.max_steps = max_steps;
  C.surv_slot = surv_slot; C.a_step = a_step; C.a_alpha = a_alpha; C.a_weight = a_weight; C.a_sdf = a_sdf; C.a_grad = a_grad;
  C.rays_o = rays_o; C.rays_d = rays_d;
  C.geom = make_geom(xyz_min_host, xyz_max_host, X, Y, Z, 0.f);
  C.near = near; C.far = far; C.stepdist = stepdist;
  C.ray_id = ray_id; C.step_id = step_id; C.rec_idx = rec_idx; C.weights = weights; C.alpha = alpha; C.sdf = sdf;
  C.gradient = gradient; C.pts = pts;
  hipLaunchKernelGGL(k_surv_compact, dim3(fgs_blocks(n_surv_total)), dim3(FGS_BLOCK), 0, fgs_s(stream), C);
  FGS_LAUNCH_OK("fgs_surv_compact");
  return 0;
}

FGS_API int fgs_march_fine_bwd(const float *rays_o, const float *rays_d, const float *viewdirs, int64_t n_rays,
                               const float *xyz_min_host, const float *xyz_max_host, int X, int Y, int Z, float voxel_size,
                               float near, float far, float stepdist, float dist, float inv_s, int max_steps,
                               const int *a_step, const int *a_surv, const float *a_alpha, const float *a_T,
                               const float *a_weight, const float *a_sdf, const float *a_grad, const int64_t *n_alive,
                               const int64_t *surv_off, const float *alphainv_last, const float *g_weights,
                               const float *g_last, const float *g_sdf, const float *g_gradient, float *grad_sdf_grid,
                               float *tot_sdf, float *tot_grad, float *g_inv_s, const fgs_dyn_t *dyn, fgs_stream_t stream) {
  FGS_REQUIRE(n_rays >= 0 && n_rays < ((int64_t)1 << 31), FGS_E_RANGE, "fgs_march_fine_bwd: n_rays=%lld", (long long)n_rays);
  if (n_rays == 0) return 0;
  FGS_REQUIRE(rays_o && rays_d && viewdirs && xyz_min_host && xyz_max_host && a_step && a_surv && a_alpha && a_T && a_weight &&
                  a_sdf && a_grad && n_alive && surv_off && alphainv_last && g_weights && grad_sdf_grid,
              FGS_E_INVALID, "fgs_march_fine_bwd: null pointer");
  MarchBwdArgs A;
  A.rays_o = rays_o; A.rays_d = rays_d; A.viewdirs = viewdirs; A.n_rays = n_rays;
  A.geom = make_geom(xyz_min_host, xyz_max_host, X, Y, Z, voxel_size);
  A.near = near; A.far = far; A.stepdist = stepdist; A.dist = dist; A.inv_s = inv_s; A.inv_s_dev = fgs_dyn_inv_s(dyn); A.max_steps = max_steps;
  A.a_step = a_step; A.a_surv = a_surv; A.a_alpha = a_alpha; A.a_T = a_T; A.a_weight = a_weight; A.a_sdf = a_sdf;
  A.a_grad = a_grad; A.n_alive = n_alive; A.surv_off = surv_off; A.alphainv_last = alphainv_last;
  A.g_weights = g_weights; A.g_last = g_last; A.g_sdf = g_sdf; A.g_gradient = g_gradient; A.grad_sdf_grid = grad_sdf_grid;
  FGS_REQUIRE(!tot_sdf == !tot_grad, FGS_E_INVALID, "fgs_march_fine_bwd: tot_sdf and tot_grad go together");
  A.tot_sdf = tot_sdf; A.tot_grad = tot_grad;
  // (beside k_mlp_wgrad this kernel is the long pole of its branch: 160-170 us against 27 alone.  Level 1 lets its vector
  // instructions go first: 1.820 -> 1.810 ms/step; levels 2 / 3, the same for the sdf scatter, fewer weight-gradient workgroups
  // or k0's Adam pass moved to the branch's end all measured within +-0.5 % or worse: scripts/r3_branch_sweep*.sh)
  static const int prio = fgs_env_int("FGS_PRIO_MARCH_BWD", 1);
  A.prio = prio;
  A.g_inv_s = g_inv_s;
  hipLaunchKernelGGL(k_march_fine_bwd, dim3(fgs_blocks(n_rays * FGS_WAVE)), dim3(FGS_BLOCK), 0, fgs_s(stream), A);
  FGS_LAUNCH_OK("fgs_march_fine_bwd");
  return 0;
}

// Exclusive prefix sum helper used between the march and the survivor kernels (defined in sampling.hip's scan).
