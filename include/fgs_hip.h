/*
 * fgs_hip.h -- C ABI of libfgs_hip.so, the MI355X (gfx950) implementation of the
 * FGS-NeRF voxel render / training hot path.
 *
 * Every entry point replaces one operator of the reference's native extensions or one
 * torch op chain on the path (citations: paths under the reference tree).  Conventions:
 *
 *   - plain C: raw DEVICE pointers + element counts, no torch types;
 *   - fp32 values, int64 indices, uint8 (0/1) masks, exactly the reference dtypes;
 *   - `stream` is a hipStream_t passed as void* (PyTorch-ROCm's current stream);
 *   - nothing allocates, frees or synchronises unless stated; outputs are caller-owned
 *     and fully overwritten (the zero/one initialisations of the reference wrappers
 *     are done inside);
 *   - return 0 on success, a positive hipError_t for a HIP failure, or a negative
 *     FGS_E_* validation code; fgs_last_error() returns a thread-local message.
 *   - xyz_min / xyz_max / scale / shift are DEVICE float[3] (the reference passes them
 *     as tensors), so no call needs a device->host read.
 */
#ifndef FGS_HIP_H
#define FGS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FGS_E_INVALID (-1) /* bad argument (null pointer, negative size, ...)            */
#define FGS_E_RANGE   (-2) /* size beyond what the kernel indexes (see fgs_limits)       */
#define FGS_E_NODEV   (-3) /* no HIP device / wrong architecture                         */

typedef void *fgs_stream_t;

/* ABI version: bumped whenever an entry point is added, removed or changes its argument list.  fgs_version() returns the
 * value the library was BUILT with; a host binding compares it with the value it was written against and refuses a stale
 * library (the Python binding: fgs_nerf_amd/_lib.py ABI_VERSION -> FgsError) instead of calling it with another argument
 * list.  1 = rounds 1-2 (never bumped, although the table changed); 3 = round 3; 4 = explicit fgs_dyn_t instead of the
 * thread-local setters; 5 = fgs_dyn_t carries the in-kernel wall-clock stamps of the matrix-core launches; 6 = fgs_box_mask_fill; 7 = fgs_fine_loss_fwd takes a scratch buffer; 8 = fgs_mlp_rc2_chain;
 * 9 = fgs_step_scalars_tick2; 10 = fgs_mlp_rc2_pack, fgs_mlp_rc2_chain(prepacked);
 * 11 = fgs_fine_render_loss. */
#define FGS_ABI_VERSION 14

const char *fgs_last_error(void);
int fgs_version(void);                       /* == FGS_ABI_VERSION of the build */
/* Fills name (<=255 chars + NUL), CU count, wavefront size, LDS bytes per CU. */
int fgs_device_info(int device, char *name, int name_len, int *cu_count, int *wave_size, int64_t *lds_bytes);

/* Device-resident values of a sync-free / graph-capturable step, handed EXPLICITLY to every entry point
 * that can read them (rounds 1-2 had thread-local setters instead -- fgs_set_row_count_ptr / fgs_set_inv_s_ptr /
 * fgs_set_dx0_compact: state that changed what every later call did, per host thread; removed in ABI version 4).
 *   row_count   : the survivor count M_s of a step lives in device memory (the last entry of the march kernels' survivor
 *                 offsets).  Entry points that take a per-survivor row count M treat the HOST value as the CAPACITY of their
 *                 buffers -- they size the grid for it -- and the kernels read the ACTUAL count min(*row_count, M) from the
 *                 device: the host never needs the number, nothing in a step waits for a device->host copy, the whole step can
 *                 be captured in a hipGraph.  Entry points that cannot honour a device-side count (fgs_gemm_f32's TN and
 *                 stream-K forms) return FGS_E_INVALID when one is given.
 *   inv_s       : the march kernels (fgs_march_*) read NeuS inv_s = 1 / s_val from this device float instead of their argument
 *                 (model/nerf.py:514,522: s_val follows the iteration number, which a captured step cannot pass by value).
 *   dx0_compact : the backward entries (fgs_feat_fine_bwd, fgs_sdf_scatter_surv, fgs_feat_coarse_bwd) read dX0 in COMPACT form --
 *                 the columns of the xyz and view-direction encodings (functions of the fixed ray inputs: no gradient is needed,
 *                 model/nerf.py:837-874) are absent: row = [k0 | sdf | taps | tap differences | gradient], pitch = that width
 *                 rounded up to 4, so that the caller's dX0 product only multiplies the weight columns that matter (52 of 106);
 *                 coarse stages: row = [k0 | reflect_emb | normal] (48 of 90 columns with the shipped coarse config).
 * `dyn` == NULL, or a member NULL / 0: the host arguments alone.  The pointed-to device values must stay valid while the
 * launches issued with them run. */
typedef struct fgs_dyn {
  const int64_t *row_count;
  const float *inv_s;
  int dx0_compact;
  /* Measurement only (no effect on results; NULL = off), honoured by the matrix-core entries fgs_mlp_rc_chain, fgs_mlp_wgrad and
   * fgs_gemm_f32: the launch writes 100 MHz wall-clock readings (s_memrealtime) of its workgroups into
   * stamps + (*stamp_step % stamp_slots) * stamp_stride (uint64 units; stamp_step NULL: slot 0) -- the chain and weight-gradient
   * kernels 8 words per workgroup (word 1 = start, word 3 resp. 5 = end; a region of 2048 words: workgroups >= 256 do not stamp), the tiled product two words
   * (~min start, max end, by atomicMax: zero-initialise).  A captured step thereby times its own launches: bench.py's roofline
   * figure comes from the replays of the timed region itself (a graph replay cannot carry HIP events). */
  unsigned long long *stamps;
  const int64_t *stamp_step;
  int64_t stamp_slots;
  int64_t stamp_stride;
} fgs_dyn_t;

/* Device-resident schedule of a captured training step (model/nerf_training.py:389-436, model/adam.py:205-221).  `table` is
 * [n_rows][n_cols] floats the host fills once per stage -- row = iteration, columns = whatever per-iteration scalars the
 * step's kernels read (Adam step sizes from fgs_adam_step_size, inv_s).  fgs_step_scalars_tick copies row
 * min(*counter, n_rows - 1) to out[0..n_cols) and increments *counter.
 * fgs_count_guard: `offsets` [n] are the per-ray survivor offsets of the march kernels, offsets[n-1] the survivor count.
 * flags[1] = (count > capacity), flags[0] |= flags[1] (sticky), *total += min(count, capacity) (total may be NULL), and every
 * offset above the capacity is cut to it, so the per-ray segments later kernels walk stay inside buffers of `capacity` rows.
 * The optimizer entry points below skip their update while *skip_dev != 0: a step whose survivor list did not fit changes
 * nothing and the host can redo it (it learns about it from flags[0] whenever it next looks). */
float fgs_adam_step_size(int step, float beta1, float beta2, float lr);      /* adam_upd_kernel.cu:72, all-float */
int fgs_step_scalars_tick(const float *table, int n_rows, int n_cols, int64_t *counter, float *out, int mirror_col,
                          float *mirror_dst, fgs_stream_t stream);
/* The same, and BEFORE the row is copied: *latch_dst = out[latch_col] and *latch_flag_dst = *latch_flag_src -- the previous
 * iteration's value of one scalar (an Adam step size) and of one flag (the skip flag), for an update of that iteration that the
 * caller issues at the head of this one (the feature grid's Adam pass beside the next forward march).  NULL: not latched. */
int fgs_step_scalars_tick2(const float *table, int n_rows, int n_cols, int64_t *counter, float *out, int mirror_col,
                           float *mirror_dst, int latch_col, float *latch_dst, const int *latch_flag_src, int *latch_flag_dst,
                           fgs_stream_t stream);
int fgs_count_guard(int64_t *offsets, int64_t n, int64_t capacity, int *flags, int64_t *total, fgs_stream_t stream);
/* The voxel-increment mask of one iteration (model/nerf.py:1078-1088: linspace lattice compared with a growing box,
 * model/nerf_training.py:286-291) rebuilt on the device from six index bounds {lo_x, hi_x, lo_y, hi_y, lo_z, hi_z} in device
 * memory (floats holding integers: columns of the schedule table; lo > hi = empty): mask [X][Y][Z] bytes, 1 inside. */
int fgs_box_mask_fill(unsigned char *mask, int X, int Y, int Z, const float *bounds6_dev, fgs_stream_t stream);
/* fgs_exclusive_scan_i64(in, n, out) + fgs_count_guard(out, n + 1, capacity, flags, total) as ONE launch (the survivor offsets
 * of a sync-free step: model/nerf.py:802-833's nonzero / cumsum without the host). */
int fgs_exclusive_scan_guard_i64(const int64_t *in, int64_t n, int64_t *out, int64_t capacity, int *flags, int64_t *total,
                                 fgs_stream_t stream);
/* fgs_adam_upd / fgs_adam_upd_multi with an optional skip flag and the step size read from device memory (one float per
 * call / per tensor).  step_size_dev == NULL: the step size is computed on the host from (step, lr) / (steps, lrs) exactly as
 * the host-scalar forms do -- the form a sync-free step that is NOT captured uses: host schedule, device-side skip flag. */
int fgs_adam_upd_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, const float *perlr, int64_t n,
                     const float *step_size_dev, int step, float lr, float beta1, float beta2, float eps, int mode,
                     const int *skip_dev, fgs_stream_t stream);
int fgs_adam_upd_multi_dev(int n_tensors, float *const *params, const float *const *grads, float *const *exp_avgs,
                           float *const *exp_avg_sqs, const int64_t *sizes, const float *const *step_size_dev,
                           const int *steps, const float *lrs, const int *masked, float beta1, float beta2, float eps,
                           const int *skip_dev, fgs_stream_t stream);

/* ---------------------------------------------------------------------------------
 * render_utils_cuda  (model/cuda/render_utils.cpp:170-184)
 * ------------------------------------------------------------------------------ */

/* infer_t_minmax  -- render_utils_kernel.cu:11-35,82-104 */
int fgs_infer_t_minmax(const float *rays_o, const float *rays_d, const float *xyz_min, const float *xyz_max,
                       float near, float far, int64_t n_rays, float *t_min, float *t_max, fgs_stream_t stream);

/* infer_n_samples -- render_utils_kernel.cu:37-55,106-121 */
int fgs_infer_n_samples(const float *rays_d, const float *t_min, const float *t_max, float stepdist,
                        int64_t n_rays, int64_t *n_samples, fgs_stream_t stream);

/* infer_ray_start_dir -- render_utils_kernel.cu:57-79,123-139 */
int fgs_infer_ray_start_dir(const float *rays_o, const float *rays_d, const float *t_min, int64_t n_rays,
                            float *rays_start, float *rays_dir, fgs_stream_t stream);

/* sample_pts_on_rays, first half -- render_utils_kernel.cu:203-212.
 * Writes t_min, t_max, n_steps [n_rays] and the EXCLUSIVE prefix sum of n_steps into
 * steps_cumsum [n_rays+1] (steps_cumsum[n_rays] = total sample count M, the value the
 * reference fetches with .item<int>()).  The caller reads steps_cumsum[n_rays] when it
 * needs exact-size outputs; the fused path never does. */
int fgs_sample_count(const float *rays_o, const float *rays_d, const float *xyz_min, const float *xyz_max,
                     float near, float far, float stepdist, int64_t n_rays,
                     int64_t *n_steps, float *t_min, float *t_max, int64_t *steps_cumsum, fgs_stream_t stream);

/* sample_pts_on_rays, second half -- render_utils_kernel.cu:144-194,213-241.
 * `capacity` is the row count of the output buffers; rows >= M are left untouched.
 * M is read from steps_cumsum[n_rays] on the device. */
int fgs_sample_emit(const float *rays_o, const float *rays_d, const float *xyz_min, const float *xyz_max,
                    float stepdist, int64_t n_rays, const float *t_min, const int64_t *steps_cumsum,
                    int64_t capacity, float *rays_pts, uint8_t *mask_outbbox, int64_t *ray_id, int64_t *step_id,
                    fgs_stream_t stream);

/* The batch selection of a training iteration (model/nerf_training.py:256-261: rgb_tr[sel], rays_o_tr[sel], rays_d_tr[sel],
 * viewdirs_tr[sel] -- four advanced-indexing gathers) in one launch: out[a][i][:] = src_a[sel[i]][:], out [4][n][3], src_a
 * [n_src][3], sel device int64 (clamped into range). */
/* dst[0..n) = src[0..n), float32, both pointers 16-byte aligned (4-byte for n < 4): a staged batch into a captured step's static
 * inputs, an iteration's loss scalar into the window's log -- as a kernel launch (the runtime's device-to-device blit costs more per
 * call than it moves). */
int fgs_copy_f32(const float *src, float *dst, int64_t n, fgs_stream_t stream);
int fgs_gather_batch(const int64_t *sel, int64_t n, int64_t n_src, const float *src0, const float *src1, const float *src2,
                     const float *src3, float *out, fgs_stream_t stream);

/* sample_ndc_pts_on_rays -- render_utils_kernel.cu:244-293 */
int fgs_sample_ndc_pts(const float *rays_o, const float *rays_d, const float *xyz_min, const float *xyz_max,
                       int64_t n_samples, int64_t n_rays, float *rays_pts, uint8_t *mask_outbbox, fgs_stream_t stream);

/* sample_bg_pts_on_rays -- render_utils_kernel.cu:300-360 */
int fgs_sample_bg_pts(const float *rays_o, const float *rays_d, const float *t_max, float bg_preserve,
                      int64_t n_samples, int64_t n_rays, float *rays_pts, fgs_stream_t stream);

/* maskcache_lookup -- render_utils_kernel.cu:373-424 (out fully written, 0 outside the volume) */
int fgs_maskcache_lookup(const uint8_t *world, const float *xyz, const float *xyz2ijk_scale, const float *xyz2ijk_shift,
                         int sz_i, int sz_j, int sz_k, int64_t n_pts, uint8_t *out, fgs_stream_t stream);

/* raw2alpha / raw2alpha_nonuni -- render_utils_kernel.cu:430-504.  interval_nonuni == NULL selects the
 * uniform-interval form. */
int fgs_raw2alpha(const float *density, float shift, float interval, const float *interval_nonuni, int64_t n_pts,
                  float *exp_d, float *alpha, fgs_stream_t stream);
/* raw2alpha_backward / _nonuni_backward -- render_utils_kernel.cu:506-574 */
int fgs_raw2alpha_bwd(const float *exp_d, const float *grad_back, float interval, const float *interval_nonuni,
                      int64_t n_pts, float *grad, fgs_stream_t stream);

/* alpha2weight -- render_utils_kernel.cu:576-651.  ray_id sorted ascending.  One wavefront per ray;
 * the sequential double-precision transmittance chain of the reference is kept bit for bit. */
int fgs_alpha2weight_fwd(const float *alpha, const int64_t *ray_id, int64_t n_pts, int64_t n_rays,
                         float *weight, float *T, float *alphainv_last, int64_t *i_start, int64_t *i_end,
                         fgs_stream_t stream);
/* alpha2weight_backward -- render_utils_kernel.cu:653-707 */
int fgs_alpha2weight_bwd(const float *alpha, const float *weight, const float *T, const float *alphainv_last,
                         const int64_t *i_start, const int64_t *i_end, int64_t n_pts, int64_t n_rays,
                         const float *grad_weights, const float *grad_last, float *grad, fgs_stream_t stream);

/* ---------------------------------------------------------------------------------
 * total_variation_cuda (model/cuda/total_variation.cpp:29-32)
 * ------------------------------------------------------------------------------ */

/* total_variation_add_grad (mask == NULL, total_variation_kernel.cu:13-35,69-98) and
 * total_variation_add_grad_new (mask != NULL, :38-66,101-133).  param/grad/mask are [1,C,X,Y,Z]
 * tensors sharing the element strides (sC,sX,sY,sZ): (X*Y*Z, Y*Z, Z, 1) for the reference's
 * channel-first layout, (1, Y*Z*C, Z*C, C) for this build's channel-last DenseGrid storage. */
int fgs_tv_add_grad(const float *param, float *grad, const float *mask, float wx, float wy, float wz, int dense_mode,
                    int64_t C, int64_t X, int64_t Y, int64_t Z, int64_t sC, int64_t sX, int64_t sY, int64_t sZ,
                    fgs_stream_t stream);

/* ---------------------------------------------------------------------------------
 * adam_upd_cuda (model/cuda/adam_upd.cpp:79-86)
 * ------------------------------------------------------------------------------ */

#define FGS_ADAM_DENSE  0 /* adam_upd            adam_upd_kernel.cu:8-23,60-83    */
#define FGS_ADAM_MASKED 1 /* masked_adam_upd     adam_upd_kernel.cu:25-40,85-108  */
#define FGS_ADAM_PERLR  2 /* adam_upd_with_perlr adam_upd_kernel.cu:42-58,110-133 */
int fgs_adam_upd(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, const float *perlr, int64_t n,
                 int step, float beta1, float beta2, float lr, float eps, int mode, fgs_stream_t stream);

/* The same update for many small tensors in ONE launch (MaskedAdam.step over the 16 MLP weights / biases:
 * model/adam.py:195-221 loops one kernel per tensor).  HOST tables of n_tensors entries; per-tensor step count, lr and
 * masked flag (skip_zero_grad); arithmetic identical to fgs_adam_upd modes 0 / 1. */
int fgs_adam_upd_multi(int n_tensors, float *const *params, const float *const *grads, float *const *exp_avgs,
                       float *const *exp_avg_sqs, const int64_t *sizes, const int *steps, const float *lrs,
                       const int *masked, float beta1, float beta2, float eps, fgs_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Brick-sparse view of a channel-last grid gradient [X][Y][Z][C] (X, Y, Z multiples of 4) for the multi-GPU gradient
 * exchange (no reference counterpart: the reference is single-GPU).  Bricks are 4x4x4 voxels, numbered
 * (bx * nby + by) * nbz + bz.  flags[b] = 1 iff brick b has a non-zero element; gather/scatter move the bricks listed
 * in idx[n] to / from a dense buffer [n][4][4][4][C]; scatter multiplies by `scale` (the 1/world_size of the average).
 * ------------------------------------------------------------------------------ */
int fgs_brick_flags(const float *grad, int C, int X, int Y, int Z, int *flags, fgs_stream_t stream);
int fgs_brick_gather(const float *grad, int C, int X, int Y, int Z, const int64_t *idx, int64_t n, float *buf,
                     fgs_stream_t stream);
int fgs_brick_scatter(float *grad, int C, int X, int Y, int Z, const int64_t *idx, int64_t n, const float *buf,
                      float scale, fgs_stream_t stream);
/* Occupancy from the sample points whose 8 trilinear corners a DenseGrid backward will write (a superset of the
 * non-zero bricks, known right after the forward): ORs 1 into flags[b]; the caller zeroes flags once per step.
 * xyz_min / xyz_max on the HOST; index mapping identical to fgs_trilerp_*. */
int fgs_brick_flags_pts(const float *pts, int64_t M, const float *xyz_min_host, const float *xyz_max_host, int X, int Y,
                        int Z, int *flags, const fgs_dyn_t *dyn, fgs_stream_t stream);
/* idx[0 .. *count) = ascending indices of the set flags, entirely on the device (idx holds `total` entries, count is a
 * device int64): the exchange is sized from a count fetched asynchronously, never from a blocking nonzero(). */
int fgs_brick_compact(const int *flags, int64_t total, int64_t *idx, int64_t *count, fgs_stream_t stream);
/* The exchange with its brick count in DEVICE memory (the captured multi-GPU step: the count that sizes the exchange never
 * reaches the host).  buf holds `capacity` rows of 64 C floats and the collective always carries all of them; the launches
 * cover `capacity` rows and use the first min(*count_dev, capacity): the gather zero-fills the rows behind them, the
 * scatter leaves their bricks alone.
 * fgs_brick_count_guard (one thread, after fgs_brick_compact + the union all-reduce of the flags, hence identical on every
 * rank): if *count_dev > capacity, or *sticky is already set: *sticky = flags[0] = flags[1] = 1 -- flags as in
 * fgs_count_guard: the optimizer entry points skip their update while flags[1] != 0, and an exchange overflow keeps it
 * raised (a truncated exchange leaves unconsumed gradient behind in the persistent buffer) until the host has reset the
 * buffer and cleared *sticky; *count_dev is cut to the capacity.  peer_skip (may be NULL): a device int holding the MAX over
 * ranks of each rank's own flags[1] (it travels with the flags' all-reduce); non-zero raises flags[0] and flags[1] for this
 * step, so that a step one rank has to skip (its survivor list overflowed) is skipped by every rank. */
int fgs_brick_gather_dev(const float *grad, int C, int X, int Y, int Z, const int64_t *idx, const int64_t *count_dev,
                         int64_t capacity, float *buf, fgs_stream_t stream);
int fgs_brick_scatter_dev(float *grad, int C, int X, int Y, int Z, const int64_t *idx, const int64_t *count_dev,
                          int64_t capacity, const float *buf, float scale, fgs_stream_t stream);
int fgs_brick_count_guard(int64_t *count_dev, int64_t capacity, int *flags, int *sticky, const int *peer_skip,
                          fgs_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Ray-dependent loss terms of one iteration -- model/nerf_training.py:308-327 (+ nerf.orientation_loss,
 * model/nerf.py:469-478) and their gradients, two launches each instead of the ~40 of the autograd graph.
 * weights5_host = {weight_main, weight_rgbper, weight_entropy_last, weight_orientation, sigmoid_rgb_loss}.
 * viewdirs are per RAY [N,3] (the per-sample view direction is viewdirs[ray_id]).  loss_out / grad_out: device floats.
 * fwd `scratch` (optional, NULL = atomics behind a memset): >= 1 + ceil(max(3 N, M) / 256) floats, first word zero when first
 * handed in and left zero: per-block sums added up in a fixed order by the last block to arrive -- one launch, a scalar that
 * does not depend on the order in which atomics land.
 * ------------------------------------------------------------------------------ */
int fgs_fine_loss_fwd(int64_t N, int64_t M, const float *rgb_marched, const float *sigmoid_rgb, const float *target,
                      const float *alphainv_cum, const float *weights, const float *normal, const float *raw_rgb,
                      const int64_t *ray_id, const float *viewdirs, const float *weights5_host, float *loss_out,
                      float *scratch, int64_t scratch_floats, const fgs_dyn_t *dyn, fgs_stream_t stream);
int fgs_fine_loss_bwd(int64_t N, int64_t M, const float *rgb_marched, const float *sigmoid_rgb, const float *target,
                      const float *alphainv_cum, const float *weights, const float *normal, const float *raw_rgb,
                      const int64_t *ray_id, const float *viewdirs, const float *weights5_host, const float *grad_out,
                      float *g_rgb_marched, float *g_sigmoid_rgb, float *g_last, float *g_normal, float *g_raw_rgb,
                      const fgs_dyn_t *dyn, fgs_stream_t stream);
/* fgs_composite_fwd + fgs_fine_loss_fwd + fgs_fine_loss_bwd + fgs_composite_bwd as ONE launch (csrc/losses.hip k_render_loss: one wave
 * per ray, two passes over its survivors) -- what a training step runs between its two MLP chains besides the 256 -> 3 head
 * (model/nerf.py:888-920, model/nerf_training.py:308-327).  rgb [M,3] = sigmoid(head output); surv_off [N + 1]; seed_dev: device
 * float d total / d loss (NULL: 1).  Outputs: the per-ray render, loss_out, and the gradients the backward pass starts from:
 * d_out [M,3] (w.r.t. the head's pre-sigmoid output), d_w [M], g_normal [M,3], g_last [N], g_rgb_marched [N,3].
 * scratch: >= 1 + ceil(N / 4) floats, first word zero when first handed in (left zero).  The scalar is summed in a fixed order
 * (bit-reproducible; another order than the separate launches'). */
int fgs_fine_render_loss(int64_t N, int64_t M, const int64_t *surv_off, const float *weights, const float *rgb, const float *normal,
                         const int64_t *step_id, float bg, float dist, const float *viewdirs, const float *target,
                         const float *alphainv_last, const float *weights5_host, const float *seed_dev, float *rgb_marched,
                         float *sigmoid_rgb, float *pre_rgb, float *pre_sig, float *normal_marched, float *depth, float *loss_out,
                         float *scratch, int64_t scratch_floats, float *d_out, float *d_w, float *g_normal, float *g_last,
                         float *g_rgb_marched, const fgs_dyn_t *dyn, fgs_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Trilinear grid lookup -- replaces F.grid_sample(grid[1,C,X,Y,Z], ind_norm, 'bilinear',
 * align_corners=True, zeros padding) as called by DenseGrid.forward (model/grid.py:49-59),
 * nerf.grid_sampler (model/nerf.py:654-657) and MaskCache.forward (model/nerf.py:1203-1209).
 * pts are WORLD coordinates [M,3]; the xyz -> [-1,1] -> index round trip of the reference is
 * reproduced operation by operation.  out / grad_out are [M,C] row-major.
 * Element strides as in fgs_tv_add_grad.
 * ------------------------------------------------------------------------------ */
int fgs_trilerp_fwd(const float *grid, int64_t C, int64_t X, int64_t Y, int64_t Z,
                    int64_t sC, int64_t sX, int64_t sY, int64_t sZ,
                    const float *xyz_min, const float *xyz_max, const float *pts, int64_t M, float *out,
                    fgs_stream_t stream);
/* Scatter-add of grad_out into grad_grid (same strides; NOT zeroed here, accumulates with fp32 atomics). */
int fgs_trilerp_bwd(float *grad_grid, int64_t C, int64_t X, int64_t Y, int64_t Z,
                    int64_t sC, int64_t sX, int64_t sY, int64_t sZ,
                    const float *xyz_min, const float *xyz_max, const float *pts, int64_t M, const float *grad_out,
                    fgs_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Axis taps of the 1-channel SDF grid -- nerf.sample_sdfs (model/nerf.py:597-637).
 * For each of M world points and each of K displacements (HOST float[K], K <= 8, in voxels):
 * six lookups at ind -/+ displace along z, y, x (index space, clamped to the volume).
 *   feat [M, 6K] : layout ((axis_zyx*2 + sign) * K + k), the reference's `feat.view(M, 6*K)`
 *   diff [M, 3K] : clamped index distance of each -/+ pair (layout axis_zyx * K + k), may be NULL;
 *                  the reference's finite difference is (feat[+] - feat[-]) / diff / voxel_size.
 * The grid is [1,1,X,Y,Z] contiguous.  Backward scatter-adds grad_feat into grad_grid (not zeroed).
 * ------------------------------------------------------------------------------ */
int fgs_sdf_taps_fwd(const float *grid, int64_t X, int64_t Y, int64_t Z, const float *xyz_min, const float *xyz_max,
                     const float *pts, int64_t M, const float *displace_host, int K, float *feat, float *diff,
                     fgs_stream_t stream);
int fgs_sdf_taps_bwd(float *grad_grid, int64_t X, int64_t Y, int64_t Z, const float *xyz_min, const float *xyz_max,
                     const float *pts, int64_t M, const float *displace_host, int K, const float *grad_feat,
                     fgs_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Tiny-MLP products -- replace the nn.Linear / autograd SGEMMs of rgbnet and refnet
 * (model/nerf.py:125-142,877,884,1009).  Exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32).
 *   FGS_GEMM_NT  C[m,n]  = sum_k A[m,k] B[n,k] (+ bias[n]) (ReLU if relu)     forward:  A = X[M,K], B = W[N,K]
 *   FGS_GEMM_NN  C[m,n]  = sum_k A[m,k] B[k,n], zeroed where mask[m,n] <= 0   dX:       A = dY[M,K], B = W[K,N]
 *   FGS_GEMM_TN  C[m,n] += sum_k A[k,m] B[k,n]  (split-K, fp32 atomic adds)   dW:       A = dY[K,M], B = X[K,N]
 * colsum (NT/NN, may be NULL): colsum[n] += sum_m C[m,n] after the epilogue (bias gradients).
 * Operands 16-byte aligned, leading dimensions multiples of 4 floats; NT: K % 4 == 0; NN: K, N % 4 == 0;
 * TN: M, N % 4 == 0.  M (NT/NN) and K (TN) -- the sample count -- are arbitrary.
 * workspace (may be NULL): fgs_gemm_workspace_bytes() bytes of device scratch, 16-byte aligned, whose LAST
 * resident-slot-count int32 words were zero when it was first handed in (simplest: zero all of it once).  With it the
 * NT / NN products run as a stream-K grid (the chunk-units of all tiles cut into one equal range per resident
 * workgroup), which removes the partially filled last round of tiles; without it, one workgroup per tile.  A workspace
 * must not be shared by launches that can overlap (one per stream).
 * ------------------------------------------------------------------------------ */
#define FGS_GEMM_NT 0
#define FGS_GEMM_NN 1
#define FGS_GEMM_TN 2
/* Backward of one Linear layer y = x W^T (+ b) in ONE launch (both products depend only on dY; together the split-K
 * workgroups of the weight gradient fill the partially occupied last round of data-gradient tiles):
 *   dX[M, K_in]      = (dY[M, N_out] . W[N_out, K_in]) zeroed where mask[M, K_in] <= 0 (mask may be NULL);
 *                      colsum[K_in] += column sums of dX (may be NULL)         -- FGS_GEMM_NN semantics
 *   dW[N_out, K_in] += dY^T . X[M, K_in]  (fp32 atomics: zero-initialise dW)   -- FGS_GEMM_TN semantics
 * Alignment rules as for fgs_gemm_f32; N_out and K_in multiples of 4. */
int fgs_linear_bwd_f32(int64_t M, int64_t N_out, int64_t K_in, const float *dY, int64_t lddy, const float *W, int64_t ldw,
                       const float *X, int64_t ldx, float *dX, int64_t lddx, const float *mask, int64_t ldm, float *colsum,
                       float *dW, int64_t lddw, fgs_stream_t stream);
/* The whole forward chain of width-256 Linear(+ReLU) layers (rgbnet followed by refnet, model/nerf.py:877-884) in ONE
 * persistent launch: a 512-thread workgroup per CU owns blocks of 64 sample rows and walks all layers with the activation
 * block resident in LDS, weight chunks prefetched across layer boundaries; every layer's output is also written to
 * outs[l] for the backward pass.  Bit-identical to issuing the layers one by one through fgs_gemm_f32 (same k order).
 *   layer 0 input : X0[M, k0]   (k0 = K[0] <= 128, multiple of 4)
 *   layer l input : the previous output (256 columns), followed by T[M, t_cols] for a layer with K[l] = 256 + t_cols
 *                   (t_cols <= 64, multiple of 4; T may be NULL when no layer appends)
 * W, ldw, K, bias, relu, outs, ldo: HOST arrays of n_layers (<= 8) entries; W[l] is [256, ldw[l]], bias[l] [256] or NULL,
 * outs[l] [M, ldo[l] >= 256]. */
int fgs_mlp_fwd_f32(int64_t M, int n_layers, const float *X0, int64_t ldx0, int k0, const float *T, int64_t ldt, int t_cols,
                    const float *const *W, const int64_t *ldw, const int *K, const float *const *bias, const int *relu,
                    float *const *outs, const int64_t *ldo, fgs_stream_t stream);
/* The MLP chains with the activations resident in REGISTERS (csrc/mlp_rc.hip): every product is computed transposed,
 * features x samples, so that a layer's accumulator registers are the next layer's matrix-core operand as they stand; the
 * weights stream through LDS by LDS-DMA.  One persistent launch (plus a small weight-packing launch) per chain.
 *   backward = 0:  x_0 = in0[M, in0_cols];  x_{l+1} = act_l([x_l | ext_l] . W_l^T + bias_l)      (nn.Linear forward)
 *   backward = 1:  g_0 = in0[M, n_out_0];   g_{l+1} = (g_l . W_l) with the elements whose bit in mask_bits_l is 0 zeroed
 * W_l is the nn.Linear weight [n_out][ldw] (n_in valid columns) in both directions (the packing transposes).  A layer's
 * input is the previous output (<= 256 columns), carried in registers; forward layers may append ext_cols <= 64 columns
 * read from `ext` (refnet's reflection encoding: reduction over up to 320 columns).  Every layer of ONE chain must produce
 * the same number of columns rounded up to 128 / 192 / 256 (one kernel instantiation per width: the fine stage's 256, the
 * coarse stages' 192 and 128); narrow products at the ends of the backward chain (dX0, the encoding columns of dZ) go
 * through fgs_gemm_f32 on the dY tensors the chain writes out.
 *   out / ldo / n_store : row-major copy of the layer output (first n_store columns, multiple of 4), or NULL
 *   mask_bits           : forward + relu: receives one bit per output element (1 = positive), [ceil(M / 32)][64] x 16 bytes,
 *                         in the kernel's register order (element e of a word at bit 31 - e); backward: the bits to apply (those the forward wrote for the layer
 *                         whose input gradient this is); NULL: none
 *   image_ws            : scratch for the packed weight images, fgs_mlp_rc_image_floats() floats, 16-byte aligned (its last
 *                         2 KB are a sink that lanes without a sample store into; never read)
 * Honours fgs_dyn_t.row_count (M = capacity).  The reduction order of a sum differs from fgs_gemm_f32's (pairs (k, k+4));
 * results are deterministic but not bit-identical to the LDS-resident chain. */
typedef struct fgs_rc_layer {
  const float *W; int64_t ldw; int n_out, n_in;
  const float *bias; int relu;
  void *mask_bits;
  float *out; int64_t ldo; int n_store;
  const float *ext; int64_t ld_ext; int ext_cols;
} fgs_rc_layer_t;
int64_t fgs_mlp_rc_image_floats(int backward, int n_layers, const fgs_rc_layer_t *layers);
int fgs_mlp_rc_chain(int backward, int64_t M, int n_layers, const fgs_rc_layer_t *layers, const float *in0, int64_t ld_in0,
                     int in0_cols, float *image_ws, int64_t image_ws_floats, const fgs_dyn_t *dyn, fgs_stream_t stream);
/* The same chains, second form (csrc/mlp_rc2.hip; width 256 only): the four waves of a workgroup SPLIT a layer's output features
 * (64 each) over a slab of up to 4 x 32 samples whose activations live in LDS, so a CU that is dealt 7 sample tiles costs 7
 * tile-times where the register-resident form costs 8 (its unit of work per SIMD is a whole 32-sample tile: 1.73 rounds cost 2 at the
 * bench's 57 K survivors).  Weights go from L2 straight into matrix-core operand registers in fragment order (no LDS ring).
 * Arguments as fgs_mlp_rc_chain, plus SIDE layers: `side` != 0 marks a narrow product (<= 64 output columns, no bias / activation)
 * of the CURRENT carried input that is written to `out` only and leaves the carried input alone -- the backward chain's
 * reflection-encoding columns of dZ and the compact dX0, which the first form left to fgs_gemm_f32.  `side` == 2 (forward): the
 * 256 -> 3 output head -- bias, sigmoid, the first n_store <= 4 columns stored one by one (any ldo).  Main layers store all 256
 * output columns (n_store = 256) or none.  mask_bits: one 32-bit word per (32-sample tile, wave, lane), [ceil(M / 32)][4][64] --
 * the same size as the first form's buffer, another layout (private to the two chains of one step).  Deterministic; not
 * bit-identical to the first form (another k order). */
typedef struct fgs_rc2_layer {
  const float *W; int64_t ldw; int n_out, n_in;
  const float *bias; int relu;
  void *mask_bits;
  float *out; int64_t ldo; int n_store;
  const float *ext; int64_t ld_ext; int ext_cols;
  int side;
} fgs_rc2_layer_t;
int64_t fgs_mlp_rc2_image_floats(int backward, int n_layers, const fgs_rc2_layer_t *layers);
int fgs_mlp_rc2_chain(int backward, int64_t M, int n_layers, const fgs_rc2_layer_t *layers, const float *in0, int64_t ld_in0,
                      int in0_cols, float *image_ws, int64_t image_ws_floats, int prepacked, const fgs_dyn_t *dyn,
                      fgs_stream_t stream);
/* The weight images of TWO chains -- a step's forward and backward chain; n_layers_b = 0: one -- in ONE launch, for
 * fgs_mlp_rc2_chain(..., prepacked = 1, ...) calls with the same layer lists while the weights are unchanged.  Only W / ldw /
 * n_out / n_in / side / ext_cols of the layers are read. */
int fgs_mlp_rc2_pack(int backward_a, int n_layers_a, const fgs_rc2_layer_t *layers_a, int in0_cols_a, float *image_ws_a,
                     int64_t image_ws_floats_a, int backward_b, int n_layers_b, const fgs_rc2_layer_t *layers_b, int in0_cols_b,
                     float *image_ws_b, int64_t image_ws_floats_b, fgs_stream_t stream);
/* Every weight and bias gradient of the MLPs in ONE launch (csrc/mlp_wgrad.hip): for each item
 *   dW[n_out, n_in] += dY[M, n_out]^T . X[M, n_in]      dbias[n_out] += column sums of dY   (dbias may be NULL)
 * with fp32 atomics (zero-initialise dW / dbias).  The samples are split over the chip, a workgroup holds a 256 x 256 block
 * of one dW in its accumulators and streams dY / X rows straight into MFMA operands.  n_out <= 256; n_in any (blocks of 256
 * columns); all matrices row-major, 4-byte aligned.  Honours fgs_dyn_t.row_count (M = capacity). */
typedef struct fgs_wgrad_item {
  const float *dY; int64_t ld_dy; int n_out;
  const float *X; int64_t ld_x; int n_in;
  float *dW; int64_t ld_dw;
  float *dbias;
} fgs_wgrad_item_t;
int fgs_mlp_wgrad(int64_t M, int n_items, const fgs_wgrad_item_t *items, const fgs_dyn_t *dyn, fgs_stream_t stream);
/* The same products without float atomics for the weights: every (workgroup, k-split wave group) stores its partial block into a
 * slice of `ws` and a second launch adds the slices to dW in slice order -- weight gradients bit-reproducible (bias sums keep
 * their atomics).  ws: 16-byte aligned, fgs_mlp_wgrad_ws_floats() floats always suffice; NULL or too small for the shapes at
 * hand: the atomic form. */
int64_t fgs_mlp_wgrad_ws_floats(void);
int fgs_mlp_wgrad_ws(int64_t M, int n_items, const fgs_wgrad_item_t *items, float *ws, int64_t ws_floats, const fgs_dyn_t *dyn,
                     fgs_stream_t stream);
/* Diagnostics for fgs_mlp_wgrad: while a device buffer of >= 2048 uint64 is set, workgroup w records into stamps[8 w ..]
 * the shader clock at its start [0], after its prologue [2], after its sample loop [3] and after issuing its flush [4], the
 * 100 MHz wall clock at start [1] and end [5], its chunk count [6] and its block index [7].  NULL switches it off. */
int fgs_mlp_wgrad_debug_stamps(unsigned long long *stamps);
/* Diagnostics for the chain kernels: while a device buffer of >= 2048 uint64 is set, workgroup b records into stamps[8 b ..]
 * {shader clock, 100 MHz wall clock} at its start and end -- the clock the chip holds inside the kernel is d(shader) /
 * d(wall) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6) -- and the shader cycles its first wave spent in the
 * accumulator initialisation, the reduction chunks, the layer epilogues and the input loads.  NULL switches it off. */
int fgs_mlp_rc_debug_stamps(unsigned long long *stamps);
/* The general form of the one-launch chain, also used for the BACKWARD data gradients (dY of the top layer in, transposed
 * weights, per layer the ReLU mask of the layer below = its saved input, and the column sums = that layer's bias gradient):
 *   W[l] [n_rows[l] <= 256, ldw[l]]: output column n uses weight row n (only the last layer may have fewer than 256 rows);
 *   mask[l] [M, ldm[l]] or NULL: output (m, n) is zeroed where mask <= 0;   colsum[l] [256] or NULL: += column sums;
 *   outs[l] receives the first n_store[l] columns (multiple of 4).  k0 <= 256 here. */
int fgs_mlp_chain_f32(int64_t M, int n_layers, const float *X0, int64_t ldx0, int k0, const float *T, int64_t ldt, int t_cols,
                      const float *const *W, const int64_t *ldw, const int *K, const int *n_rows, const float *const *bias,
                      const int *relu, const float *const *mask, const int64_t *ldm, float *const *colsum,
                      float *const *outs, const int64_t *ldo, const int *n_store, fgs_stream_t stream);
/* dst[i] [cols[i], ld_dst[i]] = transpose of src[i] [rows[i], ld_src[i]], i < n <= 8, one launch (HOST arrays). */
int fgs_transpose_multi(int n, const float *const *src, const int *rows, const int *cols, const int64_t *ld_src,
                        float *const *dst, const int64_t *ld_dst, fgs_stream_t stream);
/* dst[i] [rows[i], ld_dst[i]] = src[i] [rows[i], >= cols[i]] with columns cols[i] .. ld_dst[i]-1 zero-filled, i < n <= 8, one
 * launch (HOST arrays): the K-padded copies of the first-layer weights the narrow data-gradient products multiply by. */
int fgs_pad_cols_multi(int n, const float *const *src, const int *rows, const int *cols, const int64_t *ld_src,
                       float *const *dst, const int64_t *ld_dst, fgs_stream_t stream);
/* The same with the number of columns WRITTEN per destination row given apart from the pitch (cols[i] <= width[i] <= ld_dst[i];
 * NULL: the pitch): destinations may be column ranges of one wider matrix, i.e. a gather of column ranges in one launch (the
 * first rgbnet layer's weights without the columns of the xyz / view-direction encodings, see fgs_dyn_t.dx0_compact). */
int fgs_copy_cols_multi(int n, const float *const *src, const int *rows, const int *cols, const int64_t *ld_src,
                        float *const *dst, const int64_t *ld_dst, const int *width, fgs_stream_t stream);
/* Diagnostics: one matrix copied with the index expression fgs_pad_cols_multi's kernel had before fgs_copy_cols_multi existed
 * (dst[e], e < rows * ld_dst: the destination taken for a whole [rows, ld_dst] matrix).  Only for the canary test that shows this
 * expression overruns a column-slice destination; the caller owns rows * ld_dst floats behind dst. */
int fgs_debug_pad_cols_old_indexing(const float *src, int rows, int cols, int64_t ld_src, float *dst, int64_t ld_dst,
                                    fgs_stream_t stream);
int64_t fgs_gemm_workspace_bytes(void);
int fgs_gemm_f32(int op, int64_t M, int64_t N, int64_t K, const float *A, int64_t lda, const float *B, int64_t ldb,
                 float *C, int64_t ldc, const float *bias, int relu, const float *mask, int64_t ldm, float *colsum,
                 void *workspace, int64_t workspace_bytes, const fgs_dyn_t *dyn, fgs_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Fused fine-stage render path -- nerf.forward_fine (model/nerf.py:776-941) as a short kernel chain.
 * Scene geometry is passed BY VALUE from the host (xyz_min/xyz_max HOST float[3], grid dims, voxel_size); device
 * arrays are caller-allocated.  Record arrays are laid out [n_rays * max_steps] with
 * max_steps >= ceil(|xyz_max - xyz_min| / stepdist) + 2 (no ray can emit more samples).
 * ------------------------------------------------------------------------------ */

/* out[0..n] = exclusive prefix sum of in[0..n) (out[n] = total); one workgroup, any n. */
int fgs_exclusive_scan_i64(const int64_t *in, int64_t n, int64_t *out, fgs_stream_t stream);

/* One wavefront per ray: sample_pts_on_rays + in-bbox / mask-cache tests + SDF trilerp + 6-tap gradient +
 * NeuS alpha + `alpha > thres` + alpha2weight (exact sequential chain, early termination at T < 1e-3) +
 * `weights > thres`  (model/nerf.py:780-833; render_utils_kernel.cu:11-242,576-605).
 * Writes the "alive" records of each ray (alpha > thres, up to and including the terminating sample -- the
 * [i_start, i_end) segment of the reference): a_step, a_alpha, a_T, a_weight, a_sdf, a_grad[3], a_surv (rank among the
 * ray's survivors or -1), surv_slot (k-th survivor -> local alive index); per ray n_alive, n_surv, n_inbbox (in-bbox
 * samples visited before termination) and alphainv_last.  mask_grid == NULL disables the mask cache. */
int fgs_march_fine_fwd(const float *rays_o, const float *rays_d, const float *viewdirs, int64_t n_rays,
                       const float *xyz_min_host, const float *xyz_max_host, int X, int Y, int Z, float voxel_size,
                       float near, float far, float stepdist, const float *sdf, float dist, float inv_s, float thres,
                       const float *mask_grid, const float *mask_min_host, const float *mask_max_host, int mX, int mY,
                       int mZ, float mask_thres, int max_steps, int *a_step, float *a_alpha, float *a_T, float *a_weight,
                       float *a_sdf, float *a_grad, int *a_surv, int *surv_slot, int64_t *n_alive, int64_t *n_surv,
                       int64_t *n_inbbox, float *alphainv_last, const fgs_dyn_t *dyn, fgs_stream_t stream);

/* Same walk without early termination and without writing records: n_m1[r] = number of samples of ray r with
 * alpha > thres (the length of the reference's first compacted list, model/nerf.py:802-810), n_inbbox[r] = in-bbox
 * samples.  Used only to materialise the result-dict entry 'mask' on demand. */
int fgs_march_count(const float *rays_o, const float *rays_d, const float *viewdirs, int64_t n_rays,
                    const float *xyz_min_host, const float *xyz_max_host, int X, int Y, int Z, float voxel_size, float near,
                    float far, float stepdist, const float *sdf, float dist, float inv_s, float thres,
                    const float *mask_grid, const float *mask_min_host, const float *mask_max_host, int mX, int mY, int mZ,
                    float mask_thres, int max_steps, int64_t *n_m1, int64_t *n_inbbox, const fgs_dyn_t *dyn, fgs_stream_t stream);

/* Survivor position t in [0, M_s) -> ray (binary search in surv_off) and the per-survivor arrays of the result
 * dict: ray_id, step_id, weights, raw alpha, sdf, gradient[3], ray_pts[3]; rec_idx = local alive index. */
int fgs_surv_compact(int64_t n_rays, int64_t n_surv_total, const int64_t *surv_off, int max_steps, const int *surv_slot,
                     const int *a_step, const float *a_alpha, const float *a_weight, const float *a_sdf,
                     const float *a_grad, const float *rays_o, const float *rays_d, const float *xyz_min_host,
                     const float *xyz_max_host, int X, int Y, int Z, float near, float far, float stepdist,
                     int64_t *ray_id, int64_t *step_id, int *rec_idx, float *weights, float *alpha, float *sdf,
                     float *gradient, float *pts, const fgs_dyn_t *dyn, fgs_stream_t stream);

/* Backward of fgs_march_fine_fwd: alpha2weight backward (render_utils_kernel.cu:653-677) over the alive records,
 * NeuS-alpha backward, plus the per-survivor gradients arriving through the feature path (g_sdf [M_s], g_gradient
 * [M_s,3], may be NULL); scatter-adds into grad_sdf_grid [X,Y,Z] (not zeroed here). */
int fgs_march_fine_bwd(const float *rays_o, const float *rays_d, const float *viewdirs, int64_t n_rays,
                       const float *xyz_min_host, const float *xyz_max_host, int X, int Y, int Z, float voxel_size,
                       float near, float far, float stepdist, float dist, float inv_s, int max_steps, const int *a_step,
                       const int *a_surv, const float *a_alpha, const float *a_T, const float *a_weight,
                       const float *a_sdf, const float *a_grad, const int64_t *n_alive, const int64_t *surv_off,
                       const float *alphainv_last, const float *g_weights, const float *g_last, const float *g_sdf,
                       const float *g_gradient, float *grad_sdf_grid, float *tot_sdf, float *tot_grad, float *g_inv_s,
                       const fgs_dyn_t *dyn, fgs_stream_t stream);
/* g_inv_s (may be NULL; fgs_march_fine_bwd and fgs_march_coarse_bwd): a device float, zeroed by the caller, into which the
 * kernel accumulates d loss / d inv_s -- the gradient of a LEARNABLE NeuS sharpness (s_learn, model/nerf.py:512-522:
 * inv_s = 1 / s_val with s_val an nn.Parameter); the caller turns it into d s_val = -g_inv_s / s_val^2.
 * tot_sdf [M_s] / tot_grad [M_s,3] (both or neither): when given, the survivors' total gradients w.r.t. their sdf value
 * and sdf gradient vector are written there instead of being scattered, and fgs_sdf_scatter_surv combines them on chip
 * with the hierarchical-tap gradients (8x8x8 LDS brick per survivor, row-wise flush: ~30 atomic line requests per
 * survivor instead of ~250).  dX0 is the gradient w.r.t. the rgbnet input buffer; X0 the saved forward buffer. */
int fgs_sdf_scatter_surv(int64_t M, const float *pts, const float *xyz_min_host, const float *xyz_max_host, int X, int Y,
                         int Z, float voxel_size, const int *layout_i, const float *displace_host, const float *X0,
                         const float *dX0, const float *tot_sdf, const float *tot_grad, float *sdf_grad_grid,
                         const fgs_dyn_t *dyn, fgs_stream_t stream);

/* Per-survivor MLP inputs (model/nerf.py:835-883).  layout_i = {k0_dim, n_posfreq, n_viewfreq, n_reffreq,
 * use_viewdir, center_sdf, use_grad_norm, K, ldx0, off_ref, ldz}; displace_host = K sorted displacements (K <= 5).
 * X0 [M, ldx0] receives torch.cat([k0, xyz_emb, viewdirs_emb, sdf, all_feat, all_grad, gradient]) (zero padded);
 * Zbuf [M, ldz] receives the reflection encoding at columns [off_ref, off_ref + 3 + 6 n_reffreq) (zero padded above);
 * normal_out [M,3].  k0 grid strides (ksC,ksX,ksY,ksZ) as in fgs_trilerp_fwd. */
int fgs_feat_fine_fwd(int64_t M, const int64_t *ray_id, const float *pts, const float *sdf, const float *gradient,
                      const float *viewdirs, const float *xyz_min_host, const float *xyz_max_host, int X, int Y, int Z,
                      float voxel_size, const int *layout_i, const float *displace_host, const float *sdf_grid,
                      const float *k0_grid, int64_t ksC, int64_t ksX, int64_t ksY, int64_t ksZ, float *X0, float *Zbuf,
                      float *normal_out, const fgs_dyn_t *dyn, fgs_stream_t stream);
/* Backward: scatter-adds into sdf_grad_grid and k0_grad_grid, writes g_sdf [M] and g_gradient [M,3] for
 * fgs_march_fine_bwd.  g_normal [M,3] (direct gradient on the `normal` output) may be NULL.  The k0 scatter and the encoding
 * part are independent kernels: k0_grad_grid == NULL issues only the latter, g_sdf == g_gradient == NULL only the former. */
int fgs_feat_fine_bwd(int64_t M, const int64_t *ray_id, const float *pts, const float *sdf, const float *gradient,
                      const float *viewdirs, const float *xyz_min_host, const float *xyz_max_host, int X, int Y, int Z,
                      float voxel_size, const int *layout_i, const float *displace_host, const float *X0,
                      const float *Zbuf, const float *dX0, const float *dZ, const float *g_normal, float *sdf_grad_grid,
                      float *k0_grad_grid, int64_t ksC, int64_t ksX, int64_t ksY, int64_t ksZ, float *g_sdf,
                      float *g_gradient, const fgs_dyn_t *dyn, fgs_stream_t stream);

/* Last refnet Linear (W -> 3) + sigmoid (model/nerf.py:884): rgb[m,:] = sigmoid(R[m,:W] . V[3,W]^T + bias). */
int fgs_head_fwd(const float *R, int64_t ldr, int W, int64_t M, const float *V, const float *bias, float *rgb,
                 const fgs_dyn_t *dyn, fgs_stream_t stream);
/* d_out [M,3] (w.r.t. the pre-sigmoid output) -> dR = (d_out . V) * (R > 0), dV += d_out^T R, dbias += colsum(d_out),
 * dR_colsum[W] += colsum(dR) (the bias gradient of the layer that produced R; may be NULL).
 * `scratch`: fgs_head_bwd_scratch_floats(W) floats (uninitialised) -> per-workgroup partial sums + a second, atomic-free
 * reduction launch; NULL -> the sums go out as float atomics (1024 workgroups on the same 64 cache lines). */
int fgs_head_bwd(const float *R, int64_t ldr, int W, int64_t M, const float *V, const float *d_out, float *dR, float *dV,
                 float *dbias, float *dR_colsum, float *scratch, const fgs_dyn_t *dyn, fgs_stream_t stream);
int64_t fgs_head_bwd_scratch_floats(int W);

/* The three segment_coo sums + background + clamp (model/nerf.py:888-903), optional normal_marched / depth
 * (:905-920).  pre_rgb / pre_sig keep the un-clamped values for the backward pass. */
int fgs_composite_fwd(int64_t n_rays, const int64_t *surv_off, const float *weights, const float *rgb, const float *normal,
                      const int64_t *step_id, float bg, float dist, float *rgb_marched, float *sigmoid_rgb, float *pre_rgb,
                      float *pre_sig, float *normal_marched, float *depth, fgs_stream_t stream);
/* Gradients of the loss w.r.t. rgb_marched / sigmoid_rgb [n_rays,3], raw_rgb [M,3], weights [M] (each may be
 * NULL) -> d_out [M,3] (pre-sigmoid head output) and d_w [M]. */
int fgs_composite_bwd(int64_t M, const int64_t *ray_id, const float *weights, const float *rgb, const float *pre_rgb,
                      const float *pre_sig, const float *g_rgb_marched, const float *g_sigmoid_rgb, const float *g_raw_rgb,
                      const float *g_weights_direct, float bg, float *d_out, float *d_w, const fgs_dyn_t *dyn, fgs_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Coarse stages (nerf.forward_coarse, model/nerf.py:943-1075) -- SURVEY.md 8a row a6 and BASELINE config 3.
 * ------------------------------------------------------------------------------------------------------------------
 * Dense per-iteration volume operators.
 * fgs_smooth3d_*: nn.Conv3d(1,1,k,padding=k//2,padding_mode='replicate') with the frozen Gaussian taps of
 *   model/nerf.py:260-278 (`smooth_conv`, applied every forward at :969).  taps_host: k^3 floats on the HOST in torch's
 *   weight order; k odd, <= 7.  in/out [X,Y,Z] fp32, must not alias.  The backward is the exact adjoint: the same
 *   LDS-tiled kernel evaluated on the padded domain into `scratch` [(X+k-1)(Y+k-1)(Z+k-1) floats], then folded.
 * fgs_sdf_gradvol_*: neus_sdf_gradient (model/nerf.py:485-508), mode 0 = 'interpolate' (central difference, zero faces), mode
 *   1 = 'raw' (forward difference, zero last face); 'grad_conv' is three fgs_smooth3d_* passes with the reference's Sobel-like
 *   taps (model/nerf.py:224-247), one per component.  grad3 [3,X,Y,Z], central
 *   differences / 2 / voxel_size, zero on the two boundary faces of each axis.  bwd: d_sdf (+)= adjoint(d_grad3). */
int fgs_smooth3d_fwd(const float *in, int X, int Y, int Z, int k, const float *taps_host, float *out, fgs_stream_t stream);
/* The two adjoints read their incoming gradient through element strides, so they can consume the voxel-interleaved
 * [X,Y,Z,4] buffer fgs_march_coarse_bwd accumulates into without a de-interleaving pass:
 *   d_out(x,y,z)     = d_out  [((x*Y + y)*Z + z) * out_stride]                 (dense: 1;        interleaved: 4)
 *   d_grad3(c,x,y,z) = d_grad3[c * chan_stride + ((x*Y + y)*Z + z) * voxel_stride]  (dense: XYZ, 1; interleaved: 1, 4) */
int fgs_smooth3d_bwd(const float *d_out, int64_t out_stride, int X, int Y, int Z, int k, const float *taps_host,
                     float *scratch, float *d_in, fgs_stream_t stream);
/* vol4 (optional, with pack_sdf [X,Y,Z]): also writes the voxel-interleaved volume [X,Y,Z,4] = {pack_sdf, g_x, g_y, g_z}
 * that fgs_march_coarse_fwd samples with one 16-byte load per trilinear corner (pack_sdf = the smoothed SDF). */
int fgs_sdf_gradvol_fwd(const float *sdf, int X, int Y, int Z, float voxel_size, int mode, float *grad3,
                        const float *pack_sdf, float *vol4, fgs_stream_t stream);
/* d_grad3_b (optional): a second upstream gradient of the volume, dense [3,X,Y,Z], added to d_grad3 on the fly (the
 * smooth-gradient TV term's beside the march kernels': saves autograd's grid-sized addition of the two). */
int fgs_sdf_gradvol_bwd(const float *d_grad3, int64_t chan_stride, int64_t voxel_stride, int X, int Y, int Z,
                        float voxel_size, int mode, float *d_sdf, int accumulate, const float *d_grad3_b, fgs_stream_t stream);

/* Smooth-gradient TV term of nerf.density_total_variation (model/nerf.py:436-446; the shipped fine config adds it every
 * third iteration, the coarse configs every iteration) over the gradient volume g3 [3,X,Y,Z], value and gradient in one
 * LDS-tiled pass per channel:
 *   *loss_out = weight * mean_masked((tv_smooth_conv(g3).detach() - g3)^2) (+ *add_in_dev),    d_g3 = d(weight * mean) / d g3.
 * taps_host: the 27 taps of tv_smooth_conv (HOST, replicate padding); mask [X,Y,Z] uint8 (nonempty_mask) or NULL;
 * inv_count_dev: DEVICE scalar 1 / (elements in the mean) -- 3 * mask.sum() or 3 X Y Z -- so no host read is needed;
 * add_in_dev: device scalar added to the term (the loss so far: saves the caller an addition launch) or NULL.
 * The scalar is a fixed-order sum (one partial per workgroup, summed by the one that arrives last): bit-reproducible.
 * scratch: fgs_smooth_tv_scratch_floats(X, Y, Z) floats whose first word is zero when first handed in (left zero). */
int64_t fgs_smooth_tv_scratch_floats(int X, int Y, int Z);
int fgs_smooth_tv_loss(const float *g3, int X, int Y, int Z, const float *taps_host, const uint8_t *mask,
                       const float *inv_count_dev, float weight, const float *add_in_dev, float *scratch, int64_t scratch_floats,
                       float *loss_out, float *d_g3, fgs_stream_t stream);

/* Fused front half of forward_coarse (model/nerf.py:946-990), one wavefront per ray: sample_pts_on_rays, optional mask
 * cache (stage 'coarse' only, :952-959) and voxel-increment MaskGrid (:962-967; inc_world uint8 [iX,iY,iZ] with the
 * MaskGrid's xyz2ijk scale / shift on the host, NULL = none), trilinear lookups of the smoothed SDF grid [X,Y,Z] and of
 * the gradient volume [3,X,Y,Z], NeuS alpha, Alphas2Weights over every sample, `weights > thres` (thres > 0 required),
 * Alphas2Weights again over the kept list.  Record arrays as in fgs_march_fine_fwd, but holding the KEPT samples only
 * (n_surv[ray] of them, a_surv[rec] = its index); n_alive[ray] = how many of them the second chain reached (the rest
 * carry weight 0, T 1).  fgs_surv_compact then builds the flat result lists exactly as for the fine stage.
 * vol4 (optional, fgs_sdf_gradvol_fwd): the same four channels voxel-interleaved; used for the lookups when non-NULL
 * (bit-identical results, a quarter of the gather instructions). */
int fgs_march_coarse_fwd(const float *rays_o, const float *rays_d, const float *viewdirs, int64_t n_rays,
                         const float *xyz_min_host, const float *xyz_max_host, int X, int Y, int Z, float near, float far,
                         float stepdist, const float *sdf_smooth, const float *gradvol, const float *vol4, float dist,
                         float inv_s, float thres, const float *mask_grid, const float *mask_min_host, const float *mask_max_host, int mX, int mY,
                         int mZ, float mask_thres, const uint8_t *inc_world, int iX, int iY, int iZ,
                         const float *inc_scale_host, const float *inc_shift_host, int max_steps, int *a_step,
                         float *a_alpha, float *a_T, float *a_weight, float *a_sdf, float *a_grad, int *a_surv,
                         int *surv_slot, int64_t *n_alive, int64_t *n_surv, int64_t *n_inbbox, float *alphainv_last,
                         const fgs_dyn_t *dyn, fgs_stream_t stream);
/* Backward of the second Alphas2Weights + NeuS alpha + the two trilinear lookups: g_weights [M_s], g_last [n_rays]
 * (may be NULL), g_gradient [M_s,3] (gradient reaching the sampled gradient vectors through the features, may be NULL)
 * -> atomically accumulated into d_grid4 [X,Y,Z,4] (not zeroed here): per voxel (d smoothed-sdf, d gradient-volume x,y,z)
 * interleaved, so one atomic wave-instruction covers the contiguous corner pairs of two samples (5 cache-line requests
 * per sample instead of 32).  The first Alphas2Weights only selects samples and carries no gradient (:990). */
int fgs_march_coarse_bwd(const float *rays_o, const float *rays_d, const float *viewdirs, int64_t n_rays,
                         const float *xyz_min_host, const float *xyz_max_host, int X, int Y, int Z, float near, float far,
                         float stepdist, float dist, float inv_s, int max_steps, const int *a_step, const float *a_alpha,
                         const float *a_T, const float *a_weight, const float *a_sdf, const float *a_grad,
                         const int64_t *n_alive, const int64_t *n_surv, const int64_t *surv_off, const float *alphainv_last,
                         const float *g_weights, const float *g_last, const float *g_gradient, float *d_grid4, float *g_inv_s,
                         const fgs_dyn_t *dyn, fgs_stream_t stream);

/* Coarse-stage MLP operand rows X0 [M, ldx0] = torch.cat([k0, xyz_emb, reflect_emb, normal, viewdirs_emb]) (zero padded),
 * model/nerf.py:992-1009.  layout_i = {k0_dim, n_posfreq, n_viewfreq, n_reffreq, use_viewdir, ldx0}. */
int fgs_feat_coarse_fwd(int64_t M, const int64_t *ray_id, const float *pts, const float *gradient, const float *viewdirs,
                        const float *xyz_min_host, const float *xyz_max_host, int X, int Y, int Z, const int *layout_i,
                        const float *k0_grid, int64_t ksC, int64_t ksX, int64_t ksY, int64_t ksZ, float *X0,
                        float *normal_out, const fgs_dyn_t *dyn, fgs_stream_t stream);
/* dX0 [M, ldx0] (+ optional direct gradient g_normal [M,3]) -> scatter-add into k0_grad_grid, g_gradient [M,3] for
 * fgs_march_coarse_bwd. */
int fgs_feat_coarse_bwd(int64_t M, const int64_t *ray_id, const float *pts, const float *gradient, const float *viewdirs,
                        const float *xyz_min_host, const float *xyz_max_host, int X, int Y, int Z, const int *layout_i,
                        const float *X0, const float *dX0, const float *g_normal, float *k0_grad_grid, int64_t ksC,
                        int64_t ksX, int64_t ksY, int64_t ksZ, float *g_gradient, const fgs_dyn_t *dyn, fgs_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Marching cubes on a device-resident field -- replaces the host call mcubes.marching_cubes(u, threshold) of
 * model/extract_geometry.py:24 (PyMCubes, third party): vertices in INDEX coordinates as float64 [V,3], shared between
 * triangles; triangles int64 [T,3].  field: [X][Y][Z] float32; a corner is flagged when field < iso; triangle normals
 * point towards decreasing field values.  Tables: fgs-nerf_amd/mc_tables.py (tri_table int8 [256][16], ntri uint8 [256]),
 * device arrays.  Lattice points are handled 256 at a time (fgs_mc_num_blocks workgroups):
 *   fgs_mc_count : vflags [X*Y*Z] (bit a: the +a edge of the point crosses iso), per-workgroup vertex / triangle totals
 *   (caller)     : exclusive scans of the two total arrays (fgs_exclusive_scan_i64), totals read back, outputs allocated
 *   fgs_mc_emit  : vbase [X*Y*Z] uint32 scratch, vertices, triangles.  Vertex ids follow (lattice point, axis) order,
 *                  triangles follow cell order: the output is deterministic.
 * ------------------------------------------------------------------------------ */
int64_t fgs_mc_num_blocks(int X, int Y, int Z);
int fgs_mc_count(const float *field, int X, int Y, int Z, float iso, const uint8_t *ntri_table, uint8_t *vflags,
                 int64_t *block_vertices, int64_t *block_triangles, fgs_stream_t stream);
int fgs_mc_emit(const float *field, int X, int Y, int Z, float iso, const int8_t *tri_table, const uint8_t *ntri_table,
                const uint8_t *vflags, const int64_t *block_vertex_offset, const int64_t *block_triangle_offset,
                uint32_t *vbase, int64_t n_vertices, int64_t n_triangles, double *vertices, int64_t *triangles,
                fgs_stream_t stream);

/* The autograd-form total-variation losses (ori_tv configurations): total_variation(v, mask) of model/nerf.py:1212-1221 /
 * model/dvgo.py:420-428 as a value pass and a gradient pass (what the reference differentiates through diff -> abs ->
 * boolean index -> sum).  v: [1,C,X,Y,Z] float32 with element strides (channel-first or channel-last dense); mask:
 * [X][Y][Z] bytes, non-zero = inside, shared by all channels, or NULL.
 * value: the seven sums {S_x, S_y, S_z, sum(v), pairs_x, pairs_y, pairs_z} (S_a = sum over the pairs along axis a whose two
 * voxels are inside the mask of |v[i+1] - v[i]|, pairs_a = their number; fixed-order double sums: bit-reproducible) and, from
 * them on the device, in double:
 *     tv = (S_x + S_y + S_z) / 3 / den,  den = *count_dev with a mask (the mask.sum() of the tensor the caller holds), sum(v)
 *          without one (sic: model/nerf.py:1221)                                   [per_axis_mean = 0]
 *     tv = (S_x / pairs_x + S_y / pairs_y + S_z / pairs_z) / 3                     [per_axis_mean = 1: model/dvgo.py:420-428]
 *   *loss_out = (float)(scale * tv) (+ *add_in_dev),   w_out[0..2] = scale * d tv / d S_a,  w_out[3] = scale * d tv / d sum(v)
 *   (0 with a mask), sums_out[0..6] = the sums.  Any of loss_out / w_out / sums_out may be NULL.
 *   scratch: fgs_tv_loss_scratch_doubles() doubles whose first word is zero when first handed in (left zero).
 * grad: grad[j] = (accumulate ? grad[j] : 0) + u * (sum_a w[a] * (sign(v[j] - v[j-1_a]) [valid] - sign(v[j+1_a] - v[j])
 * [valid]) + w[3]), sign(0) = 0; w: 4 DEVICE floats (the value pass's w_out), u = *upstream_dev (d total / d term) or 1. */
int64_t fgs_tv_loss_scratch_doubles(void);
int fgs_tv_loss_value(const float *v, const unsigned char *mask, int64_t C, int64_t X, int64_t Y, int64_t Z, int64_t sC,
                      int64_t sX, int64_t sY, int64_t sZ, const int64_t *count_dev, int per_axis_mean, double scale,
                      const float *add_in_dev, double *scratch, int64_t scratch_doubles, float *loss_out, float *w_out,
                      double *sums_out, fgs_stream_t stream);
int fgs_tv_loss_grad(const float *v, const unsigned char *mask, int64_t C, int64_t X, int64_t Y, int64_t Z, int64_t sC,
                     int64_t sX, int64_t sY, int64_t sZ, const float *w, const float *upstream_dev, float *grad, int accumulate,
                     fgs_stream_t stream);

/* masked_adam_upd (adam_upd_kernel.cu:25-40) over a SET of 4x4x4-voxel bricks of a channel-last [X][Y][Z][C] grid (C a
 * multiple of 4; sides need not be multiples of 4), and self-cleaning: only the selected bricks are visited, elements with
 * grad == 0 are skipped as in the dense masked update (bit-identical results on the same gradient), and every consumed
 * gradient element is set back to zero, so a persistent gradient buffer needs no zero fill per step.
 * Selection: idx != NULL -> ascending brick indices (fgs_brick_compact), count from count_dev (device int64) when non-NULL,
 * else n_host; idx == NULL -> every brick whose entry of `flags` (one int per brick, fgs_brick_flags_pts on the step's
 * survivor points) is non-zero.  `flags`, when given, is cleared for the processed bricks.  Step size from step_size_dev
 * (device float) when non-NULL, else from (step, lr); while *skip_dev != 0 the gradient is consumed but nothing is updated. */
int fgs_adam_upd_bricks(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int C, int X, int Y, int Z,
                        const int64_t *idx, const int64_t *count_dev, int64_t n_host, int *flags, int step, float beta1,
                        float beta2, float lr, float eps, const float *step_size_dev, const int *skip_dev,
                        fgs_stream_t stream);

/* The same at voxel granularity: masks holds 64 bytes per 4x4x4-voxel brick, byte 16 x' + 4 y' + z' != 0 = "voxel (x', y', z')
 * of the brick holds a trilinear corner of a survivor point" (fgs_brick_masks_pts sets them; the caller zeroes the 16-byte
 * aligned buffer once); fgs_adam_upd_voxels walks the recorded voxels only, applies the masked update to their C floats,
 * zeroes the consumed gradient and clears the bytes.  C a multiple of 4, <= 64. */
int fgs_brick_masks_pts(const float *pts, int64_t M, const float *xyz_min_host, const float *xyz_max_host, int X, int Y,
                        int Z, unsigned char *masks, const fgs_dyn_t *dyn, fgs_stream_t stream);
int fgs_adam_upd_voxels(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int C, int X, int Y, int Z,
                        unsigned char *masks, int step, float beta1, float beta2, float lr, float eps,
                        const float *step_size_dev, const int *skip_dev, fgs_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Integrated directional encoding -- generate_ide_fn / integrated_dir_enc_fn (model/utils.py:515-574; built at
 * model/nerf.py:179, never evaluated by the reference's forward passes).  mat: [n_pow][n] coefficient matrix
 * (n_pow = l_max + 1 <= 17), ml: int [2][n] rows (m, l), n <= 36; xyz [M,3], kappa_inv [M]; out / g_out [M][2n]
 * (real parts, then imaginary parts).  All device pointers.
 * ------------------------------------------------------------------------------ */
int fgs_ide_fwd(const float *xyz, const float *kappa_inv, const float *mat, const int *ml, int n, int n_pow, int64_t M,
                float *out, fgs_stream_t stream);
int fgs_ide_bwd(const float *xyz, const float *kappa_inv, const float *mat, const int *ml, int n, int n_pow, int64_t M,
                const float *g_out, float *g_xyz, float *g_kappa_inv, fgs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FGS_HIP_H */
