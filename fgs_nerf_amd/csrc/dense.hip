// dense.hip -- the per-iteration full-volume operators of the coarse stages:
//   * 3-D smoothing of the SDF grid: nn.Conv3d(1, 1, k, padding=k//2, padding_mode='replicate') with frozen Gaussian
//     taps (model/nerf.py:260-278, applied every forward at :791/:969), forward and backward;
//   * the central-difference gradient volume neus_sdf_gradient(mode='interpolate') (model/nerf.py:485-494),
//     forward and backward.
// The smoothing is a direct k^3-tap convolution tiled through LDS (no separability assumed: the taps are whatever the
// Conv3d holds); its backward is the same kernel run as the adjoint on the padded domain followed by a fold of the
// replicate-padding halo -- deterministic, no atomics.  The gradient volume is a pure streaming stencil.
#include "fgs_common.h"

namespace {

constexpr int MAXK = 7;  // kernel side (the reference uses 3 and 5)

struct Conv3 {
  int X, Y, Z, k;
  float w[MAXK * MAXK * MAXK];  // taps [dx][dy][dz], cross-correlation order as torch stores them
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Tiled direct convolution: out[o] = sum_t w[t] * fetch(o + t - shift), t in [0,k)^3 in (dx, dy, dz) order.
//   REPL = true : fetch(q) = in[clamp(q)]               (forward, replicate padding, shift = k/2)
//   REPL = false: fetch(q) = q inside ? in[q] : 0       (adjoint on the padded domain, shift = k-1, flipped taps)
// One 256-thread block produces an 8 x 8 x 32 output tile from an LDS halo tile of (8+K-1)(8+K-1)(32+K-1) values
// (20 KB for K = 5); a thread owns 8 consecutive z outputs of one (x, y) column and slides an (8+K-1)-value register
// window over each of the K*K tap rows: (8+K-1) LDS reads per 8K FMAs.  The accumulation order per output is the plain
// (dx, dy, dz) fmaf chain, the same order the one-thread-per-voxel form used.
constexpr int CT_X = 8, CT_Y = 8, CT_Z = 32, CT_ZPT = 8;

// EPI = 1 turns the store into the smooth-gradient TV term of nerf.density_total_variation (model/nerf.py:436-446) for
// channel blockIdx.y of a [3,X,Y,Z] gradient volume:  err = conv(g).detach() - g,  loss += weight / count * sum(m err^2),
// out = d loss / d g = -2 weight / count * m * err   (the smoothed volume is detached in the reference).
struct TvEpi {
  const uint8_t *mask;      // [X,Y,Z] nonempty mask or null
  const float *inv_count;   // device scalar 1 / (number of elements in the mean)
  float weight;
  float *loss;              // device scalar: weight * mean (+ *add_in), written by the workgroup that arrives last
  const float *add_in;      // device scalar or null
  float *partials;          // one float per workgroup
  unsigned *counter;        // arrival counter (zero when handed in, left zero)
};

template <int K, bool REPL, int EPI = 0>
__global__ __launch_bounds__(FGS_BLOCK) void k_conv3d_tiled(const float *__restrict__ in, int nx, int ny, int nz,
                                                            Conv3 c /* X,Y,Z = OUTPUT dims */, int shift, int64_t es,
                                                            float *__restrict__ out, TvEpi tv) {
  if (EPI == 1) {   // one channel per grid row
    in += (int64_t)blockIdx.y * nx * ny * nz;
    out += (int64_t)blockIdx.y * nx * ny * nz;
  }
  constexpr int LX = CT_X + K - 1, LY = CT_Y + K - 1, LZU = CT_Z + K - 1;
  constexpr int LZ = LZU | 1;  // odd row pitch: the 8 (y) x 4 (z-group) lanes of a wave hit distinct banks
  __shared__ float tile[LX * LY * LZ];
  const int tiles_z = (c.Z + CT_Z - 1) / CT_Z, tiles_y = (c.Y + CT_Y - 1) / CT_Y;
  const int bz = blockIdx.x % tiles_z, by = (blockIdx.x / tiles_z) % tiles_y, bx = blockIdx.x / (tiles_z * tiles_y);
  const int ox0 = bx * CT_X, oy0 = by * CT_Y, oz0 = bz * CT_Z;
  // halo tile -> LDS: z-rows of LZU values, one (lx, ly) row per group of LZU consecutive work items; the fixed trip
  // count lets the compiler keep several global loads in flight per thread
  constexpr int TOTAL = LX * LY * LZU, ITERS = (TOTAL + FGS_BLOCK - 1) / FGS_BLOCK;
  constexpr int UNR = ITERS <= 16 ? ITERS : (ITERS + 1) / 2;       // (the whole halo in one or two batches of loads)
#pragma unroll UNR
  for (int it = 0; it < ITERS; ++it) {
    const int i = it * FGS_BLOCK + threadIdx.x;
    const int ic = i < TOTAL ? i : TOTAL - 1;
    const int lz = ic % LZU, ly = (ic / LZU) % LY, lx = ic / (LZU * LY);
    int qx = ox0 + lx - shift, qy = oy0 + ly - shift, qz = oz0 + lz - shift;
    const bool ok = REPL || (fgs_in(qx, nx) && fgs_in(qy, ny) && fgs_in(qz, nz));
    qx = clampi(qx, 0, nx - 1); qy = clampi(qy, 0, ny - 1); qz = clampi(qz, 0, nz - 1);
    const float v = in[(((int64_t)qx * ny + qy) * nz + qz) * es];   // address always valid: branch-free load (es: element stride)
    if (i < TOTAL) tile[(lx * LY + ly) * LZ + lz] = ok ? v : 0.f;
  }
  __syncthreads();
  const int tz = threadIdx.x & 3, ty = (threadIdx.x >> 2) & 7, tx = threadIdx.x >> 5;
  float acc[CT_ZPT];
#pragma unroll
  for (int j = 0; j < CT_ZPT; ++j) acc[j] = 0.f;
#pragma unroll 1
  for (int dx = 0; dx < K; ++dx) {
#pragma unroll
    for (int dy = 0; dy < K; ++dy) {
      const float *row = tile + ((tx + dx) * LY + (ty + dy)) * LZ + tz * CT_ZPT;
      float win[CT_ZPT + K - 1];
#pragma unroll
      for (int j = 0; j < CT_ZPT + K - 1; ++j) win[j] = row[j];
#pragma unroll
      for (int dz = 0; dz < K; ++dz) {
        const float w = c.w[(dx * K + dy) * K + dz];
#pragma unroll
        for (int j = 0; j < CT_ZPT; ++j) acc[j] = fmaf(w, win[j + dz], acc[j]);
      }
    }
  }
  const int x = ox0 + tx, y = oy0 + ty, z0 = oz0 + tz * CT_ZPT;
  if (EPI == 0) {
    if (x < c.X && y < c.Y) {
      float *o = out + ((int64_t)x * c.Y + y) * c.Z;
#pragma unroll
      for (int j = 0; j < CT_ZPT; ++j)
        if (z0 + j < c.Z) o[z0 + j] = acc[j];
    }
    return;
  }
  // smooth-gradient TV: the centre value g sits in the halo tile at offset K/2 on every axis
  float part = 0.f;
  if (x < c.X && y < c.Y) {
    const float ic = *tv.inv_count;
    const float ds = -2.f * tv.weight * ic;
    const float *crow = tile + ((tx + K / 2) * LY + (ty + K / 2)) * LZ + tz * CT_ZPT + K / 2;
    const int64_t base = ((int64_t)x * c.Y + y) * c.Z;
#pragma unroll
    for (int j = 0; j < CT_ZPT; ++j)
      if (z0 + j < c.Z) {
        const float m = (tv.mask && !tv.mask[base + z0 + j]) ? 0.f : 1.f;
        const float e = acc[j] - crow[j];
        out[base + z0 + j] = ds * m * e;
        part = fmaf(m * e, e, part);
      }
  }
  // the loss: one partial per workgroup, summed in a fixed order by the workgroup that arrives last (until round 4: one float
  // atomic per wave on ONE address -- 10 K of them were 125 of this launch's 142 us at 114^3, and the sum depended on their order)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off);
  __shared__ float wpart[FGS_BLOCK / FGS_WAVE];
  __shared__ int is_last;
  if ((threadIdx.x & 63) == 0) wpart[threadIdx.x >> 6] = part;
  __syncthreads();
  const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x, n_wg = gridDim.x * gridDim.y;
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < FGS_BLOCK / FGS_WAVE; ++w) t += wpart[w];
    __hip_atomic_store(tv.partials + wg, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = fgs_arrive_is_last(tv.counter, n_wg) ? 1 : 0;
  }
  __syncthreads();
  if (!is_last) return;
  double t = fgs_partials_sum<float, double>(tv.partials, threadIdx.x, n_wg, FGS_BLOCK);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
  __shared__ double fpart[FGS_BLOCK / FGS_WAVE];
  if ((threadIdx.x & 63) == 0) fpart[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    double sum = 0.0;
    for (int w = 0; w < FGS_BLOCK / FGS_WAVE; ++w) sum += fpart[w];
    *tv.loss = (float)(sum * (double)tv.weight * (double)*tv.inv_count) + (tv.add_in ? *tv.add_in : 0.f);
    *tv.counter = 0u;
  }
}

// Fold the adjoint computed on the padded domain [(n + 2r)^3, index p = q + r] back onto the grid:
// d_in[v] = sum of dP[q] over the q that replicate padding maps to v (q == v, plus the r halo layers on a boundary face).
__global__ __launch_bounds__(FGS_BLOCK) void k_fold_padded(const float *__restrict__ dP, int X, int Y, int Z, int r,
                                                           float *__restrict__ d_in) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t N = (int64_t)X * Y * Z;
  if (idx >= N) return;
  const int z = (int)(idx % Z), y = (int)((idx / Z) % Y), x = (int)(idx / ((int64_t)Z * Y));
  const int PY = Y + 2 * r, PZ = Z + 2 * r;
  const int x0 = (x == 0) ? 0 : x + r, x1 = (x == X - 1) ? X - 1 + 2 * r : x + r;
  const int y0 = (y == 0) ? 0 : y + r, y1 = (y == Y - 1) ? Y - 1 + 2 * r : y + r;
  const int z0 = (z == 0) ? 0 : z + r, z1 = (z == Z - 1) ? Z - 1 + 2 * r : z + r;
  float acc = 0.f;
  for (int px = x0; px <= x1; ++px)
    for (int py = y0; py <= y1; ++py) {
      const float *row = dP + ((int64_t)px * PY + py) * PZ;
      for (int pz = z0; pz <= z1; ++pz) acc += row[pz];
    }
  d_in[idx] = acc;
}

template <bool REPL>
void launch_conv(const float *in, int nx, int ny, int nz, const Conv3 &c, int shift, int64_t es, float *out, hipStream_t st) {
  const int64_t tiles = (int64_t)((c.X + CT_X - 1) / CT_X) * ((c.Y + CT_Y - 1) / CT_Y) * ((c.Z + CT_Z - 1) / CT_Z);
  const dim3 grid((unsigned)tiles), block(FGS_BLOCK);
  switch (c.k) {
    case 1: hipLaunchKernelGGL((k_conv3d_tiled<1, REPL>), grid, block, 0, st, in, nx, ny, nz, c, shift, es, out, TvEpi{}); break;
    case 3: hipLaunchKernelGGL((k_conv3d_tiled<3, REPL>), grid, block, 0, st, in, nx, ny, nz, c, shift, es, out, TvEpi{}); break;
    case 5: hipLaunchKernelGGL((k_conv3d_tiled<5, REPL>), grid, block, 0, st, in, nx, ny, nz, c, shift, es, out, TvEpi{}); break;
    default: hipLaunchKernelGGL((k_conv3d_tiled<7, REPL>), grid, block, 0, st, in, nx, ny, nz, c, shift, es, out, TvEpi{}); break;
  }
}

// g[0] = (s[x+1] - s[x-1]) / 2 / vs on 1 <= x <= X-2 (zero on the two faces), likewise g[1] along y, g[2] along z
// vol4 (optional): the voxel-interleaved copy [X][Y][Z][4] = {pack_s[v], g_x[v], g_y[v], g_z[v]} the coarse march samples
// with ONE 16-byte load per trilinear corner instead of four 4-byte gathers from four arrays (pack_s = the smoothed SDF).
// MODE 0 = 'interpolate' (model/nerf.py:490-494), MODE 1 = 'raw' (:501-505): g[0] = (s[x+1] - s[x]) / vs on x <= X-2, zero on
// the last face.
template <int MODE>
__global__ __launch_bounds__(FGS_BLOCK) void k_gradvol_fwd(const float *__restrict__ s, int X, int Y, int Z, float vs,
                                                           float *__restrict__ g, const float *__restrict__ pack_s,
                                                           float4 *__restrict__ vol4) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t N = (int64_t)X * Y * Z;
  if (idx >= N) return;
  const int z = (int)(idx % Z), y = (int)((idx / Z) % Y), x = (int)(idx / ((int64_t)Z * Y));
  const int64_t sx = (int64_t)Y * Z, sy = Z;
  float gx, gy, gz;
  if (MODE == 0) {
    gx = (x >= 1 && x <= X - 2) ? (s[idx + sx] - s[idx - sx]) / 2.f / vs : 0.f;
    gy = (y >= 1 && y <= Y - 2) ? (s[idx + sy] - s[idx - sy]) / 2.f / vs : 0.f;
    gz = (z >= 1 && z <= Z - 2) ? (s[idx + 1] - s[idx - 1]) / 2.f / vs : 0.f;
  } else {
    gx = (x <= X - 2) ? (s[idx + sx] - s[idx]) / vs : 0.f;
    gy = (y <= Y - 2) ? (s[idx + sy] - s[idx]) / vs : 0.f;
    gz = (z <= Z - 2) ? (s[idx + 1] - s[idx]) / vs : 0.f;
  }
  g[idx] = gx;
  g[N + idx] = gy;
  g[2 * N + idx] = gz;
  if (vol4) vol4[idx] = make_float4(pack_s[idx], gx, gy, gz);
}

// d_s[v] (+)= sum_axis ( dg_axis[v-1] * [v-1 interior] - dg_axis[v+1] * [v+1 interior] ) / 2 / vs          (MODE 0)
// d_s[v] (+)= sum_axis ( dg_axis[v-1] * [v >= 1] - dg_axis[v] * [v <= n-2] ) / vs                            (MODE 1)
// dg component c of voxel v lives at dg[c * sC + v * sV] ((XYZ, 1) for the dense [3,X,Y,Z] layout)
// dg2 (optional): a SECOND upstream gradient of the volume, dense [3,X,Y,Z], added to dg on the fly (the smooth-gradient TV
// term's: autograd would add the two with a grid-sized launch of its own, one operand strided)
template <int MODE>
__global__ __launch_bounds__(FGS_BLOCK) void k_gradvol_bwd(const float *__restrict__ dg, int64_t sC, int64_t sV, int X,
                                                           int Y, int Z, float vs, float *__restrict__ d_s, int accumulate,
                                                           const float *__restrict__ dg2) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t N = (int64_t)X * Y * Z;
  if (idx >= N) return;
  const int z = (int)(idx % Z), y = (int)((idx / Z) % Y), x = (int)(idx / ((int64_t)Z * Y));
  const int64_t sx = (int64_t)Y * Z, sy = Z;
  float acc = 0.f;
  // component c of voxel w: dg[c sC + w sV] (+ dg2[c N + w])
#define GV(c, w) (dg[(c) * sC + (w) * sV] + (dg2 ? dg2[(c) * N + (w)] : 0.f))
  if (MODE == 0) {
    // s[v] appears as "+" in g at v-1 (needs 1 <= v-1 <= n-2) and as "-" in g at v+1 (needs 1 <= v+1 <= n-2)
    if (x - 1 >= 1 && x - 1 <= X - 2) acc += GV(0, idx - sx) / 2.f / vs;
    if (x + 1 >= 1 && x + 1 <= X - 2) acc -= GV(0, idx + sx) / 2.f / vs;
    if (y - 1 >= 1 && y - 1 <= Y - 2) acc += GV(1, idx - sy) / 2.f / vs;
    if (y + 1 >= 1 && y + 1 <= Y - 2) acc -= GV(1, idx + sy) / 2.f / vs;
    if (z - 1 >= 1 && z - 1 <= Z - 2) acc += GV(2, idx - 1) / 2.f / vs;
    if (z + 1 >= 1 && z + 1 <= Z - 2) acc -= GV(2, idx + 1) / 2.f / vs;
  } else {
    // s[v] appears as "+" in g at v-1 (needs v-1 <= n-2, i.e. always, and v >= 1) and as "-" in g at v (needs v <= n-2)
    if (x >= 1) acc += GV(0, idx - sx) / vs;
    if (x <= X - 2) acc -= GV(0, idx) / vs;
    if (y >= 1) acc += GV(1, idx - sy) / vs;
    if (y <= Y - 2) acc -= GV(1, idx) / vs;
    if (z >= 1) acc += GV(2, idx - 1) / vs;
    if (z <= Z - 2) acc -= GV(2, idx) / vs;
  }
#undef GV
  d_s[idx] = accumulate ? d_s[idx] + acc : acc;
}

int make_conv(const char *who, int X, int Y, int Z, int k, const float *w_host, Conv3 *c) {
  if (X <= 0 || Y <= 0 || Z <= 0 || (int64_t)X * Y * Z >= ((int64_t)1 << 40))
    return fgs_set_error(FGS_E_RANGE, "%s: grid %dx%dx%d", who, X, Y, Z);
  if (k < 1 || k > MAXK || !(k & 1) || !w_host) return fgs_set_error(FGS_E_INVALID, "%s: odd kernel side 1..%d expected", who, MAXK);
  c->X = X; c->Y = Y; c->Z = Z; c->k = k;
  for (int i = 0; i < k * k * k; ++i) c->w[i] = w_host[i];
  return 0;
}

}  // namespace

FGS_API int fgs_smooth3d_fwd(const float *in, int X, int Y, int Z, int k, const float *taps_host, float *out,
                             fgs_stream_t stream) {
  Conv3 c;
  if (int e = make_conv("fgs_smooth3d_fwd", X, Y, Z, k, taps_host, &c)) return e;
  FGS_REQUIRE(in && out && in != out, FGS_E_INVALID, "fgs_smooth3d_fwd: null or aliased pointers");
  launch_conv<true>(in, X, Y, Z, c, k / 2, 1, out, fgs_s(stream));
  FGS_LAUNCH_OK("fgs_smooth3d_fwd");
  return 0;
}

// d_in = adjoint(d_out).  `scratch` holds the adjoint on the padded domain: (X+k-1)(Y+k-1)(Z+k-1) floats.
// d_out element (x,y,z) lives at d_out[((x*Y + y)*Z + z) * out_stride] (1 = dense; 4 = channel 0 of a [X,Y,Z,4] buffer).
FGS_API int fgs_smooth3d_bwd(const float *d_out, int64_t out_stride, int X, int Y, int Z, int k, const float *taps_host,
                             float *scratch, float *d_in, fgs_stream_t stream) {
  Conv3 c;
  if (int e = make_conv("fgs_smooth3d_bwd", X, Y, Z, k, taps_host, &c)) return e;
  FGS_REQUIRE(d_out && d_in && scratch && d_in != d_out && scratch != d_out && scratch != d_in && out_stride >= 1,
              FGS_E_INVALID, "fgs_smooth3d_bwd: null or aliased pointers, or out_stride < 1");
  // dP[q] = sum_t w[t] d_out[q - (t - r)] = sum_t' w[k-1-t'] d_out[q + t' - r]: flipped taps on the padded output domain
  Conv3 f = c;
  const int n3 = k * k * k;
  for (int i = 0; i < n3; ++i) f.w[i] = c.w[n3 - 1 - i];
  f.X = X + k - 1; f.Y = Y + k - 1; f.Z = Z + k - 1;
  launch_conv<false>(d_out, X, Y, Z, f, k - 1, out_stride, scratch, fgs_s(stream));
  FGS_LAUNCH_OK("fgs_smooth3d_bwd/conv");
  hipLaunchKernelGGL(k_fold_padded, dim3(fgs_blocks((int64_t)X * Y * Z)), dim3(FGS_BLOCK), 0, fgs_s(stream), scratch, X, Y, Z,
                     k / 2, d_in);
  FGS_LAUNCH_OK("fgs_smooth3d_bwd/fold");
  return 0;
}

FGS_API int fgs_sdf_gradvol_fwd(const float *sdf, int X, int Y, int Z, float voxel_size, int mode, float *grad3,
                                const float *pack_sdf, float *vol4, fgs_stream_t stream) {
  FGS_REQUIRE(X > 0 && Y > 0 && Z > 0 && (int64_t)X * Y * Z < ((int64_t)1 << 38), FGS_E_RANGE, "fgs_sdf_gradvol_fwd: size");
  FGS_REQUIRE(mode == 0 || mode == 1, FGS_E_INVALID, "fgs_sdf_gradvol_fwd: mode=%d (0 = interpolate, 1 = raw)", mode);
  FGS_REQUIRE(sdf && grad3 && (!vol4 || (pack_sdf && (reinterpret_cast<uintptr_t>(vol4) & 15) == 0)), FGS_E_INVALID,
              "fgs_sdf_gradvol_fwd: null pointer (vol4 needs pack_sdf and 16-byte alignment)");
  if (mode == 0)
    hipLaunchKernelGGL(k_gradvol_fwd<0>, dim3(fgs_blocks((int64_t)X * Y * Z)), dim3(FGS_BLOCK), 0, fgs_s(stream), sdf, X, Y, Z,
                       voxel_size, grad3, pack_sdf, reinterpret_cast<float4 *>(vol4));
  else
    hipLaunchKernelGGL(k_gradvol_fwd<1>, dim3(fgs_blocks((int64_t)X * Y * Z)), dim3(FGS_BLOCK), 0, fgs_s(stream), sdf, X, Y, Z,
                       voxel_size, grad3, pack_sdf, reinterpret_cast<float4 *>(vol4));
  FGS_LAUNCH_OK("fgs_sdf_gradvol_fwd");
  return 0;
}

FGS_API int fgs_sdf_gradvol_bwd(const float *d_grad3, int64_t chan_stride, int64_t voxel_stride, int X, int Y, int Z,
                                float voxel_size, int mode, float *d_sdf, int accumulate, const float *d_grad3_b,
                                fgs_stream_t stream) {
  FGS_REQUIRE(X > 0 && Y > 0 && Z > 0 && (int64_t)X * Y * Z < ((int64_t)1 << 38), FGS_E_RANGE, "fgs_sdf_gradvol_bwd: size");
  FGS_REQUIRE(mode == 0 || mode == 1, FGS_E_INVALID, "fgs_sdf_gradvol_bwd: mode=%d (0 = interpolate, 1 = raw)", mode);
  FGS_REQUIRE(d_grad3 && d_sdf && chan_stride >= 1 && voxel_stride >= 1, FGS_E_INVALID, "fgs_sdf_gradvol_bwd: bad arguments");
  if (mode == 0)
    hipLaunchKernelGGL(k_gradvol_bwd<0>, dim3(fgs_blocks((int64_t)X * Y * Z)), dim3(FGS_BLOCK), 0, fgs_s(stream), d_grad3,
                       chan_stride, voxel_stride, X, Y, Z, voxel_size, d_sdf, accumulate, d_grad3_b);
  else
    hipLaunchKernelGGL(k_gradvol_bwd<1>, dim3(fgs_blocks((int64_t)X * Y * Z)), dim3(FGS_BLOCK), 0, fgs_s(stream), d_grad3,
                       chan_stride, voxel_stride, X, Y, Z, voxel_size, d_sdf, accumulate, d_grad3_b);
  FGS_LAUNCH_OK("fgs_sdf_gradvol_bwd");
  return 0;
}

// Smooth-gradient TV term of nerf.density_total_variation (model/nerf.py:436-446) over a gradient volume g3 [3,X,Y,Z]:
//   loss_accum += weight * mean_masked((conv3(g).detach() - g)^2),   d_g3 = d loss / d g3   (fully written)
// taps_host: the 27 taps of tv_smooth_conv (replicate padding); mask [X,Y,Z] uint8 or NULL; inv_count_dev = 1 / (number
// of elements the mean runs over: 3 * mask.sum(), or 3 X Y Z), a DEVICE scalar so that no host read is needed.
FGS_API int64_t fgs_smooth_tv_scratch_floats(int X, int Y, int Z) {
  if (X <= 0 || Y <= 0 || Z <= 0) return 0;
  return 1 + 3 * (int64_t)((X + CT_X - 1) / CT_X) * ((Y + CT_Y - 1) / CT_Y) * ((Z + CT_Z - 1) / CT_Z);
}

FGS_API int fgs_smooth_tv_loss(const float *g3, int X, int Y, int Z, const float *taps_host, const uint8_t *mask,
                               const float *inv_count_dev, float weight, const float *add_in_dev, float *scratch,
                               int64_t scratch_floats, float *loss_out, float *d_g3, fgs_stream_t stream) {
  Conv3 c;
  if (int e = make_conv("fgs_smooth_tv_loss", X, Y, Z, 3, taps_host, &c)) return e;
  FGS_REQUIRE(g3 && inv_count_dev && loss_out && scratch && d_g3 && g3 != d_g3, FGS_E_INVALID,
              "fgs_smooth_tv_loss: null or aliased pointers");
  const int64_t tiles = (int64_t)((X + CT_X - 1) / CT_X) * ((Y + CT_Y - 1) / CT_Y) * ((Z + CT_Z - 1) / CT_Z);
  FGS_REQUIRE(scratch_floats >= 1 + 3 * tiles, FGS_E_INVALID, "fgs_smooth_tv_loss: scratch %lld floats, need %lld",
              (long long)scratch_floats, (long long)(1 + 3 * tiles));
  TvEpi tv{mask, inv_count_dev, weight, loss_out, add_in_dev, scratch + 1, reinterpret_cast<unsigned *>(scratch)};
  hipLaunchKernelGGL((k_conv3d_tiled<3, true, 1>), dim3((unsigned)tiles, 3), dim3(FGS_BLOCK), 0, fgs_s(stream), g3, X, Y, Z, c,
                     1, (int64_t)1, d_g3, tv);
  FGS_LAUNCH_OK("fgs_smooth_tv_loss");
  return 0;
}
