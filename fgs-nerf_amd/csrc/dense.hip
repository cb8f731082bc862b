// dense.hip -- the per-iteration full-volume operators of the coarse stages:
//   * 3-D smoothing of the SDF grid: nn.Conv3d(1, 1, k, padding=k//2, padding_mode='replicate') with frozen Gaussian
//     taps (model/nerf.py:260-278, applied every forward at :791/:969), forward and backward;
//   * the central-difference gradient volume neus_sdf_gradient(mode='interpolate') (model/nerf.py:485-494),
//     forward and backward.
// Both are pure streaming stencils over [X,Y,Z] fp32 (16 MB at 160^3): one thread per voxel, neighbours served by
// L1/L2; the backward passes are written as gathers (no atomics).
#include "fgs_common.h"

namespace {

constexpr int MAXK = 7;  // kernel side (the reference uses 3 and 5)

struct Conv3 {
  int X, Y, Z, k;
  float w[MAXK * MAXK * MAXK];  // taps [dx][dy][dz], cross-correlation order as torch stores them
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// out[x,y,z] = sum_t w[t] * in[clamp(x+tx), clamp(y+ty), clamp(z+tz)]     (replicate padding)
__global__ __launch_bounds__(FGS_BLOCK) void k_smooth3d_fwd(const float *__restrict__ in, Conv3 c,
                                                            float *__restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t N = (int64_t)c.X * c.Y * c.Z;
  if (idx >= N) return;
  const int z = (int)(idx % c.Z), y = (int)((idx / c.Z) % c.Y), x = (int)(idx / ((int64_t)c.Z * c.Y));
  const int r = c.k / 2;
  float acc = 0.f;
  for (int dx = 0; dx < c.k; ++dx) {
    const int xx = clampi(x + dx - r, 0, c.X - 1);
    for (int dy = 0; dy < c.k; ++dy) {
      const int yy = clampi(y + dy - r, 0, c.Y - 1);
      const float *row = in + ((int64_t)xx * c.Y + yy) * c.Z;
      const float *wr = c.w + (dx * c.k + dy) * c.k;
      for (int dz = 0; dz < c.k; ++dz) acc = fmaf(wr[dz], row[clampi(z + dz - r, 0, c.Z - 1)], acc);
    }
  }
  out[idx] = acc;
}

// Along one axis: the output coordinates o whose tap t lands (after clamping) on input coordinate v.
//   interior v: o = v - t ; v == 0: o in [0, -t] ; v == n-1: o in [n-1-t, n-1]   (intersected with [0, n))
__device__ __forceinline__ void src_range(int v, int t, int n, int &lo, int &hi) {
  lo = hi = v - t;
  if (v == 0) lo = 0;          // o + t <= 0
  if (v == n - 1) hi = n - 1;  // o + t >= n-1
  if (lo < 0) lo = 0;
  if (hi > n - 1) hi = n - 1;
}

// d_in[v] = sum over (o, t) with clamp(o + t) == v of w[t] * d_out[o]
__global__ __launch_bounds__(FGS_BLOCK) void k_smooth3d_bwd(const float *__restrict__ d_out, Conv3 c,
                                                            float *__restrict__ d_in) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t N = (int64_t)c.X * c.Y * c.Z;
  if (idx >= N) return;
  const int z = (int)(idx % c.Z), y = (int)((idx / c.Z) % c.Y), x = (int)(idx / ((int64_t)c.Z * c.Y));
  const int r = c.k / 2;
  float acc = 0.f;
  for (int dx = 0; dx < c.k; ++dx) {
    int x0, x1;
    src_range(x, dx - r, c.X, x0, x1);
    for (int dy = 0; dy < c.k; ++dy) {
      int y0, y1;
      src_range(y, dy - r, c.Y, y0, y1);
      for (int dz = 0; dz < c.k; ++dz) {
        int z0, z1;
        src_range(z, dz - r, c.Z, z0, z1);
        const float w = c.w[(dx * c.k + dy) * c.k + dz];
        for (int ox = x0; ox <= x1; ++ox)
          for (int oy = y0; oy <= y1; ++oy) {
            const float *row = d_out + ((int64_t)ox * c.Y + oy) * c.Z;
            for (int oz = z0; oz <= z1; ++oz) acc = fmaf(w, row[oz], acc);
          }
      }
    }
  }
  d_in[idx] = acc;
}

// g[0] = (s[x+1] - s[x-1]) / 2 / vs on 1 <= x <= X-2 (zero on the two faces), likewise g[1] along y, g[2] along z
__global__ __launch_bounds__(FGS_BLOCK) void k_gradvol_fwd(const float *__restrict__ s, int X, int Y, int Z, float vs,
                                                           float *__restrict__ g) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t N = (int64_t)X * Y * Z;
  if (idx >= N) return;
  const int z = (int)(idx % Z), y = (int)((idx / Z) % Y), x = (int)(idx / ((int64_t)Z * Y));
  const int64_t sx = (int64_t)Y * Z, sy = Z;
  g[idx] = (x >= 1 && x <= X - 2) ? (s[idx + sx] - s[idx - sx]) / 2.f / vs : 0.f;
  g[N + idx] = (y >= 1 && y <= Y - 2) ? (s[idx + sy] - s[idx - sy]) / 2.f / vs : 0.f;
  g[2 * N + idx] = (z >= 1 && z <= Z - 2) ? (s[idx + 1] - s[idx - 1]) / 2.f / vs : 0.f;
}

// d_s[v] (+)= sum_axis ( dg_axis[v-1] * [v-1 interior] - dg_axis[v+1] * [v+1 interior] ) / 2 / vs
__global__ __launch_bounds__(FGS_BLOCK) void k_gradvol_bwd(const float *__restrict__ dg, int X, int Y, int Z, float vs,
                                                           float *__restrict__ d_s, int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t N = (int64_t)X * Y * Z;
  if (idx >= N) return;
  const int z = (int)(idx % Z), y = (int)((idx / Z) % Y), x = (int)(idx / ((int64_t)Z * Y));
  const int64_t sx = (int64_t)Y * Z, sy = Z;
  float acc = 0.f;
  // s[v] appears as "+" in g at v-1 (needs 1 <= v-1 <= n-2) and as "-" in g at v+1 (needs 1 <= v+1 <= n-2)
  if (x - 1 >= 1 && x - 1 <= X - 2) acc += dg[idx - sx] / 2.f / vs;
  if (x + 1 >= 1 && x + 1 <= X - 2) acc -= dg[idx + sx] / 2.f / vs;
  if (y - 1 >= 1 && y - 1 <= Y - 2) acc += dg[N + idx - sy] / 2.f / vs;
  if (y + 1 >= 1 && y + 1 <= Y - 2) acc -= dg[N + idx + sy] / 2.f / vs;
  if (z - 1 >= 1 && z - 1 <= Z - 2) acc += dg[2 * N + idx - 1] / 2.f / vs;
  if (z + 1 >= 1 && z + 1 <= Z - 2) acc -= dg[2 * N + idx + 1] / 2.f / vs;
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
  hipLaunchKernelGGL(k_smooth3d_fwd, dim3(fgs_blocks((int64_t)X * Y * Z)), dim3(FGS_BLOCK), 0, fgs_s(stream), in, c, out);
  FGS_LAUNCH_OK("fgs_smooth3d_fwd");
  return 0;
}

FGS_API int fgs_smooth3d_bwd(const float *d_out, int X, int Y, int Z, int k, const float *taps_host, float *d_in,
                             fgs_stream_t stream) {
  Conv3 c;
  if (int e = make_conv("fgs_smooth3d_bwd", X, Y, Z, k, taps_host, &c)) return e;
  FGS_REQUIRE(d_out && d_in && d_in != d_out, FGS_E_INVALID, "fgs_smooth3d_bwd: null or aliased pointers");
  hipLaunchKernelGGL(k_smooth3d_bwd, dim3(fgs_blocks((int64_t)X * Y * Z)), dim3(FGS_BLOCK), 0, fgs_s(stream), d_out, c, d_in);
  FGS_LAUNCH_OK("fgs_smooth3d_bwd");
  return 0;
}

FGS_API int fgs_sdf_gradvol_fwd(const float *sdf, int X, int Y, int Z, float voxel_size, float *grad3, fgs_stream_t stream) {
  FGS_REQUIRE(X > 0 && Y > 0 && Z > 0 && (int64_t)X * Y * Z < ((int64_t)1 << 38), FGS_E_RANGE, "fgs_sdf_gradvol_fwd: size");
  FGS_REQUIRE(sdf && grad3, FGS_E_INVALID, "fgs_sdf_gradvol_fwd: null pointer");
  hipLaunchKernelGGL(k_gradvol_fwd, dim3(fgs_blocks((int64_t)X * Y * Z)), dim3(FGS_BLOCK), 0, fgs_s(stream), sdf, X, Y, Z,
                     voxel_size, grad3);
  FGS_LAUNCH_OK("fgs_sdf_gradvol_fwd");
  return 0;
}

FGS_API int fgs_sdf_gradvol_bwd(const float *d_grad3, int X, int Y, int Z, float voxel_size, float *d_sdf, int accumulate,
                                fgs_stream_t stream) {
  FGS_REQUIRE(X > 0 && Y > 0 && Z > 0 && (int64_t)X * Y * Z < ((int64_t)1 << 38), FGS_E_RANGE, "fgs_sdf_gradvol_bwd: size");
  FGS_REQUIRE(d_grad3 && d_sdf, FGS_E_INVALID, "fgs_sdf_gradvol_bwd: null pointer");
  hipLaunchKernelGGL(k_gradvol_bwd, dim3(fgs_blocks((int64_t)X * Y * Z)), dim3(FGS_BLOCK), 0, fgs_s(stream), d_grad3, X, Y, Z,
                     voxel_size, d_sdf, accumulate);
  FGS_LAUNCH_OK("fgs_sdf_gradvol_bwd");
  return 0;
}
