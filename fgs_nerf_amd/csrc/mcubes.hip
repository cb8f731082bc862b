// mcubes.hip -- marching cubes on a device-resident scalar field (BASELINE config 2 "marching-cubes
// mesh extraction").  The reference calls PyMCubes on the host (model/extract_geometry.py:24: vertices in index
// coordinates as float64, shared between triangles); this is the same contract on the device, three streaming passes:
//
//   k_mc_count     : per lattice point, which of its three +x/+y/+z edges cross the iso level (one mesh vertex each, owned
//                    by the point) and how many triangles its cell emits; per-workgroup totals for the host-side scan
//   k_mc_vertices  : vertex ids = rank in (lattice point, axis) order; position = point + t * axis with
//                    t = (iso - f1) / (f2 - f1) in double (PyMCubes' mc_isovalue_interpolation); writes vbase[point]
//   k_mc_triangles : cell case -> triangle table (fgs-nerf_amd/mc_tables.py) -> vertex ids of the owning points
//
// A corner is flagged when field < iso (NaN: not flagged).  Lattice points are numbered z fastest ([X][Y][Z] storage); a
// workgroup handles 256 consecutive points, so field reads and all outputs are coalesced, and ids are deterministic.
#include "fgs_common.h"

namespace {

struct McGrid {
  int X, Y, Z;
  int64_t total;
};

// exclusive prefix of v over the 256 threads of the workgroup (+ the workgroup total in *sum)
__device__ __forceinline__ int block_exclusive(int v, int *wave_tot /*[4] LDS*/, int *sum) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int t = wave_tot[w];
    if (w < wave) base += t;
    tot += t;
  }
  __syncthreads();
  *sum = tot;
  return base + inc - v;
}

__device__ __forceinline__ int cell_case(const float *__restrict__ f, const McGrid &g, int64_t v, float iso) {
  const int64_t sx = (int64_t)g.Y * g.Z, sy = g.Z;
  int c = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) c |= (f[v + (k & 1) * sx + ((k >> 1) & 1) * sy + (k >> 2)] < iso) ? (1 << k) : 0;
  return c;
}

__global__ __launch_bounds__(256) void k_mc_count(const float *__restrict__ f, McGrid g, float iso,
                                                  const uint8_t *__restrict__ ntri, uint8_t *__restrict__ vflags,
                                                  int64_t *__restrict__ block_v, int64_t *__restrict__ block_t) {
  __shared__ int wave_tot[4];
  __shared__ uint8_t s_ntri[256];
  s_ntri[threadIdx.x] = ntri[threadIdx.x];
  __syncthreads();
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int nv = 0, nt = 0;
  if (v < g.total) {
    const int k = (int)(v % g.Z), j = (int)((v / g.Z) % g.Y), i = (int)(v / ((int64_t)g.Z * g.Y));
    const bool b0 = f[v] < iso;
    int fl = 0;
    if (i + 1 < g.X && ((f[v + (int64_t)g.Y * g.Z] < iso) != b0)) fl |= 1;
    if (j + 1 < g.Y && ((f[v + g.Z] < iso) != b0)) fl |= 2;
    if (k + 1 < g.Z && ((f[v + 1] < iso) != b0)) fl |= 4;
    vflags[v] = (uint8_t)fl;
    nv = __popc(fl);
    if (i + 1 < g.X && j + 1 < g.Y && k + 1 < g.Z) nt = s_ntri[cell_case(f, g, v, iso)];
  }
  int sum_v, sum_t;
  block_exclusive(nv, wave_tot, &sum_v);
  block_exclusive(nt, wave_tot, &sum_t);
  if (threadIdx.x == 0) {
    block_v[blockIdx.x] = sum_v;
    block_t[blockIdx.x] = sum_t;
  }
}

__global__ __launch_bounds__(256) void k_mc_vertices(const float *__restrict__ f, McGrid g, float iso,
                                                     const uint8_t *__restrict__ vflags,
                                                     const int64_t *__restrict__ block_voff, uint32_t *__restrict__ vbase,
                                                     double *__restrict__ vertices) {
  __shared__ int wave_tot[4];
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int fl = (v < g.total) ? vflags[v] : 0;
  int sum;
  const int64_t base = block_voff[blockIdx.x] + block_exclusive(__popc(fl), wave_tot, &sum);
  if (v >= g.total) return;
  vbase[v] = (uint32_t)base;
  if (!fl) return;
  const int k = (int)(v % g.Z), j = (int)((v / g.Z) % g.Y), i = (int)(v / ((int64_t)g.Z * g.Y));
  const double f1 = (double)f[v], level = (double)iso;
  const int64_t stride[3] = {(int64_t)g.Y * g.Z, (int64_t)g.Z, 1};
  int r = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (!(fl & (1 << a))) continue;
    const double f2 = (double)f[v + stride[a]];
    double p[3] = {(double)i, (double)j, (double)k};
    // mc_isovalue_interpolation(iso, f1, f2, x1, x2) with x2 - x1 == 1
    p[a] = (f2 == f1) ? (p[a] + (p[a] + 1.0)) / 2.0 : (level - f1) / (f2 - f1) + p[a];
    double *out = vertices + (base + r) * 3;
    out[0] = p[0];
    out[1] = p[1];
    out[2] = p[2];
    ++r;
  }
}

__global__ __launch_bounds__(256) void k_mc_triangles(const float *__restrict__ f, McGrid g, float iso,
                                                      const int8_t *__restrict__ tri, const uint8_t *__restrict__ ntri,
                                                      const uint8_t *__restrict__ vflags, const uint32_t *__restrict__ vbase,
                                                      const int64_t *__restrict__ block_toff,
                                                      int64_t *__restrict__ triangles) {
  __shared__ int wave_tot[4];
  __shared__ int8_t s_tri[256 * 16];
  __shared__ uint8_t s_ntri[256];
  for (int q = threadIdx.x; q < 256 * 16; q += 256) s_tri[q] = tri[q];
  s_ntri[threadIdx.x] = ntri[threadIdx.x];
  __syncthreads();
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int nt = 0, c = 0;
  if (v < g.total) {
    const int k = (int)(v % g.Z), j = (int)((v / g.Z) % g.Y), i = (int)(v / ((int64_t)g.Z * g.Y));
    if (i + 1 < g.X && j + 1 < g.Y && k + 1 < g.Z) {
      c = cell_case(f, g, v, iso);
      nt = s_ntri[c];
    }
  }
  int sum;
  const int64_t base = block_toff[blockIdx.x] + block_exclusive(nt, wave_tot, &sum);
  if (!nt) return;
  const int64_t sx = (int64_t)g.Y * g.Z, sy = g.Z;
  for (int t = 0; t < nt; ++t) {
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int e = s_tri[c * 16 + 3 * t + m];
      const int a = e >> 2, o1 = e & 1, o2 = (e >> 1) & 1;
      // the edge's lower end: offsets of the two axes other than a, in increasing axis order
      const int dx = (a == 0) ? 0 : o1, dy = (a == 0) ? o1 : (a == 1 ? 0 : o2), dz = (a == 2) ? 0 : o2;
      const int64_t owner = v + dx * sx + dy * sy + dz;
      const int64_t vid = (int64_t)vbase[owner] + __popc(vflags[owner] & ((1 << a) - 1));
      triangles[(base + t) * 3 + m] = vid;
    }
  }
}

bool mc_dims_ok(int X, int Y, int Z) { return X >= 2 && Y >= 2 && Z >= 2 && (int64_t)X * Y * Z <= ((int64_t)1 << 32); }

}  // namespace

FGS_API int64_t fgs_mc_num_blocks(int X, int Y, int Z) { return ((int64_t)X * Y * Z + 255) / 256; }

FGS_API int fgs_mc_count(const float *field, int X, int Y, int Z, float iso, const uint8_t *ntri_table, uint8_t *vflags,
                         int64_t *block_vertices, int64_t *block_triangles, fgs_stream_t stream) {
  FGS_REQUIRE(mc_dims_ok(X, Y, Z), FGS_E_INVALID, "fgs_mc_count: field dims %d x %d x %d unsupported", X, Y, Z);
  FGS_REQUIRE(field && ntri_table && vflags && block_vertices && block_triangles, FGS_E_INVALID, "fgs_mc_count: null pointer");
  const McGrid g{X, Y, Z, (int64_t)X * Y * Z};
  hipLaunchKernelGGL(k_mc_count, dim3((unsigned)fgs_mc_num_blocks(X, Y, Z)), dim3(256), 0, fgs_s(stream), field, g, iso,
                     ntri_table, vflags, block_vertices, block_triangles);
  FGS_LAUNCH_OK("fgs_mc_count");
  return 0;
}

FGS_API int fgs_mc_emit(const float *field, int X, int Y, int Z, float iso, const int8_t *tri_table,
                        const uint8_t *ntri_table, const uint8_t *vflags, const int64_t *block_vertex_offset,
                        const int64_t *block_triangle_offset, uint32_t *vbase, int64_t n_vertices, int64_t n_triangles,
                        double *vertices, int64_t *triangles, fgs_stream_t stream) {
  FGS_REQUIRE(mc_dims_ok(X, Y, Z), FGS_E_INVALID, "fgs_mc_emit: field dims %d x %d x %d unsupported", X, Y, Z);
  FGS_REQUIRE(field && tri_table && ntri_table && vflags && block_vertex_offset && block_triangle_offset && vbase,
              FGS_E_INVALID, "fgs_mc_emit: null pointer");
  FGS_REQUIRE(n_vertices >= 0 && n_triangles >= 0 && n_vertices < ((int64_t)1 << 32) && (n_vertices == 0 || vertices) &&
                  (n_triangles == 0 || triangles),
              FGS_E_INVALID, "fgs_mc_emit: bad output sizes");
  const McGrid g{X, Y, Z, (int64_t)X * Y * Z};
  const dim3 grid((unsigned)fgs_mc_num_blocks(X, Y, Z));
  hipLaunchKernelGGL(k_mc_vertices, grid, dim3(256), 0, fgs_s(stream), field, g, iso, vflags, block_vertex_offset, vbase,
                     vertices);
  FGS_LAUNCH_OK("fgs_mc_emit (vertices)");
  if (n_triangles > 0) {
    hipLaunchKernelGGL(k_mc_triangles, grid, dim3(256), 0, fgs_s(stream), field, g, iso, tri_table, ntri_table, vflags, vbase,
                       block_triangle_offset, triangles);
    FGS_LAUNCH_OK("fgs_mc_emit (triangles)");
  }
  return 0;
}
