// ide.hip -- integrated directional encoding (Ref-NeRF eq. 6-8), the reference's generate_ide_fn (model/utils.py:515-574;
// constructed at model/nerf.py:179 with sh_max_level, never evaluated by its forward passes: BASELINE config 3 names it,
// SURVEY.md 8d lists it as an optional extra encoding).
//
//   ide[s, i]     = Re( (x + iy)^m_i ) * P_i(z) * exp(-sigma_i * kappa_inv[s])          i < n
//   ide[s, n + i] = Im( (x + iy)^m_i ) * P_i(z) * exp(-sigma_i * kappa_inv[s])
//   P_i(z) = sum_k mat[k, i] z^k (k <= l_max),  sigma_i = l_i (l_i + 1) / 2,  (m_i, l_i) = ml[:, i]
//
// One thread per sample; the coefficient matrix (<= 17 x 36 floats) and the (m, l) pairs sit in LDS.  Powers are repeated
// products (the torch reference evaluates complex pow through exp/log; parity is to 1e-5 against it and against scipy's
// Y_l^m).  The backward pass recomputes the terms and returns d/d xyz and d/d kappa_inv.
#include "fgs_common.h"

namespace {

constexpr int IDE_MAX_N = 36;     // deg_view 5: 2 + 3 + 5 + 9 + 17
constexpr int IDE_MAX_K = 17;     // l_max + 1 = 2^(deg_view - 1) + 1

struct IdeTables {
  float mat[IDE_MAX_K * IDE_MAX_N];
  int m[IDE_MAX_N], l[IDE_MAX_N];
};

__device__ __forceinline__ void load_tables(IdeTables &t, const float *mat, const int *ml, int n, int nk) {
  for (int q = threadIdx.x; q < nk * n; q += blockDim.x) t.mat[q] = mat[q];
  for (int q = threadIdx.x; q < n; q += blockDim.x) {
    t.m[q] = ml[q];
    t.l[q] = ml[n + q];
  }
  __syncthreads();
}

template <bool BWD>
__global__ __launch_bounds__(FGS_BLOCK) void k_ide(const float *__restrict__ xyz, const float *__restrict__ kappa_inv,
                                                   const float *__restrict__ mat, const int *__restrict__ ml, int n, int nk,
                                                   int64_t M, float *__restrict__ out, const float *__restrict__ g_out,
                                                   float *__restrict__ g_xyz, float *__restrict__ g_kappa) {
  __shared__ IdeTables t;
  load_tables(t, mat, ml, n, nk);
  const int64_t s = (int64_t)blockIdx.x * FGS_BLOCK + threadIdx.x;
  const bool live = s < M;
  const float x = live ? xyz[3 * s] : 0.f, y = live ? xyz[3 * s + 1] : 0.f, z = live ? xyz[3 * s + 2] : 0.f;
  const float kinv = live ? kappa_inv[s] : 0.f;
  float zp[IDE_MAX_K];                     // z^k
  zp[0] = 1.f;
#pragma unroll
  for (int k = 1; k < IDE_MAX_K; ++k) zp[k] = (k < nk) ? zp[k - 1] * z : 0.f;
  float gx = 0.f, gy = 0.f, gz = 0.f, gk = 0.f;
  float wr = 1.f, wi = 0.f, pr = 0.f, pi = 0.f;   // (x+iy)^m and (x+iy)^(m-1), advanced as m grows within a degree
  int m_cur = 0;
  for (int i = 0; i < n; ++i) {
    const int m = t.m[i], l = t.l[i];
    if (m < m_cur) { wr = 1.f; wi = 0.f; pr = 0.f; pi = 0.f; m_cur = 0; }    // next degree: m restarts at 0
    while (m_cur < m) {
      pr = wr; pi = wi;
      const float nr = wr * x - wi * y, ni = wr * y + wi * x;
      wr = nr; wi = ni;
      ++m_cur;
    }
    float P = 0.f, dP = 0.f;
#pragma unroll
    for (int k = 0; k < IDE_MAX_K; ++k) {          // fixed trip count: zp[] stays in registers
      if (k > l - m) continue;
      const float c = t.mat[k * n + i];
      P += c * zp[k];
      if (BWD && k > 0) dP += c * (float)k * zp[k > 0 ? k - 1 : 0];
    }
    const float a = expf(-(0.5f * (float)l * (float)(l + 1)) * kinv);
    const float o_re = wr * P * a, o_im = wi * P * a;
    if (!BWD) {
      if (live) {
        out[s * 2 * n + i] = o_re;
        out[s * 2 * n + n + i] = o_im;
      }
    } else if (live) {
      const float gr = g_out[s * 2 * n + i], gi = g_out[s * 2 * n + n + i];
      const float fm = (float)m;
      const float dwr = fm * pr, dwi = fm * pi;                 // d w / dx ; d w / dy = i * that
      gx += a * P * (gr * dwr + gi * dwi);
      gy += a * P * (gi * dwr - gr * dwi);
      gz += a * dP * (gr * wr + gi * wi);
      gk -= (0.5f * (float)l * (float)(l + 1)) * (gr * o_re + gi * o_im);
    }
  }
  if (BWD && live) {
    g_xyz[3 * s] = gx;
    g_xyz[3 * s + 1] = gy;
    g_xyz[3 * s + 2] = gz;
    g_kappa[s] = gk;
  }
}

bool ide_args_ok(int n, int nk) { return n > 0 && n <= IDE_MAX_N && nk > 0 && nk <= IDE_MAX_K; }

}  // namespace

FGS_API int fgs_ide_fwd(const float *xyz, const float *kappa_inv, const float *mat, const int *ml, int n, int n_pow, int64_t M,
                        float *out, fgs_stream_t stream) {
  FGS_REQUIRE(ide_args_ok(n, n_pow), FGS_E_INVALID, "fgs_ide_fwd: n = %d, n_pow = %d out of range", n, n_pow);
  FGS_REQUIRE(M >= 0 && M <= FGS_MAX_ELEMS, FGS_E_INVALID, "fgs_ide_fwd: bad M");
  if (M == 0) return 0;
  FGS_REQUIRE(xyz && kappa_inv && mat && ml && out, FGS_E_INVALID, "fgs_ide_fwd: null pointer");
  hipLaunchKernelGGL(k_ide<false>, dim3(fgs_blocks(M)), dim3(FGS_BLOCK), 0, fgs_s(stream), xyz, kappa_inv, mat, ml, n, n_pow,
                     M, out, nullptr, nullptr, nullptr);
  FGS_LAUNCH_OK("fgs_ide_fwd");
  return 0;
}

FGS_API int fgs_ide_bwd(const float *xyz, const float *kappa_inv, const float *mat, const int *ml, int n, int n_pow, int64_t M,
                        const float *g_out, float *g_xyz, float *g_kappa_inv, fgs_stream_t stream) {
  FGS_REQUIRE(ide_args_ok(n, n_pow), FGS_E_INVALID, "fgs_ide_bwd: n = %d, n_pow = %d out of range", n, n_pow);
  FGS_REQUIRE(M >= 0 && M <= FGS_MAX_ELEMS, FGS_E_INVALID, "fgs_ide_bwd: bad M");
  if (M == 0) return 0;
  FGS_REQUIRE(xyz && kappa_inv && mat && ml && g_out && g_xyz && g_kappa_inv, FGS_E_INVALID, "fgs_ide_bwd: null pointer");
  hipLaunchKernelGGL(k_ide<true>, dim3(fgs_blocks(M)), dim3(FGS_BLOCK), 0, fgs_s(stream), xyz, kappa_inv, mat, ml, n, n_pow, M,
                     nullptr, g_out, g_xyz, g_kappa_inv);
  FGS_LAUNCH_OK("fgs_ide_bwd");
  return 0;
}
