// fgs_common.h -- shared host/device helpers for libfgs_hip.so (gfx950 only).
//
// The library is built with -ffp-contract=off: every fused multiply-add in a kernel is an
// explicit fmaf(), placed where the oracle (oracle/fgs_oracle.c) places it, so integer outputs
// (sample counts, ray/step ids, masks, segment ends) are bit-exact against the oracle.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/fgs_hip.h"

#define FGS_API extern "C" __attribute__((visibility("default")))

int fgs_set_error(int code, const char *fmt, ...);

#define FGS_REQUIRE(cond, code, ...)                 \
  do {                                               \
    if (!(cond)) return fgs_set_error((code), __VA_ARGS__); \
  } while (0)

// After a kernel launch: surface launch-configuration errors without synchronising.
#define FGS_LAUNCH_OK(what)                                                              \
  do {                                                                                   \
    hipError_t e__ = hipGetLastError();                                                  \
    if (e__ != hipSuccess) return fgs_set_error((int)e__, "%s: %s", (what), hipGetErrorString(e__)); \
  } while (0)

static inline hipStream_t fgs_s(fgs_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Device-resident values of a sync-free / captured step, passed EXPLICITLY to the entry points that can read them (fgs_dyn_t,
// include/fgs_hip.h): the survivor row count, NeuS 1/s, the compact-dX0 operand form.  NULL = host arguments only.
static inline const int64_t *fgs_dyn_rows(const fgs_dyn_t *d) { return d ? d->row_count : nullptr; }
static inline const float *fgs_dyn_inv_s(const fgs_dyn_t *d) { return d ? d->inv_s : nullptr; }
static inline int fgs_dyn_compact(const fgs_dyn_t *d) { return d ? d->dx0_compact : 0; }

// In-kernel wall-clock stamps of a launch (fgs_dyn_t.stamps: measurement only).
struct FgsStamps {
  unsigned long long *p;
  const int64_t *step;
  int64_t slots, stride;
};
static inline FgsStamps fgs_dyn_stamps(const fgs_dyn_t *d, unsigned long long *fallback = nullptr) {
  FgsStamps s;
  s.p = (d && d->stamps) ? d->stamps : fallback;
  s.step = (d && d->stamps) ? d->stamp_step : nullptr;
  s.slots = (d && d->stamps && d->stamp_slots > 0) ? d->stamp_slots : 1;
  s.stride = (d && d->stamps) ? d->stamp_stride : 0;
  return s;
}
#ifdef __HIPCC__
__device__ __forceinline__ unsigned long long *fgs_stamp_base(const FgsStamps &s) {
  if (!s.p) return nullptr;
  return s.step ? s.p + (*s.step % s.slots) * s.stride : s.p;
}
// The 8-word record of THIS workgroup inside a launch's region of FGS_STAMP_WGS * 8 words (the size include/fgs_hip.h states);
// workgroups beyond it do not stamp (a grid larger than the CU count would otherwise write into the next launch's region).
constexpr unsigned FGS_STAMP_WGS = 256;
__device__ __forceinline__ unsigned long long *fgs_stamp_wg(const FgsStamps &s) {
  unsigned long long *b = fgs_stamp_base(s);
  return (b && blockIdx.x < FGS_STAMP_WGS) ? b + 8 * blockIdx.x : nullptr;
}
#endif

constexpr int FGS_WAVE = 64;      // gfx950 wavefront
constexpr int FGS_BLOCK = 256;    // 4 waves: one per SIMD of a CU
constexpr int64_t FGS_MAX_ELEMS = (int64_t)1 << 40;

static inline unsigned fgs_blocks(int64_t n, int per_block = FGS_BLOCK) {
  return (unsigned)((n + per_block - 1) / per_block);
}

// Tuning knob read once per call site from the environment (experiments with the co-scheduling of the backward pass's two
// graph branches: DESIGN.md section 3).  Not part of the ABI: a missing variable means the default.
#include <stdlib.h>
static inline int fgs_env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

// ------------------------------------------------------------------------------------ device

// "The last workgroup sums the partials": called by ONE thread of a workgroup after it has written the workgroup's partial
// result(s) with agent-scope atomic stores (written through the XCD's L2); returns true in the workgroup that arrives last,
// which then reads all partials with agent-scope atomic loads.  There is deliberately no agent-scope release fence: on this
// part it writes back EVERY dirty line of the L2 (whatever the neighbouring kernels left there), once per workgroup -- a
// 1024-workgroup loss kernel spent 20 of its 42 us in them.  The stores are waited for (vmcnt) before the arrival is counted.
__device__ __forceinline__ bool fgs_arrive_is_last(unsigned *counter, unsigned n_workgroups) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  return __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_workgroups - 1;
}

// ... and the reading side: sum of p[first], p[first + stride], ... (index < n) in index order, the agent-scope loads issued eight at
// a time before the first addition (one at a time, each a ~1 us round trip to memory, they were 30 of a 37 us launch).
template <typename T, typename ACC = T>
__device__ __forceinline__ ACC fgs_partials_sum(const T *p, unsigned first, unsigned n, unsigned stride) {
  ACC acc = (ACC)0;
  for (unsigned b = first; b < n; b += 8 * stride) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned i = b + u * stride;
      v[u] = i < n ? __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (T)0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += (ACC)v[u];
  }
  return acc;
}

// s_setprio with a run-time (wave-uniform) level: the instruction takes an immediate.
__device__ __forceinline__ void fgs_setprio(int p) {
  if (p == 1) __builtin_amdgcn_s_setprio(1);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else if (p >= 3) __builtin_amdgcn_s_setprio(3);
}

// Rows a kernel really has to process: the host count, or -- under fgs_dyn_t.row_count -- the device count clamped to
// the capacity the host count then stands for.  A wave-uniform scalar load.
__device__ __forceinline__ int64_t fgs_rows(int64_t host_or_capacity, const int64_t *__restrict__ count_dev) {
  if (!count_dev) return host_or_capacity;
  const int64_t m = *count_dev;
  return m < host_or_capacity ? (m < 0 ? 0 : m) : host_or_capacity;
}

struct GridDesc {
  int64_t C, X, Y, Z;      // logical [1,C,X,Y,Z]
  int64_t sC, sX, sY, sZ;  // element strides
};

// Values the program knows to be identical in all 64 lanes, moved to SGPRs so that the loops and
// branches controlled by them are scalar (s_cbranch) instead of exec-masked.
__device__ __forceinline__ int fgs_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t fgs_uniform(int64_t v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
  const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((int64_t)hi << 32) | (int64_t)lo;
}
__device__ __forceinline__ float fgs_bcast_lane(float v, int src_lane /*uniform*/) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

__device__ __forceinline__ float fgs_rnorm3(float dx, float dy, float dz) {
  // render_utils_kernel.cu:48-51 with the contraction pinned (see oracle/fgs_oracle.c orc_rnorm)
  return sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
}

// World coordinate -> continuous voxel index along one axis, reproducing the reference chain
//   u = (p - lo) / (hi - lo)            model/grid.py:55
//   c = u * 2 - 1                       model/grid.py:55
//   f = ((c + 1) / 2) * (size - 1)      ATen grid_sampler_unnormalize, align_corners=True
__device__ __forceinline__ float fgs_world_to_index(float p, float lo, float hi, int size) {
  const float u = (p - lo) / (hi - lo);
  const float c = u * 2.f - 1.f;
  return ((c + 1.f) / 2.f) * (float)(size - 1);
}

// Index coordinate -> grid_sample coordinate and back, as model/nerf.py:618 + unnormalize do
//   c = (i / (size-1)) * 2 - 1 ;  f = ((c + 1) / 2) * (size - 1)
__device__ __forceinline__ float fgs_index_roundtrip(float i, int size) {
  const float c = (i / (float)(size - 1)) * 2.f - 1.f;
  return ((c + 1.f) / 2.f) * (float)(size - 1);
}

struct TriCorners {
  int x0, y0, z0;
  float w[8];   // order: (x0,y0,z0) (x0,y0,z1) (x0,y1,z0) (x0,y1,z1) (x1,y0,z0) (x1,y0,z1) (x1,y1,z0) (x1,y1,z1)
                // = ATen's tnw,tne,tsw,tse,bnw,bne,bsw,bse with ATen (ix,iy,iz) = our (z,y,x)
};

__device__ __forceinline__ float fgs_safe_floor(float f) {
  // keeps the int conversion defined for wild inputs; any |f| this large is out of the volume anyway
  f = fminf(fmaxf(f, -4.f), 1073741824.f);
  return floorf(f);
}

__device__ __forceinline__ TriCorners fgs_tri_setup(float fx, float fy, float fz) {
  TriCorners t;
  const float flx = fgs_safe_floor(fx), fly = fgs_safe_floor(fy), flz = fgs_safe_floor(fz);
  t.x0 = (int)flx; t.y0 = (int)fly; t.z0 = (int)flz;
  const float ax1 = (flx + 1.f) - fx, ax0 = fx - flx;   // weight of x0 / x1
  const float ay1 = (fly + 1.f) - fy, ay0 = fy - fly;
  const float az1 = (flz + 1.f) - fz, az0 = fz - flz;
  // ATen multiplies (x-term * y-term) * z-term in ITS axis naming = (our z * our y) * our x
  t.w[0] = (az1 * ay1) * ax1;
  t.w[1] = (az0 * ay1) * ax1;
  t.w[2] = (az1 * ay0) * ax1;
  t.w[3] = (az0 * ay0) * ax1;
  t.w[4] = (az1 * ay1) * ax0;
  t.w[5] = (az0 * ay1) * ax0;
  t.w[6] = (az1 * ay0) * ax0;
  t.w[7] = (az0 * ay0) * ax0;
  return t;
}

__device__ __forceinline__ bool fgs_in(int i, int n) { return (unsigned)i < (unsigned)n; }

// Trilinear value of channel `c` with zeros padding (corners outside the volume contribute 0).
__device__ __forceinline__ float fgs_tri_sample(const float *__restrict__ g, const GridDesc &d, int64_t c,
                                                const TriCorners &t) {
  // All eight loads are issued unconditionally on clamped (always valid) addresses and an out-of-volume corner gets a
  // zero weight: fmaf(v, 0, acc) == acc for the finite grid values, so the result equals skipping the corner, but
  // the loads carry no branch and the compiler can keep all of them (and those of neighbouring lookups) in flight
  // behind ONE s_waitcnt instead of one wait per load.
  const float *base = g + c * d.sC;
  float v[8], w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int x = t.x0 + (k >> 2), y = t.y0 + ((k >> 1) & 1), z = t.z0 + (k & 1);
    const bool ok = fgs_in(x, (int)d.X) && fgs_in(y, (int)d.Y) && fgs_in(z, (int)d.Z);
    const int xc = min(max(x, 0), (int)d.X - 1), yc = min(max(y, 0), (int)d.Y - 1), zc = min(max(z, 0), (int)d.Z - 1);
    v[k] = base[xc * d.sX + yc * d.sY + zc * d.sZ];
    w[k] = ok ? t.w[k] : 0.f;
  }
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc = fmaf(v[k], w[k], acc);
  return acc;
}

__device__ __forceinline__ void fgs_tri_scatter(float *__restrict__ g, const GridDesc &d, int64_t c,
                                                const TriCorners &t, float go) {
  float *base = g + c * d.sC;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int x = t.x0 + (k >> 2), y = t.y0 + ((k >> 1) & 1), z = t.z0 + (k & 1);
    if (fgs_in(x, (int)d.X) && fgs_in(y, (int)d.Y) && fgs_in(z, (int)d.Z))
      atomicAdd(base + x * d.sX + y * d.sY + z * d.sZ, t.w[k] * go);
  }
}
