// fgs_taps.h -- index-space helpers shared by trilerp.hip (operator-at-a-time kernels) and the fused
// march / feature kernels: world point -> index coordinates, the clamped axis taps of
// nerf.sample_sdfs (model/nerf.py:597-637) and the SDF value + 6-tap gradient of nerf.grid_sampler
// (model/nerf.py:639-672).
#pragma once

#include "fgs_common.h"

struct PointIdx {
  float fx, fy, fz;
};

__device__ __forceinline__ PointIdx fgs_point_to_index(float px, float py, float pz, const float *lo, const float *hi,
                                                       const GridDesc &d) {
  PointIdx p;
  p.fx = fgs_world_to_index(px, lo[0], hi[0], (int)d.X);
  p.fy = fgs_world_to_index(py, lo[1], hi[1], (int)d.Y);
  p.fz = fgs_world_to_index(pz, lo[2], hi[2], (int)d.Z);
  return p;
}

// tap t of a displacement: pair = 0,1: -z,+z   2,3: -y,+y   4,5: -x,+x (the reference offsets act on the zyx-flipped
// index).  The tap point is ind -/+ disp on that axis, clamped to the volume in INDEX space, then pushed through the
// reference's index -> [-1,1] -> index round trip (model/nerf.py:618 + grid_sample's unnormalize) before the lookup.
struct TapPoint {
  float fx, fy, fz;  // index coordinates actually sampled
  float clamped;     // the clamped (pre round trip) coordinate on the displaced axis, for `diff`
};

__device__ __forceinline__ TapPoint fgs_tap_point(const PointIdx &p, const GridDesc &d, int pair, float disp) {
  const int axis_zyx = pair >> 1;               // 0 -> z, 1 -> y, 2 -> x
  const float off = (pair & 1) ? disp : -disp;  // offset row (-1 | +1) * displace
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

__device__ __forceinline__ float fgs_tap_value(const float *__restrict__ g, const GridDesc &d, const TapPoint &tp) {
  return fgs_tri_sample(g, d, 0, fgs_tri_setup(tp.fx, tp.fy, tp.fz));
}

// Host-known scene geometry passed by value to the fused kernels (no device reads of the bbox).
struct SceneGeom {
  float lo[3], hi[3];
  int X, Y, Z;
  float voxel_size;  // fp32 value of model.voxel_size
};

__device__ __forceinline__ GridDesc fgs_sdf_desc(const SceneGeom &s) {
  return GridDesc{1, s.X, s.Y, s.Z, (int64_t)s.X * s.Y * s.Z, (int64_t)s.Y * s.Z, (int64_t)s.Z, 1};
}

// sdf value + xyz-ordered finite-difference gradient at a world point: grid_sampler(sample_ret, sample_grad)
// with displace 1.0 (model/nerf.py:654-666): grad_axis = ((f+ - f-) / diff) / voxel_size.
struct SdfSample {
  float sdf, gx, gy, gz;
};

__device__ __forceinline__ SdfSample fgs_sdf_value_grad(const float *__restrict__ g, const SceneGeom &s, float px,
                                                        float py, float pz) {
  const GridDesc d = fgs_sdf_desc(s);
  const PointIdx p = fgs_point_to_index(px, py, pz, s.lo, s.hi, d);
  SdfSample o;
  o.sdf = fgs_tri_sample(g, d, 0, fgs_tri_setup(p.fx, p.fy, p.fz));
  float grad_zyx[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const TapPoint tm = fgs_tap_point(p, d, 2 * a, 1.0f), tp = fgs_tap_point(p, d, 2 * a + 1, 1.0f);
    const float fm = fgs_tap_value(g, d, tm), fp = fgs_tap_value(g, d, tp);
    grad_zyx[a] = ((fp - fm) / (tp.clamped - tm.clamped)) / s.voxel_size;
  }
  o.gx = grad_zyx[2];
  o.gy = grad_zyx[1];
  o.gz = grad_zyx[0];
  return o;
}

// ---- the same value + gradient from a register-held neighbourhood ----------------------------------------------------
// The seven trilinear footprints of fgs_sdf_value_grad (centre cell + the six cells one voxel away) read 56 corners, but
// only 32 distinct voxels: the 2x2 (x,y) columns of the centre cell over z0-1 .. z0+2, and the 8 columns one step out in
// x or y over z0 .. z0+1.  z is the contiguous axis, so these are 4 four-float and 8 two-float runs: 12 vector loads
// instead of 56 scalar gathers (the march kernel is bound by the gather rate of the texture path, not by HBM).
// Every tap still computes its OWN clamped point, round trip, floor and weights exactly as above; the register table
// only replaces the loads when the tap's cell is the expected neighbour of the centre cell, which it is except within
// an ulp of a cell boundary or next to the volume faces -- those taps (or the whole sample) take the gather path, so
// the result is bit-identical by construction.
struct __attribute__((packed, aligned(4))) FgsRun4 { float v[4]; };
struct __attribute__((packed, aligned(4))) FgsRun2 { float v[2]; };

template <int OX, int OY, int OZ>
__device__ __forceinline__ float fgs_tri_from_nb(const float (&N)[4][4][4], const TriCorners &t) {
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc = fmaf(N[1 + OX + (k >> 2)][1 + OY + ((k >> 1) & 1)][1 + OZ + (k & 1)], t.w[k], acc);
  return acc;
}

template <int PAIR>
__device__ __forceinline__ float fgs_tap_from_nb(const float *__restrict__ g, const GridDesc &d, const float (&N)[4][4][4],
                                                 const TriCorners &c, const TapPoint &tp) {
  constexpr int AX = PAIR >> 1, SG = (PAIR & 1) ? 1 : -1;       // axis 0 -> z, 1 -> y, 2 -> x
  constexpr int OX = (AX == 2) ? SG : 0, OY = (AX == 1) ? SG : 0, OZ = (AX == 0) ? SG : 0;
  const TriCorners t = fgs_tri_setup(tp.fx, tp.fy, tp.fz);
  if (t.x0 == c.x0 + OX && t.y0 == c.y0 + OY && t.z0 == c.z0 + OZ) return fgs_tri_from_nb<OX, OY, OZ>(N, t);
  return fgs_tri_sample(g, d, 0, t);
}

__device__ __forceinline__ SdfSample fgs_sdf_value_grad_nb(const float *__restrict__ g, const SceneGeom &s, float px,
                                                           float py, float pz) {
  const GridDesc d = fgs_sdf_desc(s);
  const PointIdx p = fgs_point_to_index(px, py, pz, s.lo, s.hi, d);
  const TriCorners c = fgs_tri_setup(p.fx, p.fy, p.fz);
  const bool interior = c.x0 >= 1 && c.x0 + 2 < s.X && c.y0 >= 1 && c.y0 + 2 < s.Y && c.z0 >= 1 && c.z0 + 2 < s.Z;
  if (!interior) return fgs_sdf_value_grad(g, s, px, py, pz);
  float N[4][4][4];
  const float *base = g + (int64_t)(c.x0 - 1) * d.sX + (int64_t)(c.y0 - 1) * d.sY + (c.z0 - 1);
#pragma unroll
  for (int dx = 0; dx < 4; ++dx)
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
      const bool cx = dx == 1 || dx == 2, cy = dy == 1 || dy == 2;
      const float *col = base + dx * d.sX + dy * d.sY;
      if (cx && cy) {
        const FgsRun4 q = *reinterpret_cast<const FgsRun4 *>(col);
        N[dx][dy][0] = q.v[0]; N[dx][dy][1] = q.v[1]; N[dx][dy][2] = q.v[2]; N[dx][dy][3] = q.v[3];
      } else if (cx || cy) {
        const FgsRun2 q = *reinterpret_cast<const FgsRun2 *>(col + 1);
        N[dx][dy][1] = q.v[0]; N[dx][dy][2] = q.v[1];
      }
    }
  SdfSample o;
  o.sdf = fgs_tri_from_nb<0, 0, 0>(N, c);
  const TapPoint zm = fgs_tap_point(p, d, 0, 1.0f), zp = fgs_tap_point(p, d, 1, 1.0f);
  const TapPoint ym = fgs_tap_point(p, d, 2, 1.0f), yp = fgs_tap_point(p, d, 3, 1.0f);
  const TapPoint xm = fgs_tap_point(p, d, 4, 1.0f), xp = fgs_tap_point(p, d, 5, 1.0f);
  const float fzm = fgs_tap_from_nb<0>(g, d, N, c, zm), fzp = fgs_tap_from_nb<1>(g, d, N, c, zp);
  const float fym = fgs_tap_from_nb<2>(g, d, N, c, ym), fyp = fgs_tap_from_nb<3>(g, d, N, c, yp);
  const float fxm = fgs_tap_from_nb<4>(g, d, N, c, xm), fxp = fgs_tap_from_nb<5>(g, d, N, c, xp);
  o.gz = ((fzp - fzm) / (zp.clamped - zm.clamped)) / s.voxel_size;
  o.gy = ((fyp - fym) / (yp.clamped - ym.clamped)) / s.voxel_size;
  o.gx = ((fxp - fxm) / (xp.clamped - xm.clamped)) / s.voxel_size;
  return o;
}
