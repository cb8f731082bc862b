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
