/*
 * ssd_synth.h — synthetic "L515-shaped" frame source, one function evaluated per pixel,
 * compiled for both host and device from this single definition so that the two produce
 * bit-identical float32 clouds (only + - * / floor and integer hashing; no libm calls;
 * the library is built with -ffp-contract=off).
 *
 * It stands in for Camera::waitForFrames() + rs2::pointcloud::calculate()
 * (reference camera.cpp:46-49, pointcloud.cpp:138), which need the physical camera.
 * Scene coordinates: x right, y forward, z up, origin on the ground below the camera.
 * Camera coordinates (RealSense): x right, y down, z forward; output = camera xyz, invalid = (0,0,0).
 */
#ifndef SSD_SYNTH_H_
#define SSD_SYNTH_H_

#include "../../include/ssd_source.h"
#include <hip/hip_runtime.h>

namespace ssd
{

__host__ __device__ inline uint64_t synth_mix(uint64_t x)
{
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

/* nearest positive hit of the ray o + t*d with an axis-aligned rectangle lying in the plane
 * coordinate[axis] = value; (lo0,hi0),(lo1,hi1) bound the two other coordinates (in axis order). */
__host__ __device__ inline void synth_hit(const double o[3], const double d[3], int axis, double value,
                                          double lo0, double hi0, double lo1, double hi1, double &best)
{
  if(d[axis] == 0.0)
    return;
  const double t = (value - o[axis]) / d[axis];
  if(!(t > 0.0) || !(t < best))
    return;
  const int a0 = axis == 0 ? 1 : 0;
  const int a1 = axis == 2 ? 1 : 2;
  const double p0 = o[a0] + t * d[a0];
  const double p1 = o[a1] + t * d[a1];
  if(p0 >= lo0 && p0 <= hi0 && p1 >= lo1 && p1 <= hi1)
    best = t;
}

__host__ __device__ inline void synth_pixel(const ssd_scene &s, uint64_t frame_key, int u, int v, float out[3])
{
  out[0] = 0.0f; out[1] = 0.0f; out[2] = 0.0f;

  const double dx = (u - s.cx) / s.fx;
  const double dy = (v - s.cy) / s.fy;

  /* ray in scene coordinates; the optical-axis coefficient is 1, so t is the camera depth z */
  double D[3], O[3];
  for(int k = 0; k < 3; k++)
    D[k] = s.axis_right[k] * dx + s.axis_down[k] * dy + s.axis_fwd[k];
  O[0] = 0.0; O[1] = 0.0; O[2] = s.cam_height;

  double best = 1.0e30;

  /* ground plane z = 0 (unbounded) */
  if(D[2] < 0.0)
  {
    const double t = -O[2] / D[2];
    if(t > 0.0)
      best = t;
  }

  if(s.n_steps > 0)
  {
    /* stair-local frame: origin at the pivot, rotated by -yaw about the vertical */
    const double oy = O[1] - s.first_riser_y;
    double o[3], d[3];
    o[0] = s.yaw_cos * O[0] + s.yaw_sin * oy;
    o[1] = s.yaw_cos * oy - s.yaw_sin * O[0];
    o[2] = O[2];
    d[0] = s.yaw_cos * D[0] + s.yaw_sin * D[1];
    d[1] = s.yaw_cos * D[1] - s.yaw_sin * D[0];
    d[2] = D[2];

    const double hw = 0.5 * s.stair_width;
    const int K = s.n_steps;
    const double depth = K * s.tread + s.landing;
    for(int k = 0; k < K; k++)
    {
      /* riser k: plane y = k*tread, x in [-hw,hw], z in [k*rise,(k+1)*rise] */
      synth_hit(o, d, 1, k * s.tread, -hw, hw, k * s.rise, (k + 1) * s.rise, best);
      /* tread k+1: plane z = (k+1)*rise, x in [-hw,hw], y in [k*tread,(k+1)*tread] (+landing on the last) */
      const double yEnd = (k + 1 == K) ? depth : (k + 1) * s.tread;
      synth_hit(o, d, 2, (k + 1) * s.rise, -hw, hw, k * s.tread, yEnd, best);
      /* side faces under tread k+1 */
      synth_hit(o, d, 0, -hw, k * s.tread, yEnd, 0.0, (k + 1) * s.rise, best);
      synth_hit(o, d, 0, hw, k * s.tread, yEnd, 0.0, (k + 1) * s.rise, best);
    }
    /* back face */
    synth_hit(o, d, 1, depth, -hw, hw, 0.0, K * s.rise, best);
  }

  if(!(best < s.max_range))
    return;

  const uint64_t pix = uint64_t(v) * uint64_t(s.width) + uint64_t(u);
  const uint64_t k0 = synth_mix(frame_key ^ (pix * 0xD1342543DE82EF95ull));
  const uint64_t r1 = synth_mix(k0);
  const uint64_t r2 = synth_mix(k0 + 1);
  const uint64_t r3 = synth_mix(k0 + 2);
  const double inv32 = 1.0 / 4294967296.0;

  /* approx. standard normal: sum of four uniforms, centred and scaled to unit variance */
  const double usum = (double(uint32_t(r1)) + double(uint32_t(r1 >> 32)) + double(uint32_t(r2)) + double(uint32_t(r2 >> 32))) * inv32;
  const double g = (usum - 2.0) * 1.7320508075688772;

  double z = best + s.sigma * g;
  const double uo = double(uint32_t(r3)) * inv32;
  const double uv = double(uint32_t(r3 >> 32)) * inv32;
  if(uo < s.outlier_frac)
    z = s.outlier_min + uv * (s.outlier_max - s.outlier_min);
  else if(uv < s.invalid_frac)
    return;
  if(!(z > 0.0))
    return;

  out[0] = float(dx * z);
  out[1] = float(dy * z);
  out[2] = float(z);
}

/* the same pixel as the sensor would report it: camera depth z in units of depth_units, 0 = invalid */
__host__ __device__ inline unsigned short synth_depth_raw(const ssd_scene &s, uint64_t frame_key, int u, int v, float depthUnits)
{
  float p[3];
  synth_pixel(s, frame_key, u, v, p);
  if(!(p[2] > 0.0f))
    return 0;
  const float q = p[2] / depthUnits + 0.5f;
  return q >= 65535.0f ? static_cast<unsigned short>(65535) : static_cast<unsigned short>(static_cast<int>(q));
}

__host__ __device__ inline uint64_t synth_frame_key(const ssd_scene &s)
{
  return synth_mix(s.seed * 0x2545F4914F6CDD1Dull + 0x632BE59BD9B4E019ull);
}

} // namespace ssd

#endif /* SSD_SYNTH_H_ */
