/*
 * ssd_math.h — small fp64 helpers of the kernels that must match the host libm / the reference bit for bit
 * (std::hypot, LineCoordinates, Line<double>::normalized / intersection).  Shared by ssd_kernels.hip and the test
 * hooks (ssd_testhooks.hip), host and device.
 */
#ifndef SSD_MATH_H_
#define SSD_MATH_H_

#include <hip/hip_runtime.h>
#include <cmath>

namespace ssd
{

/* small fp64 helpers that must match the host libm bit for bit                */

/* std::hypot as glibc 2.35 computes it without FMA (sysdeps/ieee754/dbl-64/e_hypot.c): the oracle
 * calls the host's hypot, the device restates the published algorithm; tests/test_oracle.py checks
 * the two agree on this image. */
__host__ __device__ inline double hypot_kernel(double ax, double ay)
{
  double t1, t2;
  double h = sqrt(ax * ax + ay * ay);
  if(h <= 2.0 * ay)
  {
    const double delta = h - ay;
    t1 = ax * (2.0 * delta - ax);
    t2 = (delta - 2.0 * (ax - ay)) * delta;
  }
  else
  {
    const double delta = h - ax;
    t1 = 2.0 * delta * (ax - 2.0 * ay);
    t2 = (4.0 * delta - ay) * ay + delta * delta;
  }
  h -= (t1 + t2) / (2.0 * h);
  return h;
}

__host__ __device__ inline double hypot_ref(double x, double y)
{
  const double kScale = 0x1p-600, kLarge = 0x1p+511, kTiny = 0x1p-459, kEps = 0x1p-54;
  x = fabs(x);
  y = fabs(y);
  double ax = x < y ? y : x;
  double ay = x < y ? x : y;
  if(ax > kLarge)
  {
    if(ay <= ax * kEps)
      return ax + ay;
    return hypot_kernel(ax * kScale, ay * kScale) / kScale;
  }
  if(ay < kTiny)
  {
    if(ax >= ay / kEps)
      return ax + ay;
    return hypot_kernel(ax / kScale, ay / kScale) * kScale;
  }
  if(ax * kEps >= ay)
    return ax + ay;
  return hypot_kernel(ax, ay);
}

struct LineD { double a, b, c; };
struct LineI { int a, b, c; };

/* LineCoordinates(p, q), types.h:140-158 */
__host__ __device__ __forceinline__ LineI line_through_i(int x1, int y1, int x2, int y2)
{
  return { y2 - y1, x1 - x2, x2 * y1 - x1 * y2 };
}
__host__ __device__ __forceinline__ LineD line_through_d(double x1, double y1, double x2, double y2)
{
  return { y2 - y1, x1 - x2, x2 * y1 - x1 * y2 };
}
/* Line<double>::normalized, segmentation.cpp:383-387 */
__device__ __forceinline__ LineD normalized_line(double a, double b, double c)
{
  const double h = hypot_ref(a, b);
  return { a / h, b / h, c / h };
}
/* Line<double>::intersection, segmentation.cpp:344-362; returns false for angles <= 60 degrees */
__host__ __device__ __forceinline__ bool intersect60(const LineD &l, const LineD &o, double &x, double &y)
{
  const double kTan60 = 1.7320508075688772;       /* std::numbers::sqrt3 */
  const double numerator = l.a * o.b - o.a * l.b;
  const double denominator = l.a * o.a + l.b * o.b;
  if(fabs(numerator) > fabs(denominator) * kTan60)
  {
    x = (l.b * o.c - o.b * l.c) / numerator;
    y = (o.a * l.c - l.a * o.c) / numerator;
    return true;
  }
  return false;
}

} // namespace ssd

#endif /* SSD_MATH_H_ */
