/*
 * ssd_prexy.h — constants and error bound of K1's single-precision pre-filter of the x / y range test (PreXY, ssd_device.h),
 * as a plain host function: make_params() calls it, the test hook ssd_test_prexy hands it to the tests.
 *
 * The reference takes a point when its world x and y, computed in doubles (transformation.h:59-64), lie strictly inside the
 * measuring range (pointcloud.cpp:150-165).  Write the exact (real-number) value of the centred, normalised coordinate as
 *     D = ((a0 x + a1 y + a2 z + b) - min) / (max - min) - 1/2          in range  <=>  |D| < 1/2
 * K1 evaluates d = fma(c0, x, fma(c1, y, fma(c2, z, c3))) in single precision with c_i = fl(a_i / (max - min)),
 * c3 = fl((b - min) / (max - min) - 1/2).  For inputs with max(|x|, |y|, |z|) <= R:
 *   - each coefficient is off by at most 2^-24 |c_i| (round to nearest; the double-precision quotient's own error is 2^-29 of that),
 *   - each of the three FMAs rounds once, by at most 2^-24 of its result, and every partial result is bounded by
 *     S = (|c0| + |c1| + |c2|) R + |c3| (up to a factor 1 + 2^-22),
 *   so |d - D| <= 2^-24 (S + 3 S) (1 + 2^-22) < 4.01 * 2^-24 S.
 * The reference's own doubles are within 4 * 2^-53 (|a0 x| + |a1 y| + |a2 z| + |b|) of the real value: 1e-13 of the range
 * for R = 64 m.  e = 8 * 2^-24 S + 2^-20 leaves a factor of two and a floor of one micro-range on top of both.
 *   max(|dx|, |dy|) < 0.5 - e   =>  |D| < 0.5 - e + 4.01 * 2^-24 S < 0.5 - 2^-20: the doubles say inside
 *   max(|dx|, |dy|) > 0.5 + e   =>  one |D| > 0.5 + 2^-20:                        the doubles say outside
 * Everything between, inputs beyond R and NaNs (every comparison false) go through the doubles.  Subnormal coefficients or
 * products can be flushed or not: either way the absolute error is below 2^-126, inside the floor.
 * The boxes: the exact grid coordinate (D + 1/2) * 256 of a point lies within 256 e of (d + 1/2) * 256, so
 * floor((dMin + 1/2 - e) * 256) .. floor((dMax + 1/2 + e) * 256) holds every in-range point of the cell; floor(t) is taken
 * as rn(t - 1/2) by v_cvt_pk_u8_f32 (round to nearest even, saturating at 0 and 255: probed, tools/cvt_probe.hip), which
 * differs from floor(t) only for an integer t, and then by -1: for a minimum that widens the box, for a maximum the box still
 * ends at (dMax + 1/2 + e), beyond the point by the slack in e.
 *
 * Inputs beyond R.  The x / y decision only matters for a point whose z test - in doubles, always - said "in range": then
 * |w_z - b_z| <= Z = max(|zMin - b_z|, |zMax - b_z|).  With sigma the smallest singular value of A (1 for a rotation),
 * |A p| >= sigma |p|_2 >= sigma |p|_inf, so for |p|_inf = t > R one of |w_x - b_x|, |w_y - b_y| is at least
 * G(t) = sqrt((sigma^2 t^2 - Z^2) / 2), its |D| at least (G(t) - off) * smin with off = max(|b_x - x centre|, |b_y - y centre|)
 * and smin the smaller of the two 1 / (max - min), while d is off by at most 4.01 * 2^-24 (|c|_1 t + |c3|): a term that grows
 * ten million times slower in t than G.  make_pre_xy() evaluates the worst case t = R with a wide margin (the far |D| at least 4,
 * eight ranges out) and, when it holds - any calibration that is a rotation seen from within a few dozen metres -, clears
 * checkInput: K1 then skips the per-point test of the input's magnitude, because such a point reads max(|dx|, |dy|) > hi:
 * "outside", as the doubles say.  (An overflow to infinity reads the same; NaNs fail the z test.)  Otherwise K1 tests
 * max(|x|, |y|, |z|) <= R per point and sends the rest through the doubles.
 */
#ifndef SSD_PREXY_H_
#define SSD_PREXY_H_

#include "ssd_device.h"
#include <cmath>

namespace ssd
{

constexpr double kPreXYMaxInput = 64.0;       /* metres: an L515 ranges to 9 m; points beyond go through the doubles */

inline PreXY make_pre_xy(const PointParams &P)
{
  PreXY Q{};
  const double R = kPreXYMaxInput;
  const double sx = 1.0 / (P.xMax - P.xMin), sy = 1.0 / (P.yMax - P.yMin);
  double sumX = 0.0, sumY = 0.0;
  for(int i = 0; i < 3; i++)
  {
    Q.c[i][0] = static_cast<float>(P.a[i] * sx);
    Q.c[i][1] = static_cast<float>(P.a[3 + i] * sy);
    sumX += std::fabs(static_cast<double>(Q.c[i][0]));
    sumY += std::fabs(static_cast<double>(Q.c[i][1]));
  }
  Q.c[3][0] = static_cast<float>((P.b[0] - P.xMin) * sx - 0.5);
  Q.c[3][1] = static_cast<float>((P.b[1] - P.yMin) * sy - 0.5);
  const double S = std::fmax(sumX * R + std::fabs(static_cast<double>(Q.c[3][0])), sumY * R + std::fabs(static_cast<double>(Q.c[3][1])));
  const double e = 8.0 * 0x1p-24 * S + 0x1p-20;
  Q.maxInput = static_cast<float>(R);
  Q.checkInput = 1;
  if(!(e < 0.25) || !std::isfinite(S))
  {
    /* a calibration or range for which single precision says nothing: every point takes the doubles */
    Q.lo = -1.0f; Q.hi = INFINITY;
    Q.maxInput = -1.0f;
    /* d then always comes from the doubles (K1's fallback rounds the exact value to single: 2^-24) */
    Q.boxLo = static_cast<float>(128.0 - 256.0 * 0x1p-20 - 0.5);
    Q.boxHi = static_cast<float>(128.0 + 256.0 * 0x1p-20 - 0.5);
    return Q;
  }
  Q.lo = std::nextafterf(static_cast<float>(0.5 - e), 0.0f);
  Q.hi = std::nextafterf(static_cast<float>(0.5 + e), 1.0f);
  /* inputs beyond R (see above): sigma from || A^T A - I ||_F = delta (lambda_min(A^T A) >= 1 - delta) */
  {
    double delta2 = 0.0, amax = 0.0;
    for(int i = 0; i < 3; i++)
      for(int j = 0; j < 3; j++)
      {
        double g = 0.0;
        for(int k = 0; k < 3; k++)
          g += P.a[3 * k + i] * P.a[3 * k + j];
        g -= i == j ? 1.0 : 0.0;
        delta2 += g * g;
        amax = std::fmax(amax, std::fabs(P.a[3 * i + j]));
      }
    const double delta = std::sqrt(delta2);
    if(delta < 0.5)
    {
      const double sigma = std::sqrt(1.0 - delta);
      const double Z = std::fmax(std::fabs(P.zMin - P.b[2]), std::fabs(P.zMax - P.b[2]));
      const double off = std::fmax(std::fabs(P.b[0] - 0.5 * (P.xMin + P.xMax)), std::fabs(P.b[1] - 0.5 * (P.yMin + P.yMax)));
      const double smin = std::fmin(sx, sy), smax = std::fmax(sx, sy);
      const double g2 = sigma * sigma * R * R - Z * Z;
      if(g2 > 0.0)
      {
        const double far = (std::sqrt(0.5 * g2) - off) * smin                                   /* the far coordinate's |D| at t = R */
                           - 8.0 * 0x1p-24 * (3.0 * amax * smax * R + std::fabs(static_cast<double>(Q.c[3][0])) + std::fabs(static_cast<double>(Q.c[3][1])));
        /* growth in t: sigma kappa smin / sqrt(2) against 8 * 2^-24 * 3 amax smax - the first must dominate for all t >= R */
        const double kappa = std::sqrt(g2) / (sigma * R);
        const bool grows = sigma * kappa * smin * 0.70710678 > 16.0 * 0x1p-24 * 3.0 * amax * smax;
        if(far >= 4.0 && grows)
          Q.checkInput = 0;
      }
    }
  }
  /* rn(d * 256 + boxLo) = floor((d + 1/2 - e) * 256) (or one less), rn(d * 256 + boxHi) = floor((d + 1/2 + e) * 256) (or one
   * less for an integer); the single-precision FMA's own rounding (2^-17 of a grid cell) is inside e's slack */
  Q.boxLo = static_cast<float>(128.0 - 256.0 * e - 0.5);
  Q.boxHi = static_cast<float>(128.0 + 256.0 * e - 0.5);
  return Q;
}

} // namespace ssd

#endif /* SSD_PREXY_H_ */
