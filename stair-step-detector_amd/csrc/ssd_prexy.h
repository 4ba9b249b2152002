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
#include <cstring>

namespace ssd
{

constexpr double kPreXYMaxInput = 64.0;       /* metres: an L515 ranges to 9 m; points beyond go through the doubles */

inline void make_pre_z(const PointParams &P, PreXY &Q);

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
  /* d's own bound per point (PreXY::dK, dE0): 4.01 * 2^-24 S(M) as above, 4.5 for the 4.01; a d from the doubles is D rounded once */
  Q.dK = std::nextafterf(static_cast<float>(4.5 * 0x1p-24 * std::fmax(sumX, sumY)), INFINITY);
  Q.dE0 = std::nextafterf(static_cast<float>(4.5 * 0x1p-24 * std::fmax(std::fabs(static_cast<double>(Q.c[3][0])), std::fabs(static_cast<double>(Q.c[3][1]))) + 0x1p-25), INFINITY);
  if(!(e < 0.25) || !std::isfinite(S))
  {
    Q.dK = 0.0f; Q.dE0 = 0x1p-24f;          /* every d comes from the doubles */
    /* a calibration or range for which single precision says nothing: every point takes the doubles */
    Q.lo = -1.0f; Q.hi = INFINITY;
    Q.maxInput = -1.0f;
    /* d then always comes from the doubles (K1's fallback rounds the exact value to single: 2^-24) */
    Q.boxLo = static_cast<float>(128.0 - 256.0 * 0x1p-20 - 0.5);
    Q.boxHi = static_cast<float>(128.0 + 256.0 * 0x1p-20 - 0.5);
    make_pre_z(P, Q);
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
  make_pre_z(P, Q);
  return Q;
}

/*
 * Round 6: the z row, the z-range test and the height bin in single precision first (PreXY::zc .. zCheckTop).
 *
 * The reference (pointcloud.cpp:150-178, transformation.h:59-64), all in doubles, every operation rounded:
 *     wz = ((a6 x + a7 y) + a8 z) + b2;    in range  <=>  wz > zMin && wz < zMax;    bin = (int)((wz - zMin) * recip)
 * Write T for the exact real value of ((a6 x + a7 y + a8 z + b2) - zMin) * recip (the doubles a, b, zMin, recip taken as they are)
 * and A = |a6 x| + |a7 y| + |a8 z| + |b2|.
 *   (i)  the reference's wz is within 4 * 2^-53 A of the real a.p + b2 (three products, three sums), so
 *          wz > zMin  <=  T >  dRef,     wz < zMin  <=  T < -dRef,     likewise around Tmax = (zMax - zMin) * recip,
 *        and its u = fl(fl(wz - zMin) * recip) is within 2^-53 recip (6 A + 2 |zMin|) of T:   dRef := 2^-50 recip (A + |zMin|);
 *   (ii) K1 evaluates t = fma(c0, x, fma(c1, y, fma(c2, z, c3))) in single precision, c_i = fl32(a_(6+i) * recip),
 *        c3 = fl32((b2 - zMin) * recip): as for x / y above, |t - T| <= 4.01 * 2^-24 S with S = (|c0| + |c1| + |c2|) M + |c3| for
 *        M = max(|x|, |y|, |z|) - and this bound is taken PER POINT (one v_max3_f32 and one FMA), not for a fixed input range:
 *        at 100 bins per metre a bound for 64 m would put one point in a hundred into the band, at the 1 - 3 m of a real frame it
 *        is one in five thousand.
 * With e(M) = zK M + zE0 >= 4.01 * 2^-24 S + dRef + (what the test itself rounds: fract, the subtraction, the FMA: below 2^-24) + eta:
 * a t farther than e(M) from every integer lies on the same side of every integer as T and as u - the bin is floor(t), and 0 is an
 * integer: the lower limit is decided -, and the upper limit is decided when Tmax is within eta <= 2^-20 of an integer N (any range
 * that is a whole number of height intervals; zTop = N): t < N then means T < Tmax - dRef.  Otherwise (zCheckTop) zTop = fl32(Tmax),
 * its rounding goes into zE0, and K1 calls points with |t - zTop| <= e(M) unsure as well.
 * The constants here take 4.5 for the 4.01 and add a floor of 2^-18 of a bin (0.04 um).  NaNs fail the ordered compare, an input
 * that overflows single precision has M > 10^30 and a negative threshold: both "unsure".  K1 sends every unsure point that the
 * other test has not called "outside" through the reference's doubles - all three rows, every compare, the bin.
 */
inline void make_pre_z(const PointParams &P, PreXY &Q)
{
  const double R = P.recip;
  double sumC = 0.0, amax = 0.0;
  for(int i = 0; i < 3; i++)
  {
    Q.zc[i] = static_cast<float>(P.a[6 + i] * R);
    sumC += std::fabs(static_cast<double>(Q.zc[i]));
    amax = std::fmax(amax, std::fabs(P.a[6 + i]));
  }
  Q.zc[3] = static_cast<float>((P.b[2] - P.zMin) * R);
  const long double tmaxL = (static_cast<long double>(P.zMax) - static_cast<long double>(P.zMin)) * static_cast<long double>(R);
  const double tmax = static_cast<double>(tmaxL);
  const double N = std::nearbyint(tmax);
  const double eta = static_cast<double>(tmaxL > static_cast<long double>(N) ? tmaxL - static_cast<long double>(N) : static_cast<long double>(N) - tmaxL);
  Q.zCheckTop = eta <= 0x1p-20 ? 0 : 1;
  const double top = Q.zCheckTop ? tmax : N;
  Q.zTop = static_cast<float>(top);
  const double K = 4.5 * 0x1p-24 * sumC + 0x1p-50 * 3.0 * amax * R;
  const double E0 = 4.5 * 0x1p-24 * std::fabs(static_cast<double>(Q.zc[3])) + 0x1p-50 * (std::fabs(P.b[2]) + std::fabs(P.zMin)) * R
                    + (Q.zCheckTop ? 0x1p-23 * std::fabs(top) : eta) + 0x1p-18;
  const bool usable = std::isfinite(K) && std::isfinite(E0) && E0 < 0.25 && K > 0.0 && top >= 1.0 && top < 8388608.0
                      && std::isfinite(static_cast<double>(Q.zc[3])) && std::isfinite(sumC);
  if(!usable)
  {
    /* a calibration or a range single precision says nothing about: the threshold is negative, every point takes the doubles */
    Q.zNegK = -1.0f; Q.zH0 = -1.0f;
    Q.zTop = 1.0f; Q.zCheckTop = 0;
    Q.zc[0] = Q.zc[1] = Q.zc[2] = Q.zc[3] = 0.0f;
  }
  else
  {
    Q.zNegK = -std::nextafterf(static_cast<float>(K), INFINITY);
    Q.zH0 = std::nextafterf(static_cast<float>(0.5 - E0), 0.0f);
  }
  unsigned int bits;
  static_assert(sizeof(bits) == sizeof(Q.zTop), "float is 32 bits");
  std::memcpy(&bits, &Q.zTop, sizeof(bits));
  Q.zTopBits = bits;
}

/*
 * Round 6: the pixel of a candidate point (K1, single pass) from the single-precision d of the range test.
 *
 * The reference (pointcloud.cpp:79-83), in doubles: ix = (int)((wx - xMin) * xToImage), iy = (int)((yMax - wy) * yToImage) with
 * xToImage = fl(W / fl(xMax - xMin)).  With D the exact centred coordinate of the range test (above), the exact real pixel coordinate
 * is PX = (D_x + 1/2) W (1 + rho), |rho| <= 2^-51 (the two roundings in xToImage), PY = (1/2 - D_y) H (1 + rho'), and the
 * reference's own value lies within 2^-50 xToImage (A + |xMin|) of PX (as dRef above).  K1 has d with |d - D| <= 4.01 * 2^-24 S(M)
 * (a lane that went through the doubles has d = fl32(D): 2^-26) and evaluates px = fma(d.x, W, W / 2): one more rounding, at most
 * 2^-24 W.  So |px - PX| <= W (4.01 * 2^-24 S_x(M) + 2^-23) =: and a px farther than that (plus the reference's own 2^-50 ..) from
 * every integer truncates to the reference's ix; 0 and W are integers, so such a pixel also lies inside the image (quirk Q5 - a
 * point within an ulp of the range's edge that lands on column W - is a matter for the doubles).  One threshold for both
 * coordinates: the larger bound.  Constants as in make_pre_z(): 4.5 for 4.01, a floor of 2^-18 of a pixel.
 */
inline void make_pre_pixel(const PointParams &P, const PreXY &Q, PixelParams &X)
{
  const double W = X.W, H = X.H;
  X.fW = static_cast<float>(W); X.fHalfW = static_cast<float>(0.5 * W);
  X.fNegH = static_cast<float>(-H); X.fHalfH = static_cast<float>(0.5 * H);
  const double sumX = std::fabs(static_cast<double>(Q.c[0][0])) + std::fabs(static_cast<double>(Q.c[1][0])) + std::fabs(static_cast<double>(Q.c[2][0]));
  const double sumY = std::fabs(static_cast<double>(Q.c[0][1])) + std::fabs(static_cast<double>(Q.c[1][1])) + std::fabs(static_cast<double>(Q.c[2][1]));
  double amax = 0.0;
  for(int i = 0; i < 6; i++)
    amax = std::fmax(amax, std::fabs(P.a[i]));
  const double toImg = std::fmax(X.xToImage, X.yToImage);
  const double K = 4.5 * 0x1p-24 * std::fmax(W * sumX, H * sumY) + 0x1p-50 * 3.0 * amax * toImg;
  const double E0 = std::fmax(W * (4.5 * 0x1p-24 * std::fabs(static_cast<double>(Q.c[3][0])) + 0x1p-22), H * (4.5 * 0x1p-24 * std::fabs(static_cast<double>(Q.c[3][1])) + 0x1p-22))
                    + 0x1p-50 * (std::fmax(std::fabs(P.b[0]), std::fabs(P.b[1])) + std::fmax(std::fabs(P.xMin), std::fabs(P.yMax))) * toImg + 0x1p-18;
  /* width and height are exact in single precision up to 2^24; d must be the pre-filter's (a handle whose range test always takes
   * the doubles has d = fl32(D), which the bound covers as well) */
  const bool usable = std::isfinite(K) && std::isfinite(E0) && E0 < 0.25 && K > 0.0 && W <= 8192.0 && H <= 8192.0;
  if(usable)
  {
    X.pxNegK = -std::nextafterf(static_cast<float>(K), INFINITY);
    X.pxH0 = std::nextafterf(static_cast<float>(0.5 - E0), 0.0f);
  }
  else
  {
    X.pxNegK = -1.0f; X.pxH0 = -1.0f;        /* negative threshold: every candidate's pixel takes the doubles */
  }
}

} // namespace ssd

#endif /* SSD_PREXY_H_ */
