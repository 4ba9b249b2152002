/*
 * ssd_bestline.h — the residual of the line through two points of a list (ApproximationLine, segmentation.cpp:409-487): the sum
 * of the n smallest point distances |a x + b y + c| over n * hypot(a, b), in the three forms the kernels use (any list; keys in
 * passes of four; one pass with the smallest kept sorted).  Host + device code: k_outline / k_final run the device build, the
 * CPU test suite the host build (ssd_test_best_line_host, ssd_testhooks.hip) against the oracle's BestLine.
 */
#ifndef SSD_BESTLINE_H_
#define SSD_BESTLINE_H_

#include "ssd_math.h"

namespace ssd
{

/* residual of the line through points p and q: sum of the n smallest |a x + b y + c| of the other
 * points, over n * hypot(a, b) */
__host__ __device__ inline double line_residual_generic(const int *px, const int *py, int m, int p, int q, LineI &line)
{
  line = line_through_i(px[p], py[p], px[q], py[q]);
  if(m <= 2)
    return 0.0;
  const int nd = m - 2;
  const int n = nd > 4 ? (nd - 1) / 2 : 1;
  /* n rounds of "next smallest (distance, index)" — no per-lane storage */
  int sum = 0;
  int lastD = -1, lastI = -1;
  for(int r = 0; r < n; r++)
  {
    int bestD = 0x7fffffff, bestI = 0x7fffffff;
    for(int i = 0; i < m; i++)
    {
      if(i == p || i == q)
        continue;
      const int dv = px[i] * line.a + py[i] * line.b + line.c;
      const int d = dv < 0 ? -dv : dv;
      const bool after = d > lastD || (d == lastD && i > lastI);
      const bool better = d < bestD || (d == bestD && i < bestI);
      if(after && better) { bestD = d; bestI = i; }
    }
    sum += bestD;
    lastD = bestD;
    lastI = bestI;
  }
  return sum / (n * hypot_ref(static_cast<double>(line.a), static_cast<double>(line.b)));
}

__host__ __device__ __forceinline__ unsigned int umin2(unsigned int a, unsigned int b) { return a < b ? a : b; }
__host__ __device__ __forceinline__ unsigned int umax2(unsigned int a, unsigned int b) { return a > b ? a : b; }
__host__ __device__ __forceinline__ unsigned int umed3(unsigned int a, unsigned int b, unsigned int c)
{
  return umax2(umin2(a, b), umin2(umax2(a, b), c));       /* v_med3_u32 */
}
/* x * y for |x|, |y| < 2^23 (v_mul_i32_i24 on the device) */
__host__ __device__ __forceinline__ int mul24i(int x, int y)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __mul24(x, y);
#else
  return x * y;
#endif
}

/* The same sum for images with 3 W H < 2^25 and m <= 128 (every distance fits 25 bits: |a| <= H, |b| <= W,
 * |c| <= W H).  The two points of the pair lie on their line exactly (integer arithmetic), so the n smallest
 * distances of the others are the n + 2 smallest of all m, minus two zeros: no exclusion tests.  Distances
 * become distinct keys (d << 7 | i) + 1 and every pass over the points extracts the next FOUR smallest keys
 * with a min / med3 insertion network, in registers. */
__host__ __device__ inline double line_residual_keys(const int *px, const int *py, int m, int p, int q, LineI &line)
{
  line = line_through_i(px[p], py[p], px[q], py[q]);
  if(m <= 2)
    return 0.0;
  const int nd = m - 2;
  const int n = nd > 4 ? (nd - 1) / 2 : 1;
  int need = n + 2;
  unsigned int sum = 0, last = 0;
  while(need > 0)
  {
    unsigned int b1 = 0xffffffffu, b2 = 0xffffffffu, b3 = 0xffffffffu, b4 = 0xffffffffu;
    for(int i = 0; i < m; i++)
    {
      const int dv = px[i] * line.a + py[i] * line.b + line.c;
      const unsigned int d = static_cast<unsigned int>(dv < 0 ? -dv : dv);
      unsigned int k = ((d << 7) | static_cast<unsigned int>(i)) + 1u;
      k = k > last ? k : 0xffffffffu;
      const unsigned int n4 = umed3(b3, b4, k), n3 = umed3(b2, b3, k), n2 = umed3(b1, b2, k);
      b1 = umin2(b1, k); b2 = n2; b3 = n3; b4 = n4;
    }
    /* need <= number of keys above `last`, so the ones taken are real */
    sum += (b1 - 1u) >> 7; last = b1;
    if(need > 1) { sum += (b2 - 1u) >> 7; last = b2; }
    if(need > 2) { sum += (b3 - 1u) >> 7; last = b3; }
    if(need > 3) { sum += (b4 - 1u) >> 7; last = b4; }
    need -= 4;
  }
  return static_cast<int>(sum) / (n * hypot_ref(static_cast<double>(line.a), static_cast<double>(line.b)));
}

/* The same sum in ONE walk over the points (lists of at most 64 points, images with 3 W H < 2^25): the D >= n + 2 smallest
 * distances are kept sorted in registers (insertion by a min / med3 network, D operations per point), so nothing has to
 * tell a later pass which ones were taken already — plain distances, equal ones included, sum up to the same total
 * whichever of them is counted.  The products are 24-bit multiplies (|a| <= H, |b| <= W, coordinates < 2^12: exact).
 * 11 + D vector instructions per point against 15 per point and pass of four.  get(i, x, y) hands out point i (the kernels:
 * v_readlane of the lanes' registers). */
template<int D, typename Get>
__host__ __device__ __forceinline__ unsigned int smallest_sum(Get get, int m, const LineI &line, int need)
{
  unsigned int b[D];
#pragma unroll
  for(int j = 0; j < D; j++)
    b[j] = 0xffffffffu;
  for(int i = 0; i < m; i++)
  {
    int x, y;
    get(i, x, y);
    const int s = mul24i(x, line.a) + mul24i(y, line.b) + line.c;
    const unsigned int d = static_cast<unsigned int>(s < 0 ? -s : s);
#pragma unroll
    for(int j = D - 1; j > 0; j--)
      b[j] = umed3(b[j - 1], b[j], d);
    b[0] = umin2(b[0], d);
  }
  unsigned int sum = 0;
#pragma unroll
  for(int j = 0; j < D; j++)
    sum += j < need ? b[j] : 0u;
  return sum;
}

/* residual of `line` (through two of the m points) by the one-pass form */
template<typename Get>
__host__ __device__ __forceinline__ double line_residual_onepass(Get get, int m, const LineI &line)
{
  if(m <= 2)
    return 0.0;
  const int nd = m - 2;
  const int n = nd > 4 ? (nd - 1) / 2 : 1;
  const int need = n + 2;                      /* <= 32 for m <= 64; the pair's own two points are the two zeros */
  unsigned int sum;
  if(need <= 4) sum = smallest_sum<4>(get, m, line, need);
  else if(need <= 8) sum = smallest_sum<8>(get, m, line, need);
  else if(need <= 12) sum = smallest_sum<12>(get, m, line, need);
  else if(need <= 16) sum = smallest_sum<16>(get, m, line, need);
  else if(need <= 24) sum = smallest_sum<24>(get, m, line, need);
  else sum = smallest_sum<32>(get, m, line, need);
  return static_cast<int>(sum) / (n * hypot_ref(static_cast<double>(line.a), static_cast<double>(line.b)));
}

} // namespace ssd

#endif /* SSD_BESTLINE_H_ */
