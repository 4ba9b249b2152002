/*
 * ssd_predict.h — the table the single pass's predictor makes of its sample histogram (k_predict's last step), stated as a
 * plain function for host and device.  k_predict computes the same table with one thread per bin (ballots and rank counting);
 * the test hook ssd_test_predict_table_host runs THIS statement, tests/test_predict.py checks its properties on the CPU and
 * tests/test_gpu_single_pass.py that the kernel's table equals it on every frame of a batch.
 *
 * sample[b]: points of height bin b among the sampled runs (one run of 16 points in every kSpecSample).
 * plane[b]:  the plane (bit image) K1 rasters bin b's points into, 0xff = none.  Returns the number of planes.
 *
 * Rules ("The single pass"):
 *  - candidate peak: minHeight <= b < nBins - 1 (a step plateau's peak, pointcloud.cpp:402-418), a local maximum of the
 *    sample (c > left, c >= right), >= 1200 points scaled up, the neighbours' sum below 1.5 c plus 8 samples (filterPeaks,
 *    pointcloud.cpp:243-256, on the sample);
 *  - at most kMaxPlanes / 3 candidates: the fullest (ties: the lower bin);
 *  - a chosen peak gets a plane for its bin, and for each neighbour unless the OTHER neighbour is more than twice as full (+ 8):
 *    the plateau is the peak's bin and the fuller neighbour (extractPlateauPoints, pointcloud.cpp:300-335);
 *  - the peak and a neighbour beyond doubt share ONE plane; otherwise every bin has its own;
 *  - more planes than kMaxPlanes: none at all (cannot happen with eight candidates of three bins; kept as a guard).
 * sabotage (tests): 1 = the table three bins above where it belongs, 2 = no planes.
 */
#ifndef SSD_PREDICT_H_
#define SSD_PREDICT_H_

#include "ssd_device.h"

namespace ssd
{

__host__ __device__ inline bool predict_candidate(const unsigned int *sample, int nBins, int minHeight, int b)
{
  if(b < (minHeight > 1 ? minHeight : 1) || b >= nBins - 1)
    return false;
  const unsigned int c = sample[b], l = sample[b - 1], r = sample[b + 1];
  return c > l && c >= r && c * static_cast<unsigned int>(kSpecSample) >= 1200u && (l + r) * 2u < 3u * c + 16u;
}

__host__ __device__ inline int predict_table(const unsigned int *sample, int nBins, int minHeight, int sabotage, unsigned char *plane)
{
  unsigned char code[kMaxBins];          /* 1 = chosen peak, 2 = its lower neighbour wanted, 4 = its upper neighbour wanted */
  for(int b = 0; b < kMaxBins; b++)
  {
    code[b] = 0;
    plane[b] = 0xff;
  }
  if(sabotage == 2)
    return 0;
  for(int b = 0; b < nBins && b < kMaxBins; b++)
  {
    if(!predict_candidate(sample, nBins, minHeight, b))
      continue;
    int fuller = 0;
    for(int o = 0; o < nBins && o < kMaxBins; o++)
      if(o != b && predict_candidate(sample, nBins, minHeight, o) && (sample[o] > sample[b] || (sample[o] == sample[b] && o < b)))
        fuller++;
    if(fuller >= kMaxPlanes / 3)
      continue;
    const unsigned int l = sample[b - 1], r = sample[b + 1];
    code[b] = static_cast<unsigned char>(1u | (r > 2u * l + 8u ? 0u : 2u) | (l > 2u * r + 8u ? 0u : 4u));
  }
  const int shift = sabotage == 1 ? 3 : 0;
  auto code_of = [&](int k) -> unsigned int
  {
    k -= shift;
    return k >= 0 && k < kMaxBins ? code[k] : 0u;
  };
  auto wanted = [&](int k) -> bool
  {
    return k >= 0 && k < nBins && ((code_of(k - 1) & 4u) | (code_of(k) & 1u) | (code_of(k + 1) & 2u)) != 0u;
  };
  int n = 0;
  for(int b = 0; b < nBins && b < kMaxBins; b++)
  {
    if(!wanted(b))
      continue;
    const bool withBelow = wanted(b - 1) && (code_of(b - 1) == 5u || code_of(b) == 3u);
    if(!withBelow)
      n++;
    plane[b] = static_cast<unsigned char>(n - 1);
  }
  if(n > kMaxPlanes)
  {
    for(int b = 0; b < kMaxBins; b++)
      plane[b] = 0xff;
    return 0;
  }
  return n;
}

} // namespace ssd

#endif /* SSD_PREDICT_H_ */
