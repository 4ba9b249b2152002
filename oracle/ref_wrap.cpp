/*
 * ref_wrap.cpp — C entry points around the two reference translation units that
 * compile from their own sources with no third-party dependency:
 *   /root/reference/stairs.cpp            (Stairs::serialize, stairs.cpp:55-70)
 *   /root/reference/quadrilateralTest.cpp (QuadrilateralTest, quadrilateralTest.cpp:275-451)
 * The reference sources are compiled where they lie (see Makefile target `ref`);
 * nothing of them is copied into this repository.  Output: oracle/_ref/libssd_ref.so.
 * TEST INFRASTRUCTURE ONLY: used to pin oracle/ssd_oracle.cpp's restatement of
 * these two pieces against the real reference code.
 */
#include "stairs.h"
#include "quadrilateralTest.h"
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

extern "C" int ssdref_serialize(int n_steps, const double *steps_ext /* n x 9: height, 4 x (x,y) */, char *buf, int cap)
{
  stairs::Stairs s;
  for(int i = 0; i < n_steps; i++)
  {
    const double *p = steps_ext + 9 * i;
    stairs::Stairs::StairStep st;
    st.height = p[0];
    for(int k = 0; k < 4; k++)
      st.quadrilateral[k] = stairs::Point2(p[1 + 2 * k], p[2 + 2 * k]);
    s.stairSteps.push_back(st);
  }
  const std::string line = s.serialize();
  std::snprintf(buf, cap, "%s", line.c_str());
  return int(line.size());
}

/* 0 = ok, negative = the constructor threw (code by message, same numbering as ssdo_quad_test) */
extern "C" int ssdref_quad_test(const double quad[8], const double *pts_xy, int n, uint8_t *inside)
{
  const stairs::Quadrilateral_t q{ stairs::Point2(quad[0], quad[1]), stairs::Point2(quad[2], quad[3]),
                                   stairs::Point2(quad[4], quad[5]), stairs::Point2(quad[6], quad[7]) };
  try
  {
    const stairs::QuadrilateralTest qt(q);
    for(int i = 0; i < n; i++)
      inside[i] = qt.isPointWithin(stairs::Point2(pts_xy[2 * i], pts_xy[2 * i + 1])) ? 1 : 0;
  }
  catch(const std::invalid_argument &e)
  {
    const std::string m = e.what();
    if(m.find("not convex") != std::string::npos) return -1;
    if(m.find("along Y") != std::string::npos) return -2;
    if(m.find("along X") != std::string::npos) return -3;
    if(m.find("more than 2") != std::string::npos) return -4;
    if(m.find("2 empty") != std::string::npos) return -5;
    if(m.find("2 double") != std::string::npos) return -6;
    return -99;
  }
  return 0;
}
