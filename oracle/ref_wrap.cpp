/*
 * ref_wrap.cpp — C entry points around the reference pieces that compile from their own sources with no
 * third-party dependency:
 *   /root/reference/stairs.cpp            (Stairs::serialize, stairs.cpp:55-70)
 *   /root/reference/quadrilateralTest.cpp (QuadrilateralTest, quadrilateralTest.cpp:275-451)
 *   /root/reference/calibrationTriangle.cpp (CalibrationTriangle::load / isValid, calibrationTriangle.cpp:97-172)
 *   /root/reference/configuration.h       (struct Configuration: the compile-time constants, configuration.h:27-52)
 *   /root/reference/types.h               (LineCoordinates<T>: the line through two points and its three determinants, types.h:117-163)
 * The reference sources are compiled where they lie (see Makefile target `ref`);
 * nothing of them is copied into this repository.  Output: oracle/_ref/libssd_ref.so.
 * TEST INFRASTRUCTURE ONLY: used to pin oracle/ssd_oracle.cpp's restatement of
 * these two pieces against the real reference code.
 */
#include "stairs.h"
#include "quadrilateralTest.h"
#include "calibrationTriangle.h"
#include "configuration.h"
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

#include <unistd.h>

/* CalibrationTriangle::load() reads "calibration-triangle" from the working directory: run it inside `dir`.
 * returns 0 = loaded and valid (world[9], *side: 1 left, 2 right), 1 = load failed, 2 = not valid */
extern "C" int ssdref_load_triangle(const char *dir, double world[9], int *side)
{
  char cwd[4096];
  if(!getcwd(cwd, sizeof cwd) || chdir(dir) != 0)
    return -1;
  stairs::CalibrationTriangle t;
  const int rc = t.load();
  const bool valid = rc == 0 && t.isValid();
  if(chdir(cwd) != 0)
    return -1;
  if(rc != 0)
    return 1;
  const auto &c = t.getTriangleCorners();
  for(int i = 0; i < 3; i++)
  {
    world[3 * i] = c[i].x; world[3 * i + 1] = c[i].y; world[3 * i + 2] = c[i].z;
  }
  *side = t.getLowerQuadrant() == stairs::CalibrationTriangle::Side::left ? 1 : (t.getLowerQuadrant() == stairs::CalibrationTriangle::Side::right ? 2 : 0);
  return valid ? 0 : 2;
}

/* the reference's compile-time Configuration (configuration.h:27-52), default-constructed:
 * out = x.min, x.max, y.min, y.max, z.min, z.max, heightInterval, minHeightAboveGround, minStepDepth; wh = depth stream width, height */
extern "C" void ssdref_configuration(double out[9], int wh[2])
{
  const stairs::Configuration c;
  out[0] = c.measuringRange.x.min; out[1] = c.measuringRange.x.max;
  out[2] = c.measuringRange.y.min; out[3] = c.measuringRange.y.max;
  out[4] = c.measuringRange.z.min; out[5] = c.measuringRange.z.max;
  out[6] = c.heightInterval; out[7] = c.minHeightAboveGround; out[8] = c.minStepDepth;
  wh[0] = c.streams.depth.width; wh[1] = c.streams.depth.height;
}

/* LineCoordinates<T> (types.h:117-163; included through stairs.h): the base of every line of the path (Line<int>, Line<double>,
 * segmentation.cpp:321-407; StairsDetector::Line, pointcloud.cpp:514-526).  Its coefficients and determinants are protected:
 * a derived class opens them. */
namespace
{
template<typename T>
struct OpenLine : stairs::LineCoordinates<T>
{
  struct P { T x, y; };
  OpenLine(const P &p, const P &q) : stairs::LineCoordinates<T>(p, q) {}
  OpenLine(T a, T b, T c) : stairs::LineCoordinates<T>(a, b, c) {}
  void coefficients(T out[3]) const { out[0] = this->_a; out[1] = this->_b; out[2] = this->_c; }
  void dets(const OpenLine &o, T out[3]) const { out[0] = this->det(o); out[1] = this->detx(o); out[2] = this->dety(o); }
};
}
/* line through p and q: abc[3] */
extern "C" void ssdref_line_d(const double pq[4], double abc[3])
{
  OpenLine<double>(OpenLine<double>::P{ pq[0], pq[1] }, OpenLine<double>::P{ pq[2], pq[3] }).coefficients(abc);
}
extern "C" void ssdref_line_i(const int pq[4], int abc[3])
{
  OpenLine<int>(OpenLine<int>::P{ pq[0], pq[1] }, OpenLine<int>::P{ pq[2], pq[3] }).coefficients(abc);
}
/* det, detx, dety of two lines given by their coefficients */
extern "C" void ssdref_line_dets_d(const double l[3], const double o[3], double out[3])
{
  OpenLine<double>(l[0], l[1], l[2]).dets(OpenLine<double>(o[0], o[1], o[2]), out);
}

/* CalibrationTriangle::load() in `dir_in`, then ::save() (calibrationTriangle.cpp:127-146) in `dir_out`: the reference's own
 * writer of the file the loaders read.  0 = ok, 1 = load failed, 3 = save failed */
extern "C" int ssdref_resave_triangle(const char *dir_in, const char *dir_out)
{
  char cwd[4096];
  if(!getcwd(cwd, sizeof cwd) || chdir(dir_in) != 0)
    return -1;
  stairs::CalibrationTriangle t;
  const int rc = t.load();
  int rs = -1;
  if(rc == 0 && chdir(cwd) == 0 && chdir(dir_out) == 0)
    rs = t.save();
  if(chdir(cwd) != 0)
    return -1;
  return rc != 0 ? 1 : (rs != 0 ? 3 : 0);
}
