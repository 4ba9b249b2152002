/*
 * ssd_oracle.cpp — CPU oracle: a restatement of the per-frame point-cloud path
 * of peter-nebe/stair-step-detector.  TEST INFRASTRUCTURE ONLY (see ssd_oracle.h).
 *
 * Written as "C with vectors" in C++17 rather than C99 for three library
 * semantics the reference relies on and that must be the same calls here:
 * std::sort (segmentation.cpp:724, tie order), std::min_element ("first
 * minimum", segmentation.cpp:440,470) and ostream fixed/setprecision(3)
 * (stairs.cpp:36,43).  The C entry points are declared in ssd_oracle.h.
 *
 * Build: g++ -std=c++17 -O2 -ffp-contract=off, no -march, no -ffast-math —
 * the reference's CMakeLists.txt:23-28 sets only the language standard, so its
 * arithmetic is plain IEEE double, one rounding per operation, no FMA.
 *
 * PARITY STATUS: see ssd_oracle.h.  Each function cites the reference
 * file:line it follows (paths relative to /root/reference).
 */
#include "ssd_oracle.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <optional>
#include <sstream>
#include <string>
#include <memory>
#include <vector>

namespace
{

struct P2 { double x = 0, y = 0; };
struct P3 { double x = 0, y = 0, z = 0; };
struct P2i { int x, y; };
typedef std::vector<P2i> Pts2i;

/* pointcloud.cpp:47-53 */
struct PointHt { uint32_t idx; uint16_t height; };
typedef std::vector<PointHt> PointsHt;

/* ------------------------------------------------------------------------- */
/* configuration.h:27-52, pointcloud.cpp:60-106 (Projection2D, ProcessingConfiguration) */
struct Derived
{
  int W, H;
  double xMin, xMax, yMin, yMax, zMin, zMax;
  double recip;
  uint16_t minHeight;
  int minImgYExtent;
  double xToImage, yToImage, xToWorld, yToWorld;
  size_t nBins;
  double xyRatio() const { return xToImage / yToImage; }      /* pointcloud.cpp:93-96 */
  P2i worldToImage(const P3 &p) const                            /* pointcloud.cpp:79-83 */
  {
    return { static_cast<int>((p.x - xMin) * xToImage),
             static_cast<int>((yMax - p.y) * yToImage) };
  }
  P2 imageToWorld(const P2 &p) const                             /* pointcloud.cpp:84-88 */
  {
    return { xMin + p.x * xToWorld, yMax - p.y * yToWorld };
  }
};

Derived derive(const ssdo_config &c)
{
  Derived d;
  d.W = c.width; d.H = c.height;
  d.xMin = c.x_min; d.xMax = c.x_max; d.yMin = c.y_min; d.yMax = c.y_max; d.zMin = c.z_min; d.zMax = c.z_max;
  d.recip = 1.0 / c.height_interval;                                                     /* :101 */
  d.minHeight = static_cast<uint16_t>((c.min_height_above_ground - c.z_min) * d.recip);  /* :102 */
  d.minImgYExtent = static_cast<int>(c.min_step_depth * c.height / (c.y_max - c.y_min)); /* :103 */
  d.xToImage = c.width / (c.x_max - c.x_min);                                            /* :73 */
  d.yToImage = c.height / (c.y_max - c.y_min);                                           /* :74 */
  d.xToWorld = 1 / d.xToImage;
  d.yToWorld = 1 / d.yToImage;
  d.nBins = static_cast<size_t>((c.z_max - c.z_min) * d.recip) + 1;                      /* :189,196 */
  return d;
}

/* ------------------------------------------------------------------------- */
/* Boost.QVM pieces used by transformation.cpp (library absent from the image;
 * restated from its published generated operations: textbook formulas,
 * row sums left to right, normalized() = a * (1/sqrt(dot(a,a)))). */
P3 sub3(const P3 &a, const P3 &b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
P3 cross3(const P3 &a, const P3 &b)
{
  return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}
double dot3(const P3 &a, const P3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
P3 normalized3(const P3 &a)
{
  const double m2 = a.x * a.x + a.y * a.y + a.z * a.z;
  const double rm = 1.0 / std::sqrt(m2);
  return { a.x * rm, a.y * rm, a.z * rm };
}
P2 normalized2(const P2 &a)
{
  const double m2 = a.x * a.x + a.y * a.y;
  const double rm = 1.0 / std::sqrt(m2);
  return { a.x * rm, a.y * rm };
}

/* transformation.h:59-64 with a float source point: a*x + b, products double*float */
inline P3 cameraToWorld(const ssdo_calibration &k, float x, float y, float z)
{
  const double *a = k.a;
  P3 r;
  r.x = a[0] * x + a[1] * y + a[2] * z;
  r.y = a[3] * x + a[4] * y + a[5] * z;
  r.z = a[6] * x + a[7] * y + a[8] * z;
  r.x = r.x + k.b[0];
  r.y = r.y + k.b[1];
  r.z = r.z + k.b[2];
  return r;
}
inline P3 cameraToWorldD(const ssdo_calibration &k, const P3 &p)
{
  const double *a = k.a;
  P3 r;
  r.x = a[0] * p.x + a[1] * p.y + a[2] * p.z;
  r.y = a[3] * p.x + a[4] * p.y + a[5] * p.z;
  r.z = a[6] * p.x + a[7] * p.y + a[8] * p.z;
  r.x = r.x + k.b[0];
  r.y = r.y + k.b[1];
  r.z = r.z + k.b[2];
  return r;
}

/* transformation.cpp:190-194, transformation.h:59-64 (Dim 2) */
inline P3 toExternalWorld(const ssdo_calibration &k, const P3 &p)
{
  P3 r;
  r.x = k.r2[0] * p.x + k.r2[1] * p.y;
  r.y = k.r2[2] * p.x + k.r2[3] * p.y;
  r.x = r.x + k.t2[0];
  r.y = r.y + k.t2[1];
  r.z = k.world_z + p.z;
  return r;
}

bool almostEqual(double x, double y, int ulp)   /* transformation.cpp:33-45 */
{
  const double a = std::fabs(x - y);
  return a <= 2.220446049250313e-16 * std::fabs(x + y) * ulp || a < 2.2250738585072014e-308;
}

/* ------------------------------------------------------------------------- */
/* types.h:117-163 LineCoordinates; segmentation.cpp:321-407 Line<T> */
template<typename T>
struct LineT
{
  T a, b, c;
  T det(const LineT &o) const { return a * o.b - o.a * b; }
  T detx(const LineT &o) const { return b * o.c - o.b * c; }
  T dety(const LineT &o) const { return o.a * c - a * o.c; }
};
typedef LineT<int> Linei;
typedef LineT<double> Lined;

template<typename T, typename P>
LineT<T> lineThrough(const P &p, const P &q)
{
  const T x1 = p.x, y1 = p.y, x2 = q.x, y2 = q.y;
  return { y2 - y1, x1 - x2, x2 * y1 - x1 * y2 };
}
Lined toDouble(const Linei &l) { return { double(l.a), double(l.b), double(l.c) }; }

const double kTan60 = 1.7320508075688772935274463415058723669428; /* std::numbers::sqrt3, :314-315 */

/* segmentation.cpp:344-362 */
std::optional<P2> intersection(const Lined &l, const Lined &o)
{
  const double numerator = l.det(o);
  const double denominator = l.a * o.a + l.b * o.b;
  if(std::abs(numerator) > std::abs(denominator) * kTan60)
  {
    const double det = numerator;
    return P2{ l.detx(o) / det, l.dety(o) / det };
  }
  return std::nullopt;
}
Lined normalizedLine(const Lined &l)                 /* :383-387 */
{
  const double h = std::hypot(l.a, l.b);
  return { l.a / h, l.b / h, l.c / h };
}
Lined normalizedLine(const Linei &l)
{
  const double h = std::hypot(l.a, l.b);              /* int arguments promote to double */
  return { l.a / h, l.b / h, l.c / h };
}
template<typename A, typename B>
Lined angleBisector(const A &l, const B &o)          /* :388-396 */
{
  const Lined n = normalizedLine(l);
  const Lined m = normalizedLine(o);
  return { n.a + m.a, n.b + m.b, n.c + m.c };
}

/* segmentation.cpp:490-519 */
struct FlatLine
{
  double m, n;
  explicit FlatLine(const Linei &l) : m(double(-l.a) / l.b), n(double(-l.c) / l.b) {}
  P2 calcPoint(double x) const { return { x, x * m + n }; }
};

/* segmentation.cpp:409-449 ApproximationLine */
double residualOf(const Pts2i &pts, size_t pi, size_t qi, Linei &line)
{
  line = lineThrough<int>(pts[pi], pts[qi]);
  if(pts.size() <= 2)
    return 0;
  std::vector<int> dists;
  dists.reserve(pts.size() - 2);
  for(size_t i = 0; i < pts.size(); i++)
  {
    if(i == pi || i == qi)
      continue;
    dists.push_back(std::abs(pts[i].x * line.a + pts[i].y * line.b + line.c));
  }
  int sum = 0;
  const size_t n = dists.size() > 4 ? (dists.size() - 1) / 2 : 1;
  for(size_t i = n; i > 0; i--)
  {
    const auto mn = std::min_element(dists.begin(), dists.end());
    sum += *mn;
    dists.erase(mn);
  }
  return sum / (n * std::hypot(line.a, line.b));
}

/* segmentation.cpp:451-487 BestLine */
Linei bestLine(const Pts2i &pts)
{
  Linei best{ 0, 0, 0 };
  double bestRes = 0;
  bool first = true;
  for(size_t p = 0; p + 1 < pts.size(); p++)
    for(size_t q = p + 1; q < pts.size(); q++)
    {
      Linei l;
      const double r = residualOf(pts, p, q, l);
      if(first || r < bestRes)    /* min_element keeps the first of equal minima */
      {
        best = l;
        bestRes = r;
        first = false;
      }
    }
  return best;
}

/* ------------------------------------------------------------------------- */
/* image.h:30-84: a W x H byte image, zero-initialised */
struct Img
{
  int W, H;
  std::vector<uint8_t> px;
  Img(int w, int h) : W(w), H(h), px(size_t(w) * h, 0) {}
  uint8_t at(int x, int y) const { return px[size_t(y) * W + x]; }
};

/* cv::morphologyEx(img, img, MORPH_CLOSE, Mat()) (segmentation.cpp:888,928): OpenCV is
 * absent; restated from its documented semantics — 3x3 rectangle, anchor at the centre,
 * one iteration, BORDER_CONSTANT with morphologyDefaultBorderValue(), i.e. pixels outside
 * the image never win: dilation = max over the in-image neighbours, erosion = min over them. */
void close3x3(uint8_t *img, int W, int H)
{
  std::vector<uint8_t> tmp(size_t(W) * H);
  for(int y = 0; y < H; y++)
    for(int x = 0; x < W; x++)
    {
      uint8_t m = 0;
      for(int dy = -1; dy <= 1; dy++)
        for(int dx = -1; dx <= 1; dx++)
        {
          const int xx = x + dx, yy = y + dy;
          if(xx >= 0 && xx < W && yy >= 0 && yy < H)
            m = std::max(m, img[size_t(yy) * W + xx]);
        }
      tmp[size_t(y) * W + x] = m;
    }
  for(int y = 0; y < H; y++)
    for(int x = 0; x < W; x++)
    {
      uint8_t m = 255;
      for(int dy = -1; dy <= 1; dy++)
        for(int dx = -1; dx <= 1; dx++)
        {
          const int xx = x + dx, yy = y + dy;
          if(xx >= 0 && xx < W && yy >= 0 && yy < H)
            m = std::min(m, tmp[size_t(yy) * W + xx]);
        }
      img[size_t(y) * W + x] = m;
    }
}

/* fast path used by the timing entry: identical result, separable passes */
void close3x3Fast(uint8_t *img, int W, int H)
{
  std::vector<uint8_t> t1(size_t(W) * H), t2(size_t(W) * H);
  for(int y = 0; y < H; y++)
  {
    const uint8_t *r = img + size_t(y) * W;
    uint8_t *o = t1.data() + size_t(y) * W;
    for(int x = 0; x < W; x++)
    {
      uint8_t m = r[x];
      if(x > 0) m |= r[x - 1];
      if(x + 1 < W) m |= r[x + 1];
      o[x] = m;
    }
  }
  for(int y = 0; y < H; y++)
  {
    uint8_t *o = t2.data() + size_t(y) * W;
    const uint8_t *c = t1.data() + size_t(y) * W;
    const uint8_t *u = y > 0 ? c - W : c;
    const uint8_t *d = y + 1 < H ? c + W : c;
    for(int x = 0; x < W; x++)
      o[x] = c[x] | u[x] | d[x];
  }
  for(int y = 0; y < H; y++)
  {
    const uint8_t *r = t2.data() + size_t(y) * W;
    uint8_t *o = t1.data() + size_t(y) * W;
    for(int x = 0; x < W; x++)
    {
      uint8_t m = r[x];
      if(x > 0) m &= r[x - 1];
      if(x + 1 < W) m &= r[x + 1];
      o[x] = m;
    }
  }
  for(int y = 0; y < H; y++)
  {
    uint8_t *o = img + size_t(y) * W;
    const uint8_t *c = t1.data() + size_t(y) * W;
    const uint8_t *u = y > 0 ? c - W : c;
    const uint8_t *d = y + 1 < H ? c + W : c;
    for(int x = 0; x < W; x++)
      o[x] = c[x] & u[x] & d[x];
  }
}

/* ------------------------------------------------------------------------- */
/* segmentation.cpp:42-157 Scanner */
struct Scan { int x, yFirst, ySecond; };
typedef std::vector<Scan> Scans;

bool probeVertical(const Img &im, int x, int &yFirst, int &ySecond)   /* :88-111 */
{
  for(int yf = 0; yf < im.H; yf++)
    if(im.at(x, yf))
    {
      yFirst = yf;
      for(int ys = im.H - 1; ys >= yf; ys--)
        if(im.at(x, ys))
        {
          ySecond = ys;
          return true;
        }
      return false;
    }
  return false;
}

Scans scanColumns(const Img &im, int minImgYExtent, int xStart, int xStep)  /* :59-85 */
{
  Scans scans;
  int x = xStart, yFirst, ySecond;
  while(probeVertical(im, x, yFirst, ySecond))
  {
    if(ySecond - yFirst < minImgYExtent)
      break;
    scans.push_back({ x, yFirst, ySecond });
    x += xStep;
    if(x < 0 || x >= im.W)
      break;
  }
  return scans;
}

struct EdgePoints { Pts2i front, back; };
void pushScan(EdgePoints &e, const Scan &s)   /* :122-126 */
{
  e.front.push_back({ s.x, s.ySecond });
  e.back.push_back({ s.x, s.yFirst });
}

/* segmentation.cpp:129-156 */
void obtainLinePoints(const Scans &scansLeft, const Scans &scansRight, EdgePoints &left, EdgePoints &right)
{
  const size_t scansTotal = scansLeft.size() + scansRight.size();
  const size_t half = scansTotal / 2 + 1;
  size_t indLeft = 0, indRight = 0;
  if(scansLeft.size() >= half)
  {
    indLeft = scansLeft.size() - half;
    for(int i = int(indLeft); i >= 0; i--)
      pushScan(right, scansLeft[i]);
  }
  else
  {
    if(scansRight.size() > half)
      indRight = scansRight.size() - half;
    for(int i = int(indRight); i >= 0; i--)
      pushScan(left, scansRight[i]);
  }
  for( ; indRight < scansRight.size(); indRight++)
    pushScan(right, scansRight[indRight]);
  for( ; indLeft < scansLeft.size(); indLeft++)
    pushScan(left, scansLeft[indLeft]);
}

/* segmentation.cpp:159-241 BottomScanner */
bool probeBottomUp(const Img &im, int x, int &yEdge)
{
  const int yStop = im.H / 2;
  for(int y = im.H - 1; y > yStop; y--)
    if(im.at(x, y))
    {
      yEdge = y;
      return true;
    }
  return false;
}

Pts2i bottomScan(const Img &im, int xStart, int xStep)
{
  Pts2i points;
  int x = xStart, y;
  do
  {
    if(probeBottomUp(im, x, y))
    {
      points.push_back({ x, y });
      break;
    }
    x += xStep;
  }
  while(x < im.W);

  if(points.empty())
  {
    x = xStart - xStep;
    do
    {
      if(probeBottomUp(im, x, y))
      {
        points.push_back({ x, y });
        break;
      }
      x -= xStep;
    }
    while(x >= 0);
    if(points.empty())
      return points;
  }

  xStart = x;
  x += xStep;
  while(x < im.W && probeBottomUp(im, x, y))
  {
    points.push_back({ x, y });
    x += xStep;
  }
  x = xStart - xStep;
  while(x >= 0 && probeBottomUp(im, x, y))
  {
    points.push_back({ x, y });
    x -= xStep;
  }
  return points;
}

/* segmentation.cpp:243-312 VerticalEdgePointsDetector */
Pts2i detectLeftEdgePts(const Img &im, int x0, int y0, int length, int yEnd, int yStep)
{
  Pts2i pts;
  for(int y = y0; y >= yEnd; y -= yStep)
    for(int i = 0; i < length; i++)
      if(im.at(x0 + i, y))
      {
        pts.push_back({ x0 + i, y });
        break;
      }
  return pts;
}
Pts2i detectRightEdgePts(const Img &im, int x0, int y0, int length, int yEnd, int yStep)
{
  Pts2i pts;
  for(int y = y0; y >= yEnd; y -= yStep)
    for(int i = 0; i < length; i++)
      if(im.at(x0 - i, y))
      {
        pts.push_back({ x0 - i, y });
        break;
      }
  return pts;
}

/* segmentation.cpp:521-552 BoundaryPoints */
struct Bounds { P2 inner{ -1, -1 }, outer{ -1, -1 }; };
Bounds boundaryPoints(const Pts2i &pts, const FlatLine &fl, int &status)
{
  Bounds b;
  const int distanceLimit = 10;
  auto calcBound = [&](const P2i &pi, P2 &bound)
  {
    const P2 pd = fl.calcPoint(pi.x);
    if(std::abs(pd.y - pi.y) < distanceLimit)
    {
      bound = pd;
      return true;
    }
    return false;
  };
  for(size_t i = 0; i < pts.size(); i++)
    if(calcBound(pts[i], b.inner))
      break;
  for(size_t i = pts.size(); i-- > 0; )
    if(calcBound(pts[i], b.outer))
      break;
  if((b.inner.x == -1 && b.inner.y == -1) || (b.outer.x == -1 && b.outer.y == -1))
    status |= SSDO_ST_ASSERT;  /* :549-550 */
  return b;
}

struct HEdge { Pts2i pts; Linei line; Bounds bounds; };

/* segmentation.cpp:755-787 */
bool isConvex(const std::array<P2, 4> &q)
{
  const P2 v[4] = { { q[1].x - q[0].x, q[1].y - q[0].y },
                    { q[3].x - q[1].x, q[3].y - q[1].y },
                    { q[2].x - q[3].x, q[2].y - q[3].y },
                    { q[0].x - q[2].x, q[0].y - q[2].y } };
  auto pos = [](const P2 &a, const P2 &b) { return a.x * b.y - b.x * a.y > 0; };
  const bool positive = pos(v[0], v[1]);
  return positive == pos(v[1], v[2]) && positive == pos(v[2], v[3]) && positive == pos(v[3], v[0]);
}

struct Outline { std::array<P2, 4> quad{}; bool valid = false; };

/* segmentation.cpp:919-971 detectOutline (+ :602-751) ; fills the intermediates of dbg if given */
Outline detectOutline(Img &im, int minImgYExtent, double xyRatio, bool fastClose, ssdo_plateau *dbg, int &status)
{
  Outline outline;
  if(fastClose) close3x3Fast(im.px.data(), im.W, im.H); else close3x3(im.px.data(), im.W, im.H);

  const int xStep = 25;
  const int xCenter = im.W / 2;
  const Scans scansRight = scanColumns(im, minImgYExtent, xCenter, xStep);
  if(scansRight.empty())
    return outline;
  const Scans scansLeft = scanColumns(im, minImgYExtent, xCenter - xStep, -xStep);
  if(dbg)
  {
    dbg->n_scans_right = int(scansRight.size());
    dbg->n_scans_left = int(scansLeft.size());
    for(size_t i = 0; i < scansRight.size() && i < SSDO_MAX_SCANS; i++)
    { dbg->scans_right[i][0] = scansRight[i].x; dbg->scans_right[i][1] = scansRight[i].yFirst; dbg->scans_right[i][2] = scansRight[i].ySecond; }
    for(size_t i = 0; i < scansLeft.size() && i < SSDO_MAX_SCANS; i++)
    { dbg->scans_left[i][0] = scansLeft[i].x; dbg->scans_left[i][1] = scansLeft[i].yFirst; dbg->scans_left[i][2] = scansLeft[i].ySecond; }
  }
  if(scansLeft.size() + scansRight.size() < 3)
    return outline;

  /* HorizontalEdges :555-600 */
  EdgePoints left, right;
  obtainLinePoints(scansLeft, scansRight, left, right);
  HEdge e[4];
  e[SSDO_FRONT_LEFT].pts = left.front;
  e[SSDO_FRONT_RIGHT].pts = right.front;
  e[SSDO_BACK_LEFT].pts = left.back;
  e[SSDO_BACK_RIGHT].pts = right.back;
  for(int i = 0; i < 4; i++)
  {
    e[i].line = bestLine(e[i].pts);
    e[i].bounds = boundaryPoints(e[i].pts, FlatLine(e[i].line), status);
  }
  if(dbg)
  {
    dbg->outline_found = 1;
    for(int i = 0; i < 4; i++)
    {
      dbg->n_edge_pts[i] = int(e[i].pts.size());
      dbg->line[i][0] = e[i].line.a; dbg->line[i][1] = e[i].line.b; dbg->line[i][2] = e[i].line.c;
      dbg->bounds[i][0][0] = e[i].bounds.inner.x; dbg->bounds[i][0][1] = e[i].bounds.inner.y;
      dbg->bounds[i][1][0] = e[i].bounds.outer.x; dbg->bounds[i][1][1] = e[i].bounds.outer.y;
    }
  }

  /* VerticalEdgesDetector::calcBaseLine :672-679 */
  const Linei flRev{ -e[SSDO_FRONT_LEFT].line.a, -e[SSDO_FRONT_LEFT].line.b, -e[SSDO_FRONT_LEFT].line.c };
  const Linei blRev{ -e[SSDO_BACK_LEFT].line.a, -e[SSDO_BACK_LEFT].line.b, -e[SSDO_BACK_LEFT].line.c };
  const Lined front = angleBisector(flRev, e[SSDO_FRONT_RIGHT].line);
  const Lined back = angleBisector(blRev, e[SSDO_BACK_RIGHT].line);
  const Lined center = angleBisector(front, back);
  const double corr = xyRatio * xyRatio;
  const Lined sc{ center.a * corr, center.b, center.c };
  const P2i p0 = e[SSDO_FRONT_LEFT].pts.front();
  const Lined baseLine{ -sc.b, sc.a, sc.b * p0.x - sc.a * p0.y };
  if(dbg) { dbg->base_line[0] = baseLine.a; dbg->base_line[1] = baseLine.b; dbg->base_line[2] = baseLine.c; }

  /* detectEdge :681-706, findBestPoint :708-728 */
  const int xExtension = xStep, yStep = 10;
  std::optional<Lined> vline[2];
  for(int side = 0; side < 2; side++)
  {
    const P2 frontPt = e[side == 0 ? SSDO_FRONT_LEFT : SSDO_FRONT_RIGHT].bounds.outer;
    const P2 backPt = e[side == 0 ? SSDO_BACK_LEFT : SSDO_BACK_RIGHT].bounds.outer;
    int lft = std::min(frontPt.x, backPt.x) - xExtension;
    int rgt = std::max(frontPt.x, backPt.x) + xExtension;
    int yStart = frontPt.y - yStep;
    int yEnd = backPt.y + yStep;
    if(lft < 0) lft = 0;
    if(rgt >= im.W) rgt = im.W - 1;
    if(yStart >= im.H) yStart = im.H - 1;
    if(yEnd < 0) yEnd = 0;
    if(yStart < yEnd)
      continue;
    if(rgt - lft <= 0) { status |= SSDO_ST_ASSERT; continue; }   /* :254 */
    const Pts2i pts = side == 0 ? detectLeftEdgePts(im, lft, yStart, rgt - lft, yEnd, yStep)
                                : detectRightEdgePts(im, rgt, yStart, rgt - lft, yEnd, yStep);
    if(dbg)
    {
      dbg->n_vpts[side] = int(pts.size());
      for(size_t i = 0; i < pts.size() && i < SSDO_MAX_EDGE_PTS; i++)
      { dbg->vpts[side][i][0] = pts[i].x; dbg->vpts[side][i][1] = pts[i].y; }
    }
    if(pts.empty())
      continue;
    struct PD { size_t i; double dist; };
    std::vector<PD> dists;
    dists.reserve(pts.size());
    for(size_t i = 0; i < pts.size(); i++)
      dists.push_back({ i, std::abs(pts[i].x * baseLine.a + pts[i].y * baseLine.b + baseLine.c) });
    std::sort(dists.begin(), dists.end(), [](const PD &a, const PD &b) { return a.dist < b.dist; });
    const P2i bp = pts[dists[2 * dists.size() / 3].i];
    vline[side] = Lined{ baseLine.a, baseLine.b, -baseLine.a * bp.x - baseLine.b * bp.y };  /* :368-371 */
    if(dbg)
    {
      dbg->vedge_found[side] = 1;
      dbg->best_pt[side][0] = bp.x; dbg->best_pt[side][1] = bp.y;
      dbg->vline[side][0] = vline[side]->a; dbg->vline[side][1] = vline[side]->b; dbg->vline[side][2] = vline[side]->c;
    }
  }

  /* Corners :731-751 */
  std::optional<P2> c[4];
  if(vline[0])
  {
    c[SSDO_FRONT_LEFT] = intersection(*vline[0], toDouble(e[SSDO_FRONT_LEFT].line));
    c[SSDO_BACK_LEFT] = intersection(*vline[0], toDouble(e[SSDO_BACK_LEFT].line));
  }
  if(vline[1])
  {
    c[SSDO_FRONT_RIGHT] = intersection(*vline[1], toDouble(e[SSDO_FRONT_RIGHT].line));
    c[SSDO_BACK_RIGHT] = intersection(*vline[1], toDouble(e[SSDO_BACK_RIGHT].line));
  }
  for(int i = 0; i < 4; i++)
  {
    outline.quad[i] = c[i] ? *c[i] : e[i].bounds.outer;   /* :947-953 */
    if(dbg) dbg->corner_found[i] = c[i] ? 1 : 0;
  }
  outline.valid = isConvex(outline.quad);
  return outline;
}

struct FrontEdge { P2 left, right; bool valid = false; };

/* segmentation.cpp:879-917 */
FrontEdge detectFrontEdge(Img &im, bool fastClose, ssdo_result *dbg)
{
  FrontEdge fe;
  if(fastClose) close3x3Fast(im.px.data(), im.W, im.H); else close3x3(im.px.data(), im.W, im.H);
  const Pts2i pts = bottomScan(im, im.W / 2, 50);
  if(dbg)
  {
    dbg->ground_n_pts = int(pts.size());
    for(size_t i = 0; i < pts.size() && i < SSDO_MAX_SCANS; i++)
    { dbg->ground_pts[i][0] = pts[i].x; dbg->ground_pts[i][1] = pts[i].y; }
  }
  if(pts.size() >= 2)
  {
    const Linei l = bestLine(pts);
    const FlatLine edge(l);
    int xl = pts[0].x, xr = pts[0].x;
    for(const P2i &p : pts) { xl = std::min(xl, p.x); xr = std::max(xr, p.x); }
    fe.left = edge.calcPoint(xl);
    fe.right = edge.calcPoint(xr);
    fe.valid = true;
    if(dbg)
    {
      dbg->ground_line[0] = l.a; dbg->ground_line[1] = l.b; dbg->ground_line[2] = l.c;
      dbg->ground_front_img[0] = fe.left.x; dbg->ground_front_img[1] = fe.left.y;
      dbg->ground_front_img[2] = fe.right.x; dbg->ground_front_img[3] = fe.right.y;
    }
  }
  return fe;
}

/* ------------------------------------------------------------------------- */
/* quadrilateralTest.{h,cpp}: strict point-in-convex-quadrilateral test */
struct Sector
{
  double lo, up;
  Sector(double a, double b) : lo(a), up(a) { expand(b); }        /* :28-40 */
  void expand(double c) { if(lo > c) lo = c; else if(up < c) up = c; }
  bool overlaps(const Sector &o) const { return lo < o.up && up > o.lo; }
  bool isAbove(const Sector &o) const { return (lo + up) / 2 < o.lo; }
  bool isBelow(const Sector &o) const { return (lo + up) / 2 > o.up; }
  bool isWithin(double c) const { return lo < c && c < up; }
};
enum RelPos { NOWHERE, X_ABOVE, X_BELOW, Y_ABOVE, Y_BELOW, RELPOS_MAX };
struct BBox
{
  Sector x, y;
  BBox(double lx, double ux, double ly, double uy) : x(lx, ux), y(ly, uy) {}
  BBox(const P2 &p, const P2 &q) : x(p.x, q.x), y(p.y, q.y) {}
  void expand(const P2 &p) { x.expand(p.x); y.expand(p.y); }
  bool overlaps(const BBox &o) const { return x.overlaps(o.x) && y.overlaps(o.y); }
  RelPos relPos(const BBox &o) const                                /* :93-110 */
  {
    if(y.overlaps(o.y))
    {
      if(x.isAbove(o.x)) return X_ABOVE;
      if(x.isBelow(o.x)) return X_BELOW;
    }
    if(x.overlaps(o.x))
    {
      if(y.isAbove(o.y)) return Y_ABOVE;
      if(y.isBelow(o.y)) return Y_BELOW;
    }
    return NOWHERE;
  }
  bool isWithin(const P2 &p) const { return x.isWithin(p.x) && y.isWithin(p.y); }
};

struct Segment       /* :120-274 the four Flat/Steep Positive/Negative segment kinds */
{
  BBox box;
  bool steep;
  double k, c;
  bool leftIfPositive;
  Segment(const P2 &p, const P2 &q) : box(p, q)
  {
    const double dx = q.x - p.x, dy = q.y - p.y;
    const Lined l = lineThrough<double>(p, q);
    if(std::abs(dx) < std::abs(dy))
    {
      steep = true;
      k = l.b / l.a;
      c = l.c / l.a;
      leftIfPositive = !(dy > 0);
    }
    else
    {
      steep = false;
      k = l.a / l.b;
      c = l.c / l.b;
      leftIfPositive = dx > 0;
    }
  }
  bool isPositive(const P2 &p) const
  {
    return steep ? p.x + p.y * k + c > 0 : p.x * k + p.y + c > 0;
  }
  bool isLeft(const P2 &p) const { return leftIfPositive ? isPositive(p) : !isPositive(p); }
};

enum QuadErr { QE_OK = 0, QE_NOT_CONVEX = -1, QE_NO_Y = -2, QE_NO_X = -3, QE_3SEG = -4, QE_2EMPTY = -5, QE_2DOUBLE = -6 };

struct QuadTest     /* :275-451 */
{
  struct Cell { double upperX; std::vector<int> segs; bool neighbor[RELPOS_MAX] = { false, false, false, false, false }; };
  struct Row { double upperY; std::vector<Cell> cells; };
  BBox total;
  std::vector<Segment> segs;
  bool insideIsLeft = false;
  std::vector<Row> rows;
  int err = QE_OK;

  explicit QuadTest(const std::array<P2, 4> &q) : total(q[0], q[1])
  {
    total.expand(q[2]);
    total.expand(q[3]);
    segs = { Segment(q[0], q[1]), Segment(q[1], q[3]), Segment(q[3], q[2]), Segment(q[2], q[0]) };
    insideIsLeft = segs[0].isLeft(q[3]);
    if(insideIsLeft != segs[1].isLeft(q[2]) || insideIsLeft != segs[2].isLeft(q[0]) || insideIsLeft != segs[3].isLeft(q[1]))
    {
      err = QE_NOT_CONVEX;
      return;
    }
    std::array<double, 4> xs{ q[0].x, q[1].x, q[2].x, q[3].x };
    std::array<double, 4> ys{ q[0].y, q[1].y, q[2].y, q[3].y };
    std::sort(xs.begin(), xs.end());
    std::sort(ys.begin(), ys.end());
    double lowerY = ys[0];
    for(size_t yi = 1; yi < 4; yi++)
    {
      if(lowerY < ys[yi])
      {
        rows.push_back({ ys[yi], {} });
        Row &row = rows.back();
        double lowerX = xs[0];
        for(size_t xi = 1; xi < 4; xi++)
        {
          if(lowerX < xs[xi])
          {
            row.cells.push_back({ xs[xi], {} });
            Cell &cell = row.cells.back();
            const BBox cellbox(lowerX, cell.upperX, lowerY, row.upperY);
            for(int si = 0; si < 4; si++)
            {
              if(cellbox.overlaps(segs[si].box))
                cell.segs.push_back(si);
              if(cell.segs.empty())
                cell.neighbor[cellbox.relPos(segs[si].box)] = true;
            }
            lowerX = cell.upperX;
          }
        }
        lowerY = row.upperY;
      }
    }
    if(rows.empty()) { err = QE_NO_Y; return; }
    for(const Row &row : rows)
    {
      if(row.cells.empty()) { err = QE_NO_X; return; }
      for(const Cell &cell : row.cells)
        if(cell.segs.size() > 2) { err = QE_3SEG; return; }
    }
    for(Row &row : rows)
    {
      for(size_t ci = 0; ci + 1 < row.cells.size(); )
      {
        const std::vector<int> &cur = row.cells[ci].segs;
        const std::vector<int> &nxt = row.cells[ci + 1].segs;
        if(cur.empty() && nxt.empty()) { err = QE_2EMPTY; return; }
        if(cur.size() > 1 && nxt.size() > 1) { err = QE_2DOUBLE; return; }
        if(cur == nxt)
          row.cells.erase(row.cells.begin() + ci);
        else
          ++ci;
      }
    }
  }

  bool cellTest(const Cell &cell, const P2 &p) const
  {
    switch(cell.segs.size())
    {
      case 0: return cell.neighbor[X_ABOVE] && cell.neighbor[X_BELOW] && cell.neighbor[Y_ABOVE] && cell.neighbor[Y_BELOW];
      case 1: return segs[cell.segs[0]].isLeft(p) == insideIsLeft;
      default: return segs[cell.segs[0]].isLeft(p) == insideIsLeft && segs[cell.segs[1]].isLeft(p) == insideIsLeft;
    }
  }
  bool rowTest(const Row &row, const P2 &p) const
  {
    const size_t n = row.cells.size();
    if(n == 1) return cellTest(row.cells[0], p);
    if(p.x < row.cells[0].upperX) return cellTest(row.cells[0], p);
    if(n == 2) return cellTest(row.cells[1], p);
    if(p.x < row.cells[1].upperX) return cellTest(row.cells[1], p);
    return cellTest(row.cells[2], p);
  }
  bool isPointWithin(const P2 &p) const    /* :445-451 */
  {
    if(!total.isWithin(p))
      return false;
    const size_t n = rows.size();
    if(n == 1) return rowTest(rows[0], p);
    if(p.y < rows[0].upperY) return rowTest(rows[0], p);
    if(n == 2) return rowTest(rows[1], p);
    if(p.y < rows[1].upperY) return rowTest(rows[1], p);
    return rowTest(rows[2], p);
  }
};

/* ------------------------------------------------------------------------- */
/* stairs.cpp:34-70 */
std::string serializeSteps(int n, const double *steps /* n x 9 */)
{
  std::ostringstream os;
  os << "[\"stairs\",[\"stairSteps\"," << size_t(n) << ']';
  auto point = [&os](const double *p)
  {
    os << std::fixed << std::setprecision(3) << '[' << p[0] << ',' << p[1] << ']';
  };
  auto step = [&os, &point](const double *s)
  {
    os << std::fixed << std::setprecision(3) << "[[\"height\"," << s[0] << "],[\"quadrilateral\",";
    point(s + 1); os << ',';
    point(s + 3); os << ',';
    point(s + 5); os << ',';
    point(s + 7); os << "]]";
  };
  if(n > 0)
  {
    os << ",[";
    for(int i = 0; i < n - 1; i++)
    {
      step(steps + 9 * i);
      os << ',';
    }
    step(steps + 9 * (n - 1));
    os << ']';
  }
  os << ']';
  return os.str();
}

/* ------------------------------------------------------------------------- */
struct Plateau     /* pointcloud.cpp:259-265 */
{
  uint16_t height;
  PointsHt pts;
  std::array<P2, 4> quadWorld{};
  bool valid = false;
  int lo = 0, hi = 0;
};

/* pointcloud.cpp:337-343 */
void split(const PointsHt &pts, uint16_t threshold, PointsHt &lower, PointsHt &upper)
{
  for(const PointHt &p : pts)
    if(threshold < p.height) upper.push_back(p); else lower.push_back(p);
}

struct Frame
{
  const Derived &d;
  const ssdo_calibration &cal;
  std::vector<P3> points;     /* in-range world points */
  bool fast;
  int status = 0;
  int nOob = 0;

  /* pointcloud.cpp:458-471 */
  Img project(const PointsHt &pts)
  {
    Img im(d.W, d.H);
    for(const PointHt &ph : pts)
    {
      const P2i ip = d.worldToImage(points[ph.idx]);
      if(ip.x < 0 || ip.x >= d.W || ip.y < 0 || ip.y >= d.H)
      {
        /* quirk Q5: the reference writes outside the image here; the build drops the point */
        status |= SSDO_ST_OOB_PIXEL;
        nOob++;
        continue;
      }
      im.px[size_t(ip.y) * d.W + ip.x] = 0xff;
    }
    return im;
  }
  std::array<P2, 4> imgToWorld(const std::array<P2, 4> &q) const   /* :476-487 */
  {
    std::array<P2, 4> w;
    for(int i = 0; i < 4; i++) w[i] = d.imageToWorld(q[i]);
    return w;
  }
  /* :560-581 */
  int inQuad(const PointsHt &pts, const std::array<P2, 4> &quad, PointsHt &out)
  {
    const QuadTest qt(quad);
    if(qt.err != QE_OK)
      return qt.err;
    for(const PointHt &ph : pts)
    {
      const P3 &p = points[ph.idx];
      if(qt.isPointWithin({ p.x, p.y }))
        out.push_back(ph);
    }
    return 0;
  }
  double averageZ(const PointsHt &pts) const
  {
    double sum = 0;
    for(const PointHt &ph : pts)
      sum = sum + points[ph.idx].z;
    return sum / pts.size();
  }
};

/* pointcloud.cpp:489-512 */
std::array<P2, 4> calcGroundQuadrilateral(const std::array<P2, 4> &q, double yMin)
{
  auto calcDx = [yMin](const P2 &p, const P2 &q) { return (q.y - yMin) * (q.y - p.y) / (q.x - p.x); };
  std::array<P2, 4> gq;
  if(q[0].y < q[1].y)
  {
    gq[0] = { q[0].x, yMin };
    gq[1] = { q[1].x + calcDx(q[0], q[1]), yMin };
  }
  else
  {
    gq[0] = { q[0].x + calcDx(q[1], q[0]), yMin };
    gq[1] = { q[1].x, yMin };
  }
  gq[2] = q[0];
  gq[3] = q[1];
  return gq;
}

/* Pointcloud::process, pointcloud.cpp:608-626 */
int processFrame(const ssdo_config &cfg, const ssdo_calibration &cal, const float *xyz, bool fast,
                 ssdo_result *out, double *stepsExtOut, int *statusOut,
                 uint8_t *rawImages, uint8_t *closedImages, int maxImages,
                 uint8_t *groundRaw, uint8_t *groundClosed)
{
  const Derived d = derive(cfg);
  if(d.nBins > SSDO_MAX_BINS || d.nBins < 3)
    return -1;
  const size_t N = size_t(cfg.width) * cfg.height;
  Frame fr{ d, cal, {}, fast };

  /* PointsExtraction::extract :122-178 */
  size_t nNonZero = 0;
  fr.points.reserve(N);
  for(size_t i = 0; i < N; i++)
  {
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    if(!(z > 0))
      continue;
    nNonZero++;
    const P3 w = cameraToWorld(cal, x, y, z);
    if(w.x > d.xMin && w.x < d.xMax && w.y > d.yMin && w.y < d.yMax && w.z > d.zMin && w.z < d.zMax)
      fr.points.push_back(w);
  }
  PointsHt pointsHt;
  pointsHt.reserve(fr.points.size());
  for(uint32_t i = 0; i < fr.points.size(); i++)
    pointsHt.push_back({ i, static_cast<uint16_t>((fr.points[i].z - d.zMin) * d.recip) });

  /* HeightsHistogram :184-256 */
  std::vector<uint32_t> hist(d.nBins, 0);
  for(const PointHt &p : pointsHt)
    ++hist[p.height];

  std::vector<uint16_t> peaksRaw, peaks;
  {
    const size_t n = hist.size() - 1;
    bool ascending = false;
    for(uint16_t i = 0; i < n; i++)
    {
      const uint32_t curr = hist[i], succ = hist[i + 1];
      if(curr < succ) { ascending = true; continue; }
      if(curr > succ)
      {
        if(ascending) peaksRaw.push_back(i);
        ascending = false;
      }
    }
    for(uint16_t i : peaksRaw)
    {
      const uint32_t np = hist[i];
      if(np < 2000) continue;
      if((np * 2 - hist[i - 1] - hist[i + 1]) * 2 > np)
        peaks.push_back(i);
    }
  }

  if(out)
  {
    out->n_total = int(N);
    out->n_nonzero = int(nNonZero);
    out->n_inrange = int(fr.points.size());
    out->n_bins = int(d.nBins);
    out->min_height = d.minHeight;
    out->min_img_y_extent = d.minImgYExtent;
    out->x_to_image = d.xToImage;
    out->y_to_image = d.yToImage;
    out->xy_ratio = d.xyRatio();
    for(size_t i = 0; i < d.nBins; i++) out->hist[i] = hist[i];
    out->n_peaks_raw = int(peaksRaw.size());
    for(size_t i = 0; i < peaksRaw.size(); i++) out->peaks_raw[i] = peaksRaw[i];
    out->n_peaks = int(peaks.size());
    for(size_t i = 0; i < peaks.size(); i++) out->peaks[i] = peaks[i];
    out->ground_ind = -1;
    out->first_valid_ind = -1;
  }
  if(peaks.size() > SSDO_MAX_PLATEAUS)
    return -2;

  /* PlateausExtraction :280-343 */
  std::vector<Plateau> plateaus;
  plateaus.reserve(peaks.size());
  {
    PointsHt remainder;
    for(uint16_t height : peaks)
    {
      uint16_t hMin, hMax;
      const uint16_t pred = height - 1, succ = height + 1;
      if(hist[pred] > hist[succ]) { hMin = pred; hMax = height; }
      else { hMin = height; hMax = succ; }
      PointsHt upper;
      split(pointsHt, static_cast<uint16_t>(hMin - 1), remainder, upper);   /* quirk Q4: wraps for hMin = 0 */
      Plateau pl;
      pl.height = height;
      pl.lo = hMin; pl.hi = hMax;
      pointsHt.clear();
      split(upper, hMax, pl.pts, pointsHt);
      plateaus.push_back(std::move(pl));
    }
  }

  /* StairsDetector::detectStairSteps :399-456 */
  size_t maxGround = 0;
  int groundInd = -1, firstValidInd = -1;
  size_t i = 0;
  for( ; i < plateaus.size(); i++)
  {
    if(plateaus[i].height >= d.minHeight)
      break;
    if(maxGround < plateaus[i].pts.size())
    {
      maxGround = plateaus[i].pts.size();
      groundInd = int(i);
    }
  }
  const size_t firstStep = i;
  for( ; i < plateaus.size(); i++)
  {
    Plateau &pl = plateaus[i];
    Img im = fr.project(pl.pts);
    const size_t imgIdx = i - firstStep;
    if(rawImages && int(imgIdx) < maxImages)
      std::memcpy(rawImages + imgIdx * N, im.px.data(), N);
    ssdo_plateau *dbg = out ? &out->plateaus[i] : nullptr;
    const Outline ol = detectOutline(im, d.minImgYExtent, d.xyRatio(), fast, dbg, fr.status);
    if(closedImages && int(imgIdx) < maxImages)
      std::memcpy(closedImages + imgIdx * N, im.px.data(), N);
    pl.quadWorld = fr.imgToWorld(ol.quad);
    pl.valid = ol.valid;
    if(dbg)
      for(int k = 0; k < 4; k++)
      {
        dbg->quad_img[2 * k] = ol.quad[k].x; dbg->quad_img[2 * k + 1] = ol.quad[k].y;
      }
    if(pl.valid && firstValidInd < 0)
      firstValidInd = int(i);
  }

  std::vector<std::array<P3, 4>> stairSteps;
  bool threw = false;
  if(firstValidInd >= 0)
  {
    if(groundInd >= 0)
    {
      Plateau &ground = plateaus[groundInd];
      ground.quadWorld = calcGroundQuadrilateral(plateaus[firstValidInd].quadWorld, d.yMin);
      ground.valid = true;
      /* calcGround :528-547 */
      PointsHt inq;
      const int qe = fr.inQuad(ground.pts, ground.quadWorld, inq);
      if(qe != 0)
        threw = true;
      else
      {
        Img gim = fr.project(inq);
        if(groundRaw) std::memcpy(groundRaw, gim.px.data(), N);
        const FrontEdge fe = detectFrontEdge(gim, fast, out);
        if(groundClosed) std::memcpy(groundClosed, gim.px.data(), N);
        std::array<P3, 4> gq{};
        if(fe.valid)
        {
          const P2 fl = d.imageToWorld(fe.left), frp = d.imageToWorld(fe.right);
          const Lined frontLine = lineThrough<double>(fl, frp);
          auto isect = [](const Lined &l, const Lined &o)     /* pointcloud.cpp:520-525 */
          {
            const double dd = l.det(o);
            return P2{ l.detx(o) / dd, l.dety(o) / dd };
          };
          const P2 frontLeft = isect(frontLine, lineThrough<double>(ground.quadWorld[0], ground.quadWorld[2]));
          const P2 frontRight = isect(frontLine, lineThrough<double>(ground.quadWorld[1], ground.quadWorld[3]));
          const double az = fr.averageZ(inq);
          gq = { { { frontLeft.x, frontLeft.y, az }, { frontRight.x, frontRight.y, az },
                   { ground.quadWorld[2].x, ground.quadWorld[2].y, az }, { ground.quadWorld[3].x, ground.quadWorld[3].y, az } } };
          if(out) out->ground_mean_z = az;
        }
        if(out)
        {
          out->ground_front_valid = fe.valid ? 1 : 0;
          out->ground_n_in_quad = int(inq.size());
          for(int k = 0; k < 4; k++)
          {
            out->ground_quad_world[2 * k] = ground.quadWorld[k].x;
            out->ground_quad_world[2 * k + 1] = ground.quadWorld[k].y;
          }
        }
        stairSteps.push_back(gq);
      }
    }
    for(size_t k = size_t(firstValidInd); k < plateaus.size() && !threw; k++)
    {
      const Plateau &pl = plateaus[k];
      if(!pl.valid)
        continue;
      /* calcStairStep :549-558 */
      PointsHt inq;
      const int qe = fr.inQuad(pl.pts, pl.quadWorld, inq);
      if(qe != 0) { threw = true; break; }
      const double az = fr.averageZ(inq);
      stairSteps.push_back({ { { pl.quadWorld[0].x, pl.quadWorld[0].y, az }, { pl.quadWorld[1].x, pl.quadWorld[1].y, az },
                               { pl.quadWorld[2].x, pl.quadWorld[2].y, az }, { pl.quadWorld[3].x, pl.quadWorld[3].y, az } } });
      if(out)
      {
        out->plateaus[k].n_in_quad = int(inq.size());
        out->plateaus[k].mean_z = az;
      }
    }
  }
  if(threw)
  {
    fr.status |= SSDO_ST_THROW;    /* std::invalid_argument, uncaught in main: no line is printed */
    stairSteps.clear();
  }

  /* detectStairs tail :370-393 */
  std::vector<double> ext(9 * std::max<size_t>(1, stairSteps.size()));
  for(size_t s = 0; s < stairSteps.size(); s++)
  {
    P3 e[4];
    for(int k = 0; k < 4; k++) e[k] = toExternalWorld(cal, stairSteps[s][k]);
    ext[9 * s] = e[0].z;
    for(int k = 0; k < 4; k++) { ext[9 * s + 1 + 2 * k] = e[k].x; ext[9 * s + 2 + 2 * k] = e[k].y; }
  }

  if(out)
  {
    out->status = fr.status;
    out->n_oob = fr.nOob;
    out->n_plateaus = int(plateaus.size());
    out->ground_ind = groundInd;
    out->first_valid_ind = firstValidInd;
    for(size_t k = 0; k < plateaus.size(); k++)
    {
      ssdo_plateau &p = out->plateaus[k];
      p.peak_bin = plateaus[k].height;
      p.bin_lo = plateaus[k].lo; p.bin_hi = plateaus[k].hi;
      p.n_points = int(plateaus[k].pts.size());
      p.is_step = plateaus[k].height >= d.minHeight ? 1 : 0;
      p.valid = plateaus[k].valid ? 1 : 0;
      for(int c = 0; c < 4; c++)
      {
        p.quad_world[2 * c] = plateaus[k].quadWorld[c].x; p.quad_world[2 * c + 1] = plateaus[k].quadWorld[c].y;
      }
    }
    out->n_steps = int(stairSteps.size());
    for(size_t s = 0; s < stairSteps.size() && s < SSDO_MAX_STEPS; s++)
    {
      for(int k = 0; k < 4; k++)
      {
        out->steps_world[s][3 * k] = stairSteps[s][k].x;
        out->steps_world[s][3 * k + 1] = stairSteps[s][k].y;
        out->steps_world[s][3 * k + 2] = stairSteps[s][k].z;
      }
      for(int k = 0; k < 9; k++) out->steps_ext[s][k] = ext[9 * s + k];
    }
    const std::string line = threw ? std::string() : serializeSteps(int(stairSteps.size()), ext.data());
    std::snprintf(out->line, SSDO_LINE_CAP, "%s", line.c_str());
  }
  if(stepsExtOut)
    for(size_t s = 0; s < stairSteps.size() && s < SSDO_MAX_STEPS; s++)
      for(int k = 0; k < 9; k++) stepsExtOut[9 * s + k] = ext[9 * s + k];
  if(statusOut)
    *statusOut = fr.status;
  return int(stairSteps.size());
}

} // namespace

extern "C"
{

void ssdo_default_config(ssdo_config *cfg, int width, int height)
{
  /* configuration.h:27-52 */
  cfg->width = width; cfg->height = height;
  cfg->x_min = -0.6; cfg->x_max = 0.6;
  cfg->y_min = 0.1; cfg->y_max = 1.3;
  cfg->z_min = -0.1; cfg->z_max = 1.1;
  cfg->height_interval = 0.01;
  cfg->min_height_above_ground = 0.05;
  cfg->min_step_depth = 0.1;
}

/* transformation.cpp:196-215, 108-157, 65-106 */
int ssdo_calibration_from_points(const double world[9], const double cam[9], ssdo_calibration *out)
{
  int rc = 0;
  const P3 c0{ cam[0], cam[1], cam[2] }, c1{ cam[3], cam[4], cam[5] }, c2{ cam[6], cam[7], cam[8] };
  /* Transformation_<3>(triangleInPlane) */
  const P3 p = c0;
  const P3 u = sub3(c1, p), v = sub3(c2, p);
  const P3 n0 = normalized3(cross3(u, v));
  const P3 zBase{ -n0.x, -n0.y, -n0.z };
  const double yby = -zBase.z / zBase.y;
  const P3 yBase = normalized3({ 0, yby, 1 });
  const P3 xBase = cross3(yBase, zBase);
  auto mag = [](const P3 &a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); };
  if(!almostEqual(mag(n0), 1.0, 2) || !almostEqual(mag(xBase), 1.0, 2) || !almostEqual(mag(yBase), 1.0, 2))
    rc = -1;
  if(!(std::fabs(dot3(xBase, yBase)) < 1e-15) || !(std::fabs(dot3(yBase, zBase)) < 1e-15) || !(std::fabs(dot3(zBase, xBase)) < 1e-15))
    rc = -1;
  /* _a = transposed([x y z] as columns): rows are the base vectors */
  out->a[0] = xBase.x; out->a[1] = xBase.y; out->a[2] = xBase.z;
  out->a[3] = yBase.x; out->a[4] = yBase.y; out->a[5] = yBase.z;
  out->a[6] = zBase.x; out->a[7] = zBase.y; out->a[8] = zBase.z;
  const double dist = dot3(p, n0);
  if(!(dist > 0))
    rc = -1;
  out->b[0] = 0; out->b[1] = 0; out->b[2] = dist;

  /* Transformation_<2>(rp = world[0..1], rpMapping = cameraToWorld(cam[0..1])) */
  const P3 m0 = cameraToWorldD(*out, c0), m1 = cameraToWorldD(*out, c1);
  const P2 rp0{ world[0], world[1] }, rp1{ world[3], world[4] };
  const P2 d = normalized2({ rp1.x - rp0.x, rp1.y - rp0.y });
  const P2 dm = normalized2({ m1.x - m0.x, m1.y - m0.y });
  const double xBaseX = d.x * dm.x + d.y * dm.y;
  const double xBaseY = d.y * dm.x - d.x * dm.y;
  /* rot columns: xBase = (xBaseX, xBaseY), yBase = (-xBaseY, xBaseX) */
  out->r2[0] = xBaseX; out->r2[1] = -xBaseY;
  out->r2[2] = xBaseY; out->r2[3] = xBaseX;
  if(!almostEqual(std::sqrt(xBaseX * xBaseX + xBaseY * xBaseY), 1.0, 2))
    rc = -1;
  /* _b = rp.front() - _a * rpMapping.front() */
  out->t2[0] = rp0.x - (out->r2[0] * m0.x + out->r2[1] * m0.y);
  out->t2[1] = rp0.y - (out->r2[2] * m0.x + out->r2[3] * m0.y);
  out->world_z = world[2];
  return rc;
}

/* calibrationTriangle.cpp:48-68,97-125,148-172 and geometricCalibration.cpp:73-98,127-141,185-203 */
int ssdo_calibration_load(const char *triangle_path, const char *points_path, double world[9], double cam[9])
{
  auto readValue = [](std::ifstream &file, const std::string &name, auto &value)
  {
    while(file)
    {
      std::string chars;
      file >> chars;
      if(chars == name)
      {
        std::string sign;
        file >> sign;
        if(sign == "=")
        {
          file >> value;
          return static_cast<bool>(file);
        }
      }
    }
    return false;
  };
  {
    std::ifstream file(triangle_path);
    std::string id;
    std::getline(file, id);
    if(id != "calibration triangle")
      return -1;
    bool r = true;
    for(int n = 1; n <= 3; n++)
    {
      const std::string ns = std::to_string(n);
      r = r && readValue(file, "x" + ns, world[3 * (n - 1)]);
      r = r && readValue(file, "y" + ns, world[3 * (n - 1) + 1]);
      r = r && readValue(file, "z" + ns, world[3 * (n - 1) + 2]);
    }
    std::string side;
    r = r && readValue(file, std::string("lowerQuadrant"), side);
    if(!r)
      return -1;
    auto distQu = [&](int a, int b)
    {
      const double dx = world[3 * b] - world[3 * a], dy = world[3 * b + 1] - world[3 * a + 1], dz = world[3 * b + 2] - world[3 * a + 2];
      return dx * dx + dy * dy + dz * dz;
    };
    const double minDistQu = 0.01 * 0.01;
    if(distQu(0, 1) < minDistQu || distQu(1, 2) < minDistQu || distQu(2, 0) < minDistQu)
      return -1;
    if(side != "left" && side != "right")
      return -1;
  }
  {
    std::ifstream file(points_path);
    std::string id;
    std::getline(file, id);
    if(id != "calibration points")
      return -2;
    std::vector<std::array<float, 9>> sets;
    while(true)
    {
      std::array<float, 9> p;
      char ch;
      file >> p[0] >> ch >> p[1] >> ch >> p[2] >> ch >> p[3] >> ch >> p[4] >> ch >> p[5] >> ch >> p[6] >> ch >> p[7] >> ch >> p[8];
      if(!file)
        break;
      sets.push_back(p);
      if(sets.size() == 10)
        break;
    }
    if(sets.size() != 10)
      return -2;
    P3 avg[3];
    for(const auto &p : sets)
      for(int k = 0; k < 3; k++)
      {
        avg[k].x += double(p[3 * k]); avg[k].y += double(p[3 * k + 1]); avg[k].z += double(p[3 * k + 2]);
      }
    for(int k = 0; k < 3; k++)
    {
      cam[3 * k] = avg[k].x / sets.size(); cam[3 * k + 1] = avg[k].y / sets.size(); cam[3 * k + 2] = avg[k].z / sets.size();
    }
  }
  return 0;
}

int ssdo_process(const ssdo_config *cfg, const ssdo_calibration *cal, const float *xyz, ssdo_result *out,
                 uint8_t *raw_images, uint8_t *closed_images, int max_images,
                 uint8_t *ground_raw, uint8_t *ground_closed)
{
  std::memset(out, 0, sizeof(*out));
  return processFrame(*cfg, *cal, xyz, false, out, nullptr, nullptr, raw_images, closed_images, max_images, ground_raw, ground_closed);
}

int ssdo_process_lean(const ssdo_config *cfg, const ssdo_calibration *cal, const float *xyz, double *steps_ext, int *status)
{
  return processFrame(*cfg, *cal, xyz, true, nullptr, steps_ext, status, nullptr, nullptr, 0, nullptr, nullptr);
}

void ssdo_deproject(float fx, float fy, float ppx, float ppy, float depth_units, int width, int height,
                    const uint16_t *depth, float *xyz)
{
  for(int v = 0; v < height; v++)
    for(int u = 0; u < width; u++)
    {
      const size_t i = size_t(v) * width + u;
      const float px = float(u), py = float(v);
      const float x = (px - ppx) / fx;            /* rs2_deproject_pixel_to_point / pre_compute_x_y_map */
      const float y = (py - ppy) / fy;
      const float d = depth_units * float(depth[i]);
      xyz[3 * i] = d * x;
      xyz[3 * i + 1] = d * y;
      xyz[3 * i + 2] = d;
    }
}

/* EXTENSION (see ssd_oracle.h): riser evidence.  Spec, in the camera-dependent world frame:
 *   surfaces S_0..S_{n-1} = the emitted steps in output order (mean height z_i, corners FL, FR, BL, BR);
 *   riser i (0 <= i < n-1): edge = FL -> FR of S_{i+1}; len = sqrt(dx*dx + dy*dy); u = (dx, dy) / len;
 *     height interval (zLo, zHi) = (z_i + heightInterval, z_{i+1} - heightInterval); usable iff len > 0 and zLo < zHi;
 *   a bin of no plateau belongs to the lowest usable riser whose bins int((zLo - zMin) * recip) .. int((zHi - zMin) * recip)
 *     (clamped to the histogram) contain it;
 *   a point (in range, in such a bin, zLo < z < zHi) is evidence iff |s| <= tol and 0 <= t <= len with
 *     a = x - FL.x, b = y - FL.y, s = b*u.x - a*u.y, t = a*u.x + b*u.y;
 *   mean_offset = (sum of round(s * 2^40)) / 2^40 / count. */
int ssdo_risers(const ssdo_config *cfg, const ssdo_calibration *cal, const float *xyz, double tolerance, int min_support,
                ssdo_riser *out)
{
  std::unique_ptr<ssdo_result> res(new ssdo_result);
  const int rc = ssdo_process(cfg, cal, xyz, res.get(), nullptr, nullptr, 0, nullptr, nullptr);
  if(rc < 0)
    return rc;
  const int n = (res->status & SSDO_ST_THROW) ? 0 : res->n_steps;
  const int nR = n > 1 ? n - 1 : 0;
  const double recip = 1.0 / cfg->height_interval;
  const int nBins = res->n_bins;

  /* bin -> plateau, as the plateaus consume the bins in ascending order (pointcloud.cpp:300-343, quirk Q4) */
  std::vector<int> lut(size_t(nBins), -1);
  int consumedUpTo = -1;
  for(int k = 0; k < res->n_plateaus; k++)
  {
    const int hMin = res->plateaus[k].bin_lo, hMax = res->plateaus[k].bin_hi;
    int lo, hi;
    if(hMin == 0) { lo = 1; hi = 0; consumedUpTo = nBins; }
    else { lo = std::max(hMin, consumedUpTo + 1); hi = hMax; consumedUpTo = std::max(consumedUpTo, hMax); }
    for(int b = lo; b <= hi && b < nBins; b++)
      lut[size_t(b)] = k;
  }

  struct R { double ox, oy, ux, uy, len, zLo, zHi; long long sum; unsigned cnt; };
  std::vector<R> rs(size_t(nR), R{});
  std::vector<int> riserOfBin(size_t(nBins), -1);
  for(int i = 0; i < nR; i++)
  {
    const double *lower = res->steps_world[i], *upper = res->steps_world[i + 1];
    const double lx = upper[0], ly = upper[1], rx = upper[3], ry = upper[4];
    const double dx = rx - lx, dy = ry - ly;
    const double len = std::sqrt(dx * dx + dy * dy);
    R &r = rs[size_t(i)];
    r.ox = lx; r.oy = ly;
    r.zLo = lower[2] + cfg->height_interval;
    r.zHi = upper[2] - cfg->height_interval;
    const bool usable = len > 0.0 && r.zLo < r.zHi;
    r.ux = usable ? dx / len : 0.0;
    r.uy = usable ? dy / len : 0.0;
    r.len = usable ? len : -1.0;
    if(usable)
    {
      int bLo = static_cast<int>((r.zLo - cfg->z_min) * recip), bHi = static_cast<int>((r.zHi - cfg->z_min) * recip);
      bLo = std::max(0, std::min(bLo, nBins - 1));
      bHi = std::max(0, std::min(bHi, nBins - 1));
      for(int b = bLo; b <= bHi; b++)
        if(lut[size_t(b)] < 0 && riserOfBin[size_t(b)] < 0)
          riserOfBin[size_t(b)] = i;
    }
    ssdo_riser &o = out[i];
    o.height_bottom = res->steps_ext[i][0];
    o.height_top = res->steps_ext[i + 1][0];
    o.left[0] = res->steps_ext[i + 1][1]; o.left[1] = res->steps_ext[i + 1][2];
    o.right[0] = res->steps_ext[i + 1][3]; o.right[1] = res->steps_ext[i + 1][4];
  }

  const size_t nPoints = size_t(cfg->width) * size_t(cfg->height);
  for(size_t p = 0; p < nPoints && nR > 0; p++)
  {
    const float fx = xyz[3 * p], fy = xyz[3 * p + 1], fz = xyz[3 * p + 2];
    if(!(fz > 0.0f))
      continue;
    const double x = fx, y = fy, z = fz;
    double wx = (cal->a[0] * x + cal->a[1] * y) + cal->a[2] * z;
    double wy = (cal->a[3] * x + cal->a[4] * y) + cal->a[5] * z;
    double wz = (cal->a[6] * x + cal->a[7] * y) + cal->a[8] * z;
    wx = wx + cal->b[0]; wy = wy + cal->b[1]; wz = wz + cal->b[2];
    if(!(wx > cfg->x_min && wx < cfg->x_max && wy > cfg->y_min && wy < cfg->y_max && wz > cfg->z_min && wz < cfg->z_max))
      continue;
    const int bin = static_cast<int>((wz - cfg->z_min) * recip);
    if(bin < 0 || bin >= nBins)
      continue;
    const int i = riserOfBin[size_t(bin)];
    if(i < 0)
      continue;
    R &r = rs[size_t(i)];
    if(!(wz > r.zLo && wz < r.zHi))
      continue;
    const double a = wx - r.ox, b = wy - r.oy;
    const double sd = b * r.ux - a * r.uy;
    const double t = a * r.ux + b * r.uy;
    if(!(std::fabs(sd) <= tolerance && t >= 0.0 && t <= r.len))
      continue;
    r.sum += std::llrint(sd * 1099511627776.0);
    r.cnt++;
  }
  for(int i = 0; i < nR; i++)
  {
    out[i].n_points = int(rs[size_t(i)].cnt);
    out[i].detected = rs[size_t(i)].cnt >= unsigned(min_support) ? 1 : 0;
    out[i].mean_offset = rs[size_t(i)].cnt ? (double(rs[size_t(i)].sum) / 1099511627776.0) / rs[size_t(i)].cnt : 0.0;
  }
  return nR;
}

void ssdo_close3x3(uint8_t *img, int width, int height)
{
  close3x3(img, width, height);
}

int ssdo_serialize(int n_steps, const double *steps_ext, char *buf, int cap)
{
  const std::string s = serializeSteps(n_steps, steps_ext);
  std::snprintf(buf, cap, "%s", s.c_str());
  return int(s.size());
}

int ssdo_quad_test(const double quad[8], const double *pts_xy, int n, uint8_t *inside)
{
  const std::array<P2, 4> q{ { { quad[0], quad[1] }, { quad[2], quad[3] }, { quad[4], quad[5] }, { quad[6], quad[7] } } };
  const QuadTest qt(q);
  if(qt.err != QE_OK)
    return qt.err;
  for(int i = 0; i < n; i++)
    inside[i] = qt.isPointWithin({ pts_xy[2 * i], pts_xy[2 * i + 1] }) ? 1 : 0;
  return 0;
}

double ssdo_hypot(double a, double b)
{
  return std::hypot(a, b);
}

void ssdo_sort_perm(const double *dist, int n, int32_t *perm)
{
  struct PD { size_t i; double dist; };
  std::vector<PD> v;
  v.reserve(size_t(n));
  for(int i = 0; i < n; i++)
    v.push_back({ size_t(i), dist[i] });
  std::sort(v.begin(), v.end(), [](const PD &a, const PD &b) { return a.dist < b.dist; });
  for(int i = 0; i < n; i++)
    perm[i] = int32_t(v[size_t(i)].i);
}

/* the oracle's own line helpers (lineThrough, LineT::det / detx / dety above; types.h:117-163), for the pin against the
 * reference's LineCoordinates<T> */
void ssdo_line_d(const double pq[4], double abc[3])
{
  const Lined l = lineThrough<double>(P2{ pq[0], pq[1] }, P2{ pq[2], pq[3] });
  abc[0] = l.a; abc[1] = l.b; abc[2] = l.c;
}
void ssdo_line_i(const int32_t pq[4], int32_t abc[3])
{
  struct Pi { int x, y; };
  const Linei l = lineThrough<int>(Pi{ pq[0], pq[1] }, Pi{ pq[2], pq[3] });
  abc[0] = l.a; abc[1] = l.b; abc[2] = l.c;
}
void ssdo_line_dets_d(const double l[3], const double o[3], double out[3])
{
  const Lined a{ l[0], l[1], l[2] }, b{ o[0], o[1], o[2] };
  out[0] = a.det(b); out[1] = a.detx(b); out[2] = a.dety(b);
}

int ssdo_best_line(const int32_t *pts_xy, int n, int32_t line_out[3])
{
  if(n < 2)
    return -1;
  Pts2i pts;
  for(int i = 0; i < n; i++) pts.push_back({ pts_xy[2 * i], pts_xy[2 * i + 1] });
  const Linei l = bestLine(pts);
  line_out[0] = l.a; line_out[1] = l.b; line_out[2] = l.c;
  return 0;
}

} // extern "C"
