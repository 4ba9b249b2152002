/*
 * ssd_quadtest.h — QuadrilateralTest (quadrilateralTest.cpp:275-451) as the kernels build and evaluate it: the 3x3
 * cell map flattened into tables (ssd_device.h, QuadTest).  Shared by ssd_kernels.hip (k_quads, k_inquad) and the
 * test hook that runs it against the reference's golden vectors (ssd_testhooks.hip).
 */
#ifndef SSD_QUADTEST_H_
#define SSD_QUADTEST_H_

#include "ssd_device.h"
#include "ssd_math.h"

namespace ssd
{

struct SegTmp { double bxLo, bxUp, byLo, byUp; };

__device__ __forceinline__ void sector_init(double a, double b, double &lo, double &up)
{
  lo = a; up = a;                       /* Sector(a, b): quadrilateralTest.cpp:28-40 */
  if(lo > b) lo = b;
  else if(up < b) up = b;
}
__device__ __forceinline__ void sector_expand(double c, double &lo, double &up)
{
  if(lo > c) lo = c;
  else if(up < c) up = c;
}
__device__ __forceinline__ bool sector_overlaps(double lo, double up, double olo, double oup)
{
  return lo < oup && up > olo;
}

/* the builder's working arrays: indexed at run time, so as locals they would live in scratch (global) memory — k_quads
 * passes a slice of LDS instead (a frame's quadrilaterals are built by one lane each, one wave per frame: latency) */
struct QuadBuildScratch
{
  SegTmp box[4];
  double xs[4], ys[4], rowUpper[3], cellUpper[3][3];
  int nCells[3];
  unsigned char cellMask[3][3], cellCnt[3][3], cellConst[3][3];
};

/* QuadrilateralTest::QuadrilateralTest (quadrilateralTest.cpp:275-443) flattened into tables */
__device__ inline void build_quad_test(const double *q /* 4 x (x,y) */, QuadTest &t, QuadBuildScratch &w)
{
  t.err = 0;
  t.fx0 = 1.0; t.fx1 = 0.0; t.fy0 = 1.0; t.fy1 = 0.0;
  sector_init(q[0], q[2], t.bxLo, t.bxUp);
  sector_init(q[1], q[3], t.byLo, t.byUp);
  sector_expand(q[4], t.bxLo, t.bxUp); sector_expand(q[5], t.byLo, t.byUp);
  sector_expand(q[6], t.bxLo, t.bxUp); sector_expand(q[7], t.byLo, t.byUp);

  /* segments counterclockwise: 0->1, 1->3, 3->2, 2->0 */
  const int sp[4] = { 0, 1, 3, 2 }, sq[4] = { 1, 3, 2, 0 };
  SegTmp (&box)[4] = w.box;
  for(int s = 0; s < 4; s++)
  {
    const double px = q[2 * sp[s]], py = q[2 * sp[s] + 1], qx = q[2 * sq[s]], qy = q[2 * sq[s] + 1];
    sector_init(px, qx, box[s].bxLo, box[s].bxUp);
    sector_init(py, qy, box[s].byLo, box[s].byUp);
    const double dx = qx - px, dy = qy - py;
    const LineD l = line_through_d(px, py, qx, qy);
    if(fabs(dx) < fabs(dy))
    {
      t.segSteep[s] = 1;                      /* SteepLine :145-166 */
      t.segK[s] = l.b / l.a;
      t.segC[s] = l.c / l.a;
      t.segLeftIfPositive[s] = dy > 0 ? 0 : 1;
    }
    else
    {
      t.segSteep[s] = 0;                      /* FlatLine :124-143 */
      t.segK[s] = l.a / l.b;
      t.segC[s] = l.c / l.b;
      t.segLeftIfPositive[s] = dx > 0 ? 1 : 0;
    }
  }
  auto isLeft = [&](int s, double x, double y)
  {
    const bool positive = t.segSteep[s] ? (x + y * t.segK[s] + t.segC[s] > 0) : (x * t.segK[s] + y + t.segC[s] > 0);
    return t.segLeftIfPositive[s] ? positive : !positive;
  };
  const bool inside = isLeft(0, q[6], q[7]);
  t.insideIsLeft = inside ? 1 : 0;
  if(inside != isLeft(1, q[4], q[5]) || inside != isLeft(2, q[0], q[1]) || inside != isLeft(3, q[2], q[3]))
  {
    t.err = -1;
    return;
  }

  double (&xs)[4] = w.xs, (&ys)[4] = w.ys;
  xs[0] = q[0]; xs[1] = q[2]; xs[2] = q[4]; xs[3] = q[6];
  ys[0] = q[1]; ys[1] = q[3]; ys[2] = q[5]; ys[3] = q[7];
  for(int i = 1; i < 4; i++)                 /* insertion sort of 4 */
  {
    const double vx = xs[i], vy = ys[i];
    int k = i - 1;
    while(k >= 0 && xs[k] > vx) { xs[k + 1] = xs[k]; k--; }
    xs[k + 1] = vx;
    k = i - 1;
    while(k >= 0 && ys[k] > vy) { ys[k + 1] = ys[k]; k--; }
    ys[k + 1] = vy;
  }

  int nRows = 0;
  double (&rowUpper)[3] = w.rowUpper;
  int (&nCells)[3] = w.nCells;
  double (&cellUpper)[3][3] = w.cellUpper;
  unsigned char (&cellMask)[3][3] = w.cellMask, (&cellCnt)[3][3] = w.cellCnt, (&cellConst)[3][3] = w.cellConst;
  double lowerY = ys[0];
  for(int yi = 1; yi < 4; yi++)
  {
    if(!(lowerY < ys[yi]))
      continue;
    const int r = nRows++;
    rowUpper[r] = ys[yi];
    nCells[r] = 0;
    double lowerX = xs[0];
    for(int xi = 1; xi < 4; xi++)
    {
      if(!(lowerX < xs[xi]))
        continue;
      const int c = nCells[r]++;
      cellUpper[r][c] = xs[xi];
      double cxLo, cxUp, cyLo, cyUp;
      sector_init(lowerX, xs[xi], cxLo, cxUp);
      sector_init(lowerY, ys[yi], cyLo, cyUp);
      unsigned char mask = 0, cnt = 0;
      bool nb[5] = { false, false, false, false, false };
      for(int s = 0; s < 4; s++)
      {
        const bool xo = sector_overlaps(cxLo, cxUp, box[s].bxLo, box[s].bxUp);
        const bool yo = sector_overlaps(cyLo, cyUp, box[s].byLo, box[s].byUp);
        if(xo && yo)
        {
          mask |= static_cast<unsigned char>(1u << s);
          cnt++;
        }
        if(cnt == 0)
        {
          /* BBox::getRelativePosition :93-110 */
          int rel = 0;
          const double mx = (cxLo + cxUp) / 2, my = (cyLo + cyUp) / 2;
          if(yo && mx < box[s].bxLo) rel = 1;
          else if(yo && mx > box[s].bxUp) rel = 2;
          else if(xo && my < box[s].byLo) rel = 3;
          else if(xo && my > box[s].byUp) rel = 4;
          nb[rel] = true;
        }
      }
      cellMask[r][c] = mask;
      cellCnt[r][c] = cnt;
      cellConst[r][c] = (nb[1] && nb[2] && nb[3] && nb[4]) ? 1 : 0;
      lowerX = xs[xi];
    }
    lowerY = ys[yi];
  }
  if(nRows == 0) { t.err = -2; return; }
  for(int r = 0; r < nRows; r++)
  {
    if(nCells[r] == 0) { t.err = -3; return; }
    for(int c = 0; c < nCells[r]; c++)
      if(cellCnt[r][c] > 2) { t.err = -4; return; }
  }
  /* merge equal neighbours (:377-394) */
  for(int r = 0; r < nRows; r++)
  {
    int c = 0;
    while(c + 1 < nCells[r])
    {
      const unsigned char cur = cellMask[r][c], nxt = cellMask[r][c + 1];
      if(cur == 0 && nxt == 0) { t.err = -5; return; }
      if(cellCnt[r][c] > 1 && cellCnt[r][c + 1] > 1) { t.err = -6; return; }
      if(cur == nxt)
      {
        for(int k = c; k + 1 < nCells[r]; k++)
        {
          cellUpper[r][k] = cellUpper[r][k + 1];
          cellMask[r][k] = cellMask[r][k + 1];
          cellCnt[r][k] = cellCnt[r][k + 1];
          cellConst[r][k] = cellConst[r][k + 1];
        }
        nCells[r]--;
      }
      else
        c++;
    }
  }
  t.nRows = static_cast<unsigned char>(nRows);
  for(int r = 0; r < 3; r++)
  {
    t.nCells[r] = r < nRows ? static_cast<unsigned char>(nCells[r]) : 0;
    for(int c = 0; c < 3; c++)
    {
      const bool live = r < nRows && c < nCells[r];
      t.cellMask[r][c] = live ? cellMask[r][c] : 0;
      t.cellConst[r][c] = live ? cellConst[r][c] : 0;
    }
    t.xTrans[r][0] = r < nRows && nCells[r] > 1 ? cellUpper[r][0] : 0.0;
    t.xTrans[r][1] = r < nRows && nCells[r] > 2 ? cellUpper[r][1] : 0.0;
  }
  t.yTrans[0] = nRows > 1 ? rowUpper[0] : 0.0;
  t.yTrans[1] = nRows > 2 ? rowUpper[1] : 0.0;

  /* fast cell: the selectors put (x, y) into row r / cell c exactly when lower <= coordinate < upper
   * (quadrilateralTest.cpp:487-571), first and last cells being bounded by the strict bounding box */
  double bestArea = -1.0;
  for(int r = 0; r < nRows; r++)
  {
    const double y0 = r == 0 ? nextafter(t.byLo, 1e300) : rowUpper[r - 1];
    const double y1 = r == nRows - 1 ? t.byUp : rowUpper[r];
    for(int c = 0; c < nCells[r]; c++)
    {
      if(cellMask[r][c] != 0 || cellConst[r][c] == 0)
        continue;
      const double x0 = c == 0 ? nextafter(t.bxLo, 1e300) : cellUpper[r][c - 1];
      const double x1 = c == nCells[r] - 1 ? t.bxUp : cellUpper[r][c];
      const double area = (x1 - x0) * (y1 - y0);
      if(area > bestArea)
      {
        bestArea = area;
        t.fx0 = x0; t.fx1 = x1; t.fy0 = y0; t.fy1 = y1;
      }
    }
  }
}

__device__ inline void build_quad_test(const double *q, QuadTest &t)
{
  QuadBuildScratch w;
  build_quad_test(q, t, w);
}

/* QuadrilateralTest::isPointWithin (quadrilateralTest.cpp:445-451 and the selector lambdas :487-571) */
__device__ __forceinline__ bool quad_test(const QuadTest &t, double x, double y)
{
  if(!(t.bxLo < x && x < t.bxUp && t.byLo < y && y < t.byUp))
    return false;
  int r = 0;
  if(t.nRows > 1 && !(y < t.yTrans[0]))
    r = (t.nRows == 2 || y < t.yTrans[1]) ? 1 : 2;
  const int nc = t.nCells[r];
  int c = 0;
  if(nc > 1 && !(x < t.xTrans[r][0]))
    c = (nc == 2 || x < t.xTrans[r][1]) ? 1 : 2;
  unsigned int mask = t.cellMask[r][c];
  if(mask == 0)
    return t.cellConst[r][c] != 0;                     /* the large middle cell of a tread: no arithmetic */
  const bool inside = t.insideIsLeft != 0;
  bool ok = true;
  while(mask)                                          /* at most two segments per cell (:370-372) */
  {
    const int s = __ffs(static_cast<int>(mask)) - 1;
    mask &= mask - 1;
    const double k = t.segK[s], cc = t.segC[s];
    const bool positive = t.segSteep[s] ? (x + y * k + cc > 0) : (x * k + y + cc > 0);
    const bool left = t.segLeftIfPositive[s] ? positive : !positive;
    ok = ok && (left == inside);
  }
  return ok;
}

/* The edges of a built test as half-planes on K1's box grid (QuadGridSegs, ssd_device.h), for k_inquad's "is this cell wholly
 * inside the tread?" — sharper than the constant cell (fx0 ..) when the tread is turned against the axes.
 * A point that lies inside all four half-planes by the margin passes isPointWithin whichever cell of the 3 x 3 map it falls
 * into, PROVIDED the map itself agrees with the geometry: a cell that holds segments tests only those (all pass), a cell
 * without segments answers with its constant.  Such a cell is crossed by no edge, so it lies wholly inside or wholly
 * outside the quadrilateral; ok = 0 when one that lies inside (tested at its centre) carries the constant "outside". */
__device__ inline void build_grid_segs(const QuadTest &t, double xMin, double yMin, double boxX, double boxY, QuadGridSegs &o)
{
  const double margin = 4.0e-9;
  o.ok = t.err == 0 ? 1 : 0;
  o.pad = 0;
  double cx[4], cy[4], cc[4];                       /* sign-adjusted: inside <=> cx * x + cy * y + cc > 0 */
  for(int s = 0; s < 4; s++)
  {
    const bool positiveInside = (t.segLeftIfPositive[s] != 0) == (t.insideIsLeft != 0);
    const double sgn = positiveInside ? 1.0 : -1.0;
    const double k = t.segK[s];
    cx[s] = sgn * (t.segSteep[s] ? 1.0 : k);
    cy[s] = sgn * (t.segSteep[s] ? k : 1.0);
    cc[s] = sgn * t.segC[s];
    o.g[s][0] = cx[s] / boxX;
    o.g[s][1] = cy[s] / boxY;
    o.g[s][2] = (cc[s] + cx[s] * xMin + cy[s] * yMin) - margin * (fabs(cx[s]) + fabs(cy[s]));
    if(!(fabs(o.g[s][0]) < 1.0e6 && fabs(o.g[s][1]) < 1.0e6 && fabs(o.g[s][2]) < 1.0e6))
      o.ok = 0;                                     /* degenerate edge (infinite or NaN slope) */
  }
  for(int r = 0; r < t.nRows && r < 3; r++)
  {
    const double y0 = r == 0 ? t.byLo : t.yTrans[r - 1];
    const double y1 = r == t.nRows - 1 ? t.byUp : t.yTrans[r];
    for(int c = 0; c < t.nCells[r] && c < 3; c++)
    {
      if(t.cellMask[r][c] != 0 || t.cellConst[r][c] != 0)
        continue;
      const double x0 = c == 0 ? t.bxLo : t.xTrans[r][c - 1];
      const double x1 = c == t.nCells[r] - 1 ? t.bxUp : t.xTrans[r][c];
      const double mx = (x0 + x1) / 2, my = (y0 + y1) / 2;
      bool inside = true;
      for(int s = 0; s < 4; s++)
        inside = inside && (cx[s] * mx + cy[s] * my + cc[s] > 0);
      if(inside)
        o.ok = 0;
    }
  }
}

/* the boxes [x0, x1] x [y0, y1] of K1's grid cells that lie wholly inside all four edges (QuadGridSegs, ssd_device.h) */
__device__ __forceinline__ bool grid_box_inside(const QuadGridSegs &sg, int x0, int x1, int y0, int y1)
{
  const double dx0 = x0, dx1 = x1 + 1, dy0 = y0, dy1 = y1 + 1;
  bool in = sg.ok != 0;
#pragma unroll
  for(int s = 0; s < 4; s++)
  {
    const double gx = sg.g[s][0], gy = sg.g[s][1];
    in = in && (gx * (gx > 0 ? dx0 : dx1) + gy * (gy > 0 ? dy0 : dy1) + sg.g[s][2] > 0);
  }
  return in;
}

} // namespace ssd

#endif /* SSD_QUADTEST_H_ */
