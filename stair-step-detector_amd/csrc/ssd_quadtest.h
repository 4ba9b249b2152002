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

/* (selects, not conditional stores: the compiler turned "if(lo > b) lo = b; else if(up < b) up = b;" into ONE store through
 * a selected address, which put every sector into scratch memory) */
__host__ __device__ __forceinline__ void sector_init(double a, double b, double &lo, double &up)
{
  const bool below = a > b;             /* Sector(a, b): quadrilateralTest.cpp:28-40 */
  const bool above = !below && a < b;
  lo = below ? b : a;
  up = above ? b : a;
}
__host__ __device__ __forceinline__ void sector_expand(double c, double &lo, double &up)
{
  const double l = lo, u = up;
  const bool below = l > c;
  const bool above = !below && u < c;
  lo = below ? c : l;
  up = above ? c : u;
}
__host__ __device__ __forceinline__ bool sector_overlaps(double lo, double up, double olo, double oup)
{
  return lo < oup && up > olo;
}

/* i-th of three values, i in 0..2 (selects, no indexed memory) */
template<typename T>
__host__ __device__ __forceinline__ T pick3(T a0, T a1, T a2, int i)
{
  return i == 0 ? a0 : (i == 1 ? a1 : a2);
}

/* one step of the merge loop (:377-394) at the static position C of a row held in scalars: cells C and C + 1 are compared;
 * equal masks -> the cells from C + 1 on move down one place.  Returns true when merged (the loop then stays at C). */
template<int C>
__host__ __device__ __forceinline__ bool quad_merge_step(unsigned int &p0, unsigned int &p1, unsigned int &p2,
                                                         double &u0, double &u1, double &u2, int &n, int &err)
{
  const unsigned int cur = C == 0 ? p0 : p1, nxt = C == 0 ? p1 : p2;
  if((cur & 0xffu) == 0u && (nxt & 0xffu) == 0u) { err = -5; return false; }
  if(((cur >> 8) & 0xffu) > 1u && ((nxt >> 8) & 0xffu) > 1u) { err = -6; return false; }
  if((cur & 0xffu) != (nxt & 0xffu))
    return false;
  if(C == 0) { p0 = p1; u0 = u1; }
  p1 = p2; u1 = u2;
  n--;
  return true;
}

/* QuadrilateralTest::QuadrilateralTest (quadrilateralTest.cpp:275-443) flattened into tables.
 * Every array here is indexed by compile-time constants only (loops unrolled, run-time positions taken by selects), so the
 * whole construction lives in registers: k_quads builds a frame's quadrilaterals one lane each, one wave per frame — pure
 * latency.  As run-time-indexed arrays the working set sat in scratch memory (24 us per frame), then in LDS (15 us: a
 * chain of some hundred dependent LDS round trips); this form: see DESIGN.md, single-frame latency.
 * On an error (t.err != 0: the reference throws) nothing but t.err is meaningful. */
__host__ __device__ inline void build_quad_test(const double *qin /* 4 x (x,y) */, QuadTest &t)
{
  double q[8];
#pragma unroll
  for(int k = 0; k < 8; k++)
    q[k] = qin[k];
  int err = 0;
  t.fx0 = 1.0; t.fx1 = 0.0; t.fy0 = 1.0; t.fy1 = 0.0;
  double bxLo, bxUp, byLo, byUp;
  sector_init(q[0], q[2], bxLo, bxUp);
  sector_init(q[1], q[3], byLo, byUp);
  sector_expand(q[4], bxLo, bxUp); sector_expand(q[5], byLo, byUp);
  sector_expand(q[6], bxLo, bxUp); sector_expand(q[7], byLo, byUp);
  t.bxLo = bxLo; t.bxUp = bxUp; t.byLo = byLo; t.byUp = byUp;

  /* segments counterclockwise: 0->1, 1->3, 3->2, 2->0 */
  constexpr int sp[4] = { 0, 1, 3, 2 }, sq[4] = { 1, 3, 2, 0 };
  double sxLo[4], sxUp[4], syLo[4], syUp[4], segK[4], segC[4];
  bool steep[4], leftIfPositive[4];
#pragma unroll
  for(int s = 0; s < 4; s++)
  {
    const double px = q[2 * sp[s]], py = q[2 * sp[s] + 1], qx = q[2 * sq[s]], qy = q[2 * sq[s] + 1];
    sector_init(px, qx, sxLo[s], sxUp[s]);
    sector_init(py, qy, syLo[s], syUp[s]);
    const double dx = qx - px, dy = qy - py;
    const LineD l = line_through_d(px, py, qx, qy);
    steep[s] = fabs(dx) < fabs(dy);
    const double den = steep[s] ? l.a : l.b;      /* SteepLine :145-166 / FlatLine :124-143 */
    segK[s] = (steep[s] ? l.b : l.a) / den;
    segC[s] = l.c / den;
    leftIfPositive[s] = steep[s] ? !(dy > 0) : (dx > 0);
    t.segSteep[s] = steep[s] ? 1 : 0;
    t.segLeftIfPositive[s] = leftIfPositive[s] ? 1 : 0;
    t.segK[s] = segK[s];
    t.segC[s] = segC[s];
  }
#define SSD_QUAD_IS_LEFT(s, x, y) \
  ((steep[s] ? ((x) + (y) * segK[s] + segC[s] > 0) : ((x) * segK[s] + (y) + segC[s] > 0)) == leftIfPositive[s])
  const bool inside = SSD_QUAD_IS_LEFT(0, q[6], q[7]);
  t.insideIsLeft = inside ? 1 : 0;
  if(inside != SSD_QUAD_IS_LEFT(1, q[4], q[5]) || inside != SSD_QUAD_IS_LEFT(2, q[0], q[1]) || inside != SSD_QUAD_IS_LEFT(3, q[2], q[3]))
    err = -1;
#undef SSD_QUAD_IS_LEFT

  /* insertion sort of the four x and the four y (the reference sorts them; written as its compare-and-shift steps so that
   * unordered values (NaN) end up where a run-time loop would leave them) */
  double xs[4] = { q[0], q[2], q[4], q[6] }, ys[4] = { q[1], q[3], q[5], q[7] };
#pragma unroll
  for(int i = 1; i < 4; i++)
  {
    const double vx = xs[i], vy = ys[i];
    bool gx = true, gy = true;
#pragma unroll
    for(int k = i - 1; k >= 0; k--)
    {
      const bool mx = gx && xs[k] > vx;
      xs[k + 1] = mx ? xs[k] : (gx ? vx : xs[k + 1]);
      gx = mx;
      const bool my = gy && ys[k] > vy;
      ys[k + 1] = my ? ys[k] : (gy ? vy : ys[k + 1]);
      gy = my;
    }
    if(gx) xs[0] = vx;
    if(gy) ys[0] = vy;
  }

  /* rows and columns of the map: a new one wherever the sorted coordinate rises (:300-309); a row's columns do not depend
   * on the row before the merge */
  bool colLive[3], rowLive[3];
  double colLo[3], rowLo[3];
  {
    double lowerX = xs[0], lowerY = ys[0];
#pragma unroll
    for(int i = 0; i < 3; i++)
    {
      colLive[i] = lowerX < xs[i + 1];
      colLo[i] = lowerX;
      if(colLive[i]) lowerX = xs[i + 1];
      rowLive[i] = lowerY < ys[i + 1];
      rowLo[i] = lowerY;
      if(rowLive[i]) lowerY = ys[i + 1];
    }
  }
  /* cell (ri, ci) = [colLo, xs[ci + 1]] x [rowLo, ys[ri + 1]], packed: mask | count << 8 | constant << 16 */
  unsigned int cell[3][3];
#pragma unroll
  for(int ri = 0; ri < 3; ri++)
#pragma unroll
    for(int ci = 0; ci < 3; ci++)
    {
      double cxLo, cxUp, cyLo, cyUp;
      sector_init(colLo[ci], xs[ci + 1], cxLo, cxUp);
      sector_init(rowLo[ri], ys[ri + 1], cyLo, cyUp);
      const double mx = (cxLo + cxUp) / 2, my = (cyLo + cyUp) / 2;
      unsigned int mask = 0, cnt = 0, nb = 0;
#pragma unroll
      for(int s = 0; s < 4; s++)
      {
        /* (bitwise on purpose: selects and mask arithmetic instead of 36 x 6 branches) */
        const bool xo = (cxLo < sxUp[s]) & (cxUp > sxLo[s]);
        const bool yo = (cyLo < syUp[s]) & (cyUp > syLo[s]);
        const bool both = xo & yo;
        mask |= both ? 1u << s : 0u;
        cnt += both ? 1u : 0u;
        /* BBox::getRelativePosition :93-110, asked only while no segment has met the cell */
        const bool r1 = yo & (mx < sxLo[s]);
        const bool r2 = yo & (mx > sxUp[s]);
        const bool r3 = xo & (my < syLo[s]);
        const bool r4 = xo & (my > syUp[s]);
        const unsigned int rel = r1 ? 2u : (r2 ? 4u : (r3 ? 8u : (r4 ? 16u : 1u)));
        nb |= cnt == 0u ? rel : 0u;
      }
      cell[ri][ci] = mask | (cnt << 8) | ((nb & 0x1eu) == 0x1eu ? 0x10000u : 0u);
    }
  /* the live rows / columns move to the front: source of place 0 / 1 / 2 */
  const int nRows = (rowLive[0] ? 1 : 0) + (rowLive[1] ? 1 : 0) + (rowLive[2] ? 1 : 0);
  const int nCols = (colLive[0] ? 1 : 0) + (colLive[1] ? 1 : 0) + (colLive[2] ? 1 : 0);
  const int srcR[3] = { rowLive[0] ? 0 : (rowLive[1] ? 1 : 2), (rowLive[0] && rowLive[1]) ? 1 : 2, 2 };
  const int srcC[3] = { colLive[0] ? 0 : (colLive[1] ? 1 : 2), (colLive[0] && colLive[1]) ? 1 : 2, 2 };
  double rowUpper[3], colUpper[3];
  unsigned int p[3][3];
#pragma unroll
  for(int r = 0; r < 3; r++)
  {
    rowUpper[r] = pick3(ys[1], ys[2], ys[3], srcR[r]);
    colUpper[r] = pick3(xs[1], xs[2], xs[3], srcC[r]);
    unsigned int rowCells[3];
#pragma unroll
    for(int ci = 0; ci < 3; ci++)
      rowCells[ci] = pick3(cell[0][ci], cell[1][ci], cell[2][ci], srcR[r]);
#pragma unroll
    for(int c = 0; c < 3; c++)
      p[r][c] = pick3(rowCells[0], rowCells[1], rowCells[2], srcC[c]);
  }
  if(err == 0 && nRows == 0) err = -2;
  if(err == 0 && nCols == 0) err = -3;
#pragma unroll
  for(int r = 0; r < 3; r++)
#pragma unroll
    for(int c = 0; c < 3; c++)
      if(err == 0 && r < nRows && c < nCols && ((p[r][c] >> 8) & 0xffu) > 2u)
        err = -4;
  /* merge equal neighbours (:377-394): at most two steps per row of three cells */
  int nCells[3];
  double u[3][3];
#pragma unroll
  for(int r = 0; r < 3; r++)
  {
    nCells[r] = nCols;
    u[r][0] = colUpper[0]; u[r][1] = colUpper[1]; u[r][2] = colUpper[2];
    if(err == 0 && r < nRows && nCells[r] > 1)
    {
      const bool merged = quad_merge_step<0>(p[r][0], p[r][1], p[r][2], u[r][0], u[r][1], u[r][2], nCells[r], err);
      if(err == 0)
      {
        if(merged)
        {
          if(nCells[r] > 1)
            quad_merge_step<0>(p[r][0], p[r][1], p[r][2], u[r][0], u[r][1], u[r][2], nCells[r], err);
        }
        else if(nCells[r] > 2)
          quad_merge_step<1>(p[r][0], p[r][1], p[r][2], u[r][0], u[r][1], u[r][2], nCells[r], err);
      }
    }
  }
  t.err = err;
  t.nRows = static_cast<unsigned char>(nRows);
#pragma unroll
  for(int r = 0; r < 3; r++)
  {
    t.nCells[r] = r < nRows ? static_cast<unsigned char>(nCells[r]) : 0;
#pragma unroll
    for(int c = 0; c < 3; c++)
    {
      const bool live = r < nRows && c < nCells[r];
      t.cellMask[r][c] = live ? static_cast<unsigned char>(p[r][c] & 0xffu) : 0;
      t.cellConst[r][c] = live ? static_cast<unsigned char>((p[r][c] >> 16) & 1u) : 0;
    }
    t.xTrans[r][0] = r < nRows && nCells[r] > 1 ? u[r][0] : 0.0;
    t.xTrans[r][1] = r < nRows && nCells[r] > 2 ? u[r][1] : 0.0;
  }
  t.yTrans[0] = nRows > 1 ? rowUpper[0] : 0.0;
  t.yTrans[1] = nRows > 2 ? rowUpper[1] : 0.0;

  /* fast cell: the selectors put (x, y) into row r / cell c exactly when lower <= coordinate < upper
   * (quadrilateralTest.cpp:487-571), first and last cells being bounded by the strict bounding box */
  const double xAfterLo = nextafter(bxLo, 1e300), yAfterLo = nextafter(byLo, 1e300);
  double bestArea = -1.0;
  double fx0 = 1.0, fx1 = 0.0, fy0 = 1.0, fy1 = 0.0;
#pragma unroll
  for(int r = 0; r < 3; r++)
  {
    const double y0 = r == 0 ? yAfterLo : rowUpper[r > 0 ? r - 1 : 0];
    const double y1 = r == nRows - 1 ? byUp : rowUpper[r];
#pragma unroll
    for(int c = 0; c < 3; c++)
    {
      const bool candidate = err == 0 && r < nRows && c < nCells[r] && (p[r][c] & 0xffu) == 0u && ((p[r][c] >> 16) & 1u) != 0u;
      const double x0 = c == 0 ? xAfterLo : u[r][c > 0 ? c - 1 : 0];
      const double x1 = c == nCells[r] - 1 ? bxUp : u[r][c];
      const double area = (x1 - x0) * (y1 - y0);
      if(candidate && area > bestArea)
      {
        bestArea = area;
        fx0 = x0; fx1 = x1; fy0 = y0; fy1 = y1;
      }
    }
  }
  t.fx0 = fx0; t.fx1 = fx1; t.fy0 = fy0; t.fy1 = fy1;
}

/* QuadrilateralTest::isPointWithin (quadrilateralTest.cpp:445-451 and the selector lambdas :487-571) */
__host__ __device__ __forceinline__ bool quad_test(const QuadTest &t, double x, double y)
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
    const int s = __builtin_ffs(static_cast<int>(mask)) - 1;
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
__host__ __device__ inline void build_grid_segs(const QuadTest &t, double xMin, double yMin, double boxX, double boxY, QuadGridSegs &o)
{
  const double margin = 4.0e-9;
  int ok = t.err == 0 ? 1 : 0;
  o.pad = 0;
  double cx[4], cy[4], cc[4];                       /* sign-adjusted: inside <=> cx * x + cy * y + cc > 0 */
#pragma unroll
  for(int s = 0; s < 4; s++)
  {
    const bool positiveInside = (t.segLeftIfPositive[s] != 0) == (t.insideIsLeft != 0);
    const double sgn = positiveInside ? 1.0 : -1.0;
    const double k = t.segK[s];
    cx[s] = sgn * (t.segSteep[s] ? 1.0 : k);
    cy[s] = sgn * (t.segSteep[s] ? k : 1.0);
    cc[s] = sgn * t.segC[s];
    const double g0 = cx[s] / boxX, g1 = cy[s] / boxY;
    const double g2 = (cc[s] + cx[s] * xMin + cy[s] * yMin) - margin * (fabs(cx[s]) + fabs(cy[s]));
    o.g[s][0] = g0;
    o.g[s][1] = g1;
    o.g[s][2] = g2;
    if(!(fabs(g0) < 1.0e6 && fabs(g1) < 1.0e6 && fabs(g2) < 1.0e6))
      ok = 0;                                       /* degenerate edge (infinite or NaN slope) */
  }
  /* (the loops are unrolled over the whole 3 x 3 map and predicated: compile-time indices, registers) */
#pragma unroll
  for(int r = 0; r < 3; r++)
  {
    const double y0 = r == 0 ? t.byLo : t.yTrans[r > 0 ? r - 1 : 0];
    const double y1 = r == t.nRows - 1 ? t.byUp : t.yTrans[r < 2 ? r : 1];
#pragma unroll
    for(int c = 0; c < 3; c++)
    {
      const bool live = r < t.nRows && c < t.nCells[r] && t.cellMask[r][c] == 0 && t.cellConst[r][c] == 0;
      const double x0 = c == 0 ? t.bxLo : t.xTrans[r][c > 0 ? c - 1 : 0];
      const double x1 = c == t.nCells[r] - 1 ? t.bxUp : t.xTrans[r][c < 2 ? c : 1];
      const double mx = (x0 + x1) / 2, my = (y0 + y1) / 2;
      bool inside = true;
#pragma unroll
      for(int s = 0; s < 4; s++)
        inside = inside && (cx[s] * mx + cy[s] * my + cc[s] > 0);
      if(live && inside)
        ok = 0;
    }
  }
  o.ok = ok;
}

/* the boxes [x0, x1] x [y0, y1] of K1's grid cells that lie wholly inside all four edges (QuadGridSegs, ssd_device.h) */
__host__ __device__ __forceinline__ bool grid_box_inside(const QuadGridSegs &sg, int x0, int x1, int y0, int y1)
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
