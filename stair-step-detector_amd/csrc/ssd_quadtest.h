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

/*
 * Round 6: the quadrilateral test per POINT in single precision first (QuadEdgesF, ssd_device.h) - for k_inquad, whose time is
 * the cells an edge runs through (profiles/r06_k4_ground_kernel.txt): there the reference's test - bounding box, row and cell of the
 * 3 x 3 map, up to two segments - ran for every point, nested and divergent, on doubles.
 *
 * Edge s of a built test is the line cx x + cy y + cc = 0 (sign-adjusted as in build_grid_segs: positive on the inner side).  On
 * the centred, normalised coordinates D of the range test (x = xMin + (D.x + 1/2) Rx ..: ssd_prexy.h) the same line reads
 *     E_s(D) = (cx Rx D.x + cy Ry D.y + cc + cx (xMin + Rx / 2) + cy (yMin + Ry / 2)) / N,      N = |cx Rx| + |cy Ry|,
 * so that the coefficients of D.x and D.y have absolute sum 1.  k_inquad evaluates e_s = fma(gy, d.y, fma(gx, d.x, g2)) with the
 * single-precision coefficients and the d of the pre-filter:
 *   - |d - D| <= dK M + dE0 in either coordinate for a point of magnitude M (PreXY::dK, dE0: make_pre_xy), which moves E_s by at
 *     most that (absolute sum 1);
 *   - coefficients rounded to single precision and two FMAs: at most 4.01 * 2^-24 (|gx d.x| + |gy d.y| + |g2|) <= 4.01 * 2^-24
 *     (0.51 + |g2|) for a point in range (|d| <= 1/2 + e);
 *   - the reference's own doubles (the world coordinates, then x k + y + c or x + y k + c): 2^-50 of the operands.
 * With m = 4.5 * 2^-24 (0.51 + max |g2|) + 2^-20 and h = m + dK M + dE0:  min_s e_s > h  =>  every E_s > 2^-20: the point lies on the
 * inner side of all four lines by a margin the reference's arithmetic cannot cross;  min_s e_s < -h  =>  some E_s < -2^-20.
 *
 * What the reference answers for such points is a matter of ITS map, not of geometry:
 *   inside all four lines: a cell that holds segments tests only those - all pass -, a cell without segments answers with its
 *     constant; okInside (QuadGridSegs::ok, build_grid_segs) says that no constant "outside" cell lies inside the quadrilateral.
 *     The strict bounding box holds the quadrilateral (err == 0: all four turns the same way - convex).
 *   beyond a line: the reference may still say "inside" - a cell tests only the segments whose boxes meet it, and for a long, thin,
 *     tilted quadrilateral that is not always the edge the point lies beyond (an example is in tests/test_quad_edges.py).  So the
 *     map is CHECKED, cell by cell (quad_cell_unsound): the region a cell accepts - the cell's rectangle cut by the lines of its own segments, or all
 *     of it for a constant "inside" - is a convex polygon, and E_s' >= -2^-22 for every other edge s' on it iff that holds at its
 *     vertices.  The candidates below are a superset of the vertices (rectangle corners, each line with each side, the two lines
 *     with each other), filtered to the polygon with a tolerance that only ever adds candidates.  One failing candidate: m =
 *     infinity - every point of this quadrilateral takes the doubles.  Otherwise no point a cell accepts has an E_s' below -2^-22,
 *     a point with some E_s' < -2^-20 is accepted by no cell - and a point outside the bounding box is "outside" anyway.
 */
/* the edges of a built test, sign-adjusted (cx x + cy y + cc > 0 on the inner side), with 1 / N: e[s] = { cx, cy, cc, 1 / N } */
struct QuadEdgesD
{
  double e[4][4];
  double g2max;
  int fine, pad;
};

/* step 1: the coefficients in both forms; fine = 0: a degenerate edge or range, or the reference throws */
__host__ __device__ inline void quad_edges_coeffs(const QuadTest &t, int okInside, double xMin, double xMax, double yMin, double yMax, QuadEdgesD &w, QuadEdgesF &o)
{
  const double Rx = xMax - xMin, Ry = yMax - yMin;
  const double xMid = xMin + 0.5 * Rx, yMid = yMin + 0.5 * Ry;
  bool fine = t.err == 0 && okInside != 0;
  double g2max = 0.0;
#pragma unroll
  for(int s = 0; s < 4; s++)
  {
    const bool positiveInside = (t.segLeftIfPositive[s] != 0) == (t.insideIsLeft != 0);
    const double sgn = positiveInside ? 1.0 : -1.0;
    const double k = t.segK[s];
    const double cx = sgn * (t.segSteep[s] ? 1.0 : k), cy = sgn * (t.segSteep[s] ? k : 1.0), cc = sgn * t.segC[s];
    const double N = fabs(cx * Rx) + fabs(cy * Ry);
    const double inv = 1.0 / N;
    const double g2 = (cc + cx * xMid + cy * yMid) * inv;
    w.e[s][0] = cx; w.e[s][1] = cy; w.e[s][2] = cc; w.e[s][3] = inv;
    o.gx[s] = static_cast<float>(cx * Rx * inv);
    o.gy[s] = static_cast<float>(cy * Ry * inv);
    o.g2[s] = static_cast<float>(g2);
    if(!(N > 1.0e-12 && N < 1.0e12 && fabs(g2) < 64.0))
      fine = false;                                   /* degenerate edge or range (NaNs fail the compares) */
    g2max = fmax(g2max, fabs(g2));
  }
  w.g2max = g2max;
  w.fine = fine ? 1 : 0;
  w.pad = 0;
  o.m = INFINITY;
  o.pad[0] = o.pad[1] = o.pad[2] = 0.0f;
}

/* step 2, per cell (r, c) of the map: true when the region the cell accepts reaches beyond another edge - the single-precision test
 * must not be used for this quadrilateral.  (r and c index the tables at run time: t in memory - LDS in k_quads, which deals the
 * cells of a frame's quadrilaterals out to its lanes -, not in registers.) */
__host__ __device__ inline bool quad_cell_unsound(const QuadTest &t, const QuadEdgesD &w, int r, int c)
{
  const double tol = 0x1p-22, tolFilter = 0x1p-30, tolRect = 1.0e-9;
  const unsigned int mask = t.cellMask[r][c];
  if(!(r < t.nRows && c < t.nCells[r] && (mask != 0u || t.cellConst[r][c] != 0)))
    return false;                                     /* no such cell, or a constant "outside" */
  const double y0 = r == 0 ? t.byLo : t.yTrans[r > 0 ? r - 1 : 0];
  const double y1 = r == t.nRows - 1 ? t.byUp : t.yTrans[r < 2 ? r : 1];
  const double x0 = c == 0 ? t.bxLo : t.xTrans[r][c > 0 ? c - 1 : 0];
  const double x1 = c == t.nCells[r] - 1 ? t.bxUp : t.xTrans[r][c < 2 ? c : 1];
  /* the cell's own segments: at most two (more: the reference throws, err = -4) */
  double ax = 0.0, ay = 0.0, ac = 0.0, bx = 0.0, by = 0.0, bc = 0.0;
  int n = 0;
#pragma unroll
  for(int s = 0; s < 4; s++)
  {
    const bool bit = ((mask >> s) & 1u) != 0u;
    const bool first = bit && n == 0, second = bit && n == 1;
    ax = first ? w.e[s][0] : ax; ay = first ? w.e[s][1] : ay; ac = first ? w.e[s][2] : ac;
    bx = second ? w.e[s][0] : bx; by = second ? w.e[s][1] : by; bc = second ? w.e[s][2] : bc;
    n += bit ? 1 : 0;
  }
  const double det = ax * by - bx * ay;
  const double px[13] = { x0, x1, x0, x1,
                          x0, x1, -(ac + ay * y0) / ax, -(ac + ay * y1) / ax,
                          x0, x1, -(bc + by * y0) / bx, -(bc + by * y1) / bx,
                          (ay * bc - by * ac) / det };
  const double py[13] = { y0, y0, y1, y1,
                          -(ac + ax * x0) / ay, -(ac + ax * x1) / ay, y0, y1,
                          -(bc + bx * x0) / by, -(bc + bx * x1) / by, y0, y1,
                          (bx * ac - ax * bc) / det };
  bool bad = false;
#pragma unroll
  for(int k = 0; k < 13; k++)
  {
    const bool valid = k < 4 || (k < 8 ? n >= 1 : n >= 2);
    /* (a line parallel to a side, two parallel lines: infinity or NaN, which the rectangle's compares reject) */
    bool in = valid && px[k] >= x0 - tolRect && px[k] <= x1 + tolRect && py[k] >= y0 - tolRect && py[k] <= y1 + tolRect;
    double e[4];
#pragma unroll
    for(int s = 0; s < 4; s++)
    {
      e[s] = (w.e[s][0] * px[k] + w.e[s][1] * py[k] + w.e[s][2]) * w.e[s][3];
      in = in && (((mask >> s) & 1u) == 0u || e[s] >= -tolFilter);
    }
#pragma unroll
    for(int s = 0; s < 4; s++)
      bad = bad || (in && ((mask >> s) & 1u) == 0u && !(e[s] >= -tol));
  }
  return bad;
}

/* step 3: the margin - or infinity */
__host__ __device__ inline float quad_edges_margin(const QuadEdgesD &w, bool unsound)
{
  const double m = (4.5 * 0x1p-24 * (0.51 + w.g2max) + 0x1p-20) * (1.0 + 0x1p-20);
  return (w.fine != 0 && !unsound) ? static_cast<float>(m) : INFINITY;
}

/* the three steps for one quadrilateral (the host's form: tests/test_quad_edges.py through the test hook) */
__host__ __device__ inline void build_quad_edges(const QuadTest &t, int okInside, double xMin, double xMax, double yMin, double yMax, QuadEdgesF &o)
{
  QuadEdgesD w;
  quad_edges_coeffs(t, okInside, xMin, xMax, yMin, yMax, w, o);
  bool bad = false;
  if(w.fine)
    for(int r = 0; r < 3; r++)
      for(int c = 0; c < 3; c++)
        bad = bad || quad_cell_unsound(t, w, r, c);
  o.m = quad_edges_margin(w, bad);
}

/* +1: inside by the reference's test for sure, -1: outside for sure, 0: ask the doubles.  dBound = PreXY::dK * M + PreXY::dE0 for
 * the point's magnitude M = max(|x|, |y|, |z|) of the camera coordinates (k_inquad folds dE0 into its copy of m). */
__host__ __device__ __forceinline__ int quad_edges_classify(const QuadEdgesF &E, float dx, float dy, float dBound)
{
  float emin = INFINITY;
#pragma unroll
  for(int s = 0; s < 4; s++)
    emin = fminf(emin, __builtin_fmaf(E.gy[s], dy, __builtin_fmaf(E.gx[s], dx, E.g2[s])));
  const float h = E.m + dBound;
  return emin > h ? 1 : (emin < -h ? -1 : 0);
}

} // namespace ssd

#endif /* SSD_QUADTEST_H_ */
