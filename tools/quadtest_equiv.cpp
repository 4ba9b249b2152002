// host equivalence check: new static-index builder vs the run-time-indexed one of the previous commit
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <cmath>
#include "ssd_device.h"
#include "ssd_math.h"
namespace ssd_old { using namespace ssd; }
#include "old_quadtest.h"
#include "ssd_quadtest.h"
using namespace ssd;
static bool same_d(double a, double b) { return std::memcmp(&a, &b, 8) == 0 || a == b || (std::isnan(a) && std::isnan(b)); }
int main(int argc, char **argv)
{
  long N = argc > 1 ? atol(argv[1]) : 2000000;
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  long nerr[8] = {0}, bad = 0, segbad = 0;
  for(long it = 0; it < N; it++)
  {
    double q[8];
    int mode = it % 8;
    if(mode < 4)
    { // tread-like: roughly axis aligned rectangle, rotated and perturbed
      double cx = U(rng), cy = 1.5 + U(rng), w = 0.2 + 0.6 * std::fabs(U(rng)), h = 0.1 + 0.3 * std::fabs(U(rng)), a = (mode == 0 ? 0.05 : 0.8) * U(rng);
      double px[4] = { -w, w, -w, w }, py[4] = { -h, -h, h, h };
      for(int k = 0; k < 4; k++)
      {
        double x = px[k] * std::cos(a) - py[k] * std::sin(a), y = px[k] * std::sin(a) + py[k] * std::cos(a);
        q[2 * k] = cx + x + 0.02 * U(rng); q[2 * k + 1] = cy + y + 0.02 * U(rng);
      }
    }
    else if(mode < 6)
      for(int k = 0; k < 8; k++) q[k] = U(rng);                 // arbitrary
    else if(mode == 6)
      for(int k = 0; k < 8; k++) q[k] = std::round(U(rng) * 3) / 3; // many equal coordinates
    else
    {
      for(int k = 0; k < 8; k++) q[k] = std::round(U(rng) * 2) / 2;
      if(it % 64 == 7) q[rng() % 8] = NAN;
      if(it % 64 == 15) q[rng() % 8] = INFINITY;
    }
    QuadTest a, b;
    std::memset(&a, 0, sizeof a); std::memset(&b, 0, sizeof b);
    ssd_old::QuadBuildScratch w;
    ssd_old::build_quad_test(q, a, w);
    ssd::build_quad_test(q, b);
    nerr[a.err < 0 ? -a.err : 0]++;
    bool ok = a.err == b.err;
    if(ok && a.err == 0)
    {
      ok = same_d(a.fx0, b.fx0) && same_d(a.fx1, b.fx1) && same_d(a.fy0, b.fy0) && same_d(a.fy1, b.fy1)
        && same_d(a.bxLo, b.bxLo) && same_d(a.bxUp, b.bxUp) && same_d(a.byLo, b.byLo) && same_d(a.byUp, b.byUp)
        && a.nRows == b.nRows && a.insideIsLeft == b.insideIsLeft && same_d(a.yTrans[0], b.yTrans[0]) && same_d(a.yTrans[1], b.yTrans[1]);
      for(int s = 0; s < 4; s++)
        ok = ok && same_d(a.segK[s], b.segK[s]) && same_d(a.segC[s], b.segC[s]) && a.segSteep[s] == b.segSteep[s] && a.segLeftIfPositive[s] == b.segLeftIfPositive[s];
      for(int r = 0; r < 3; r++)
      {
        ok = ok && a.nCells[r] == b.nCells[r] && same_d(a.xTrans[r][0], b.xTrans[r][0]) && same_d(a.xTrans[r][1], b.xTrans[r][1]);
        for(int c = 0; c < 3; c++)
          ok = ok && a.cellMask[r][c] == b.cellMask[r][c] && a.cellConst[r][c] == b.cellConst[r][c];
      }
      QuadGridSegs ga, gb;
      ssd_old::build_grid_segs(a, -2.0, 0.3, 4.0 / 256, 3.0 / 256, ga);
      ssd::build_grid_segs(b, -2.0, 0.3, 4.0 / 256, 3.0 / 256, gb);
      bool sok = ga.ok == gb.ok;
      for(int s = 0; s < 4; s++) for(int k = 0; k < 3; k++) sok = sok && same_d(ga.g[s][k], gb.g[s][k]);
      if(!sok) { if(segbad < 4) { std::printf("SEG it=%ld mode=%d ok %d vs %d\n", it, mode, ga.ok, gb.ok); for(int s=0;s<4;s++) std::printf("  %.17g %.17g %.17g | %.17g %.17g %.17g\n", ga.g[s][0],ga.g[s][1],ga.g[s][2],gb.g[s][0],gb.g[s][1],gb.g[s][2]); } segbad++; }
    }
    if(!ok)
    {
      if(bad < 5)
        std::printf("MISMATCH it=%ld mode=%d err %d vs %d q=%.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", it, mode, a.err, b.err, q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]);
      bad++;
    }
  }
  std::printf("%ld quadrilaterals: %ld builder mismatches, %ld grid-seg mismatches; err histogram 0:%ld -1:%ld -2:%ld -3:%ld -4:%ld -5:%ld -6:%ld\n",
              N, bad, segbad, nerr[0], nerr[1], nerr[2], nerr[3], nerr[4], nerr[5], nerr[6]);
  return bad || segbad ? 1 : 0;
}
