/*
 * survey_probe_frame.cpp — regenerates the frame of SURVEY.md Appendix A ("Observed (seed 12345, sigma = 1 mm,
 * K = 3, XGA)"): the survey's L515-shaped recipe with std::mt19937(12345) and
 * std::normal_distribution noise along the ray, consumed in row-major pixel order.
 *
 * TEST INFRASTRUCTURE.  Original code (the frame source of the survey's probe, not reference source); compiled by
 * tests/test_oracle.py with g++ at test time.  libstdc++'s mt19937 / normal_distribution sequences are what the
 * anchor values were produced with (same image on the GPU box).
 *
 * usage: survey_probe_frame W H out.bin [K sigma [camH pitchDeg y0 tread rise halfW outliers]]
 *        ->  writes W*H float32 xyz; prints the 3 camera-space calibration points (9 numbers) that go with the
 *            world points (-0.5,1.2,0) (0.5,1.2,0) (0.4,0.3,0).
 * The optional scene parameters are the probe's knobs (camera height and pitch, first riser distance, tread, rise,
 * half stair width, fraction of pixels replaced by a uniform depth in [0.3, 3] m).
 */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

namespace
{
struct Vec { double x, y, z; };
Vec operator+(Vec a, Vec b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
Vec operator*(Vec a, double s) { return { a.x * s, a.y * s, a.z * s }; }
double dot(Vec a, Vec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
}

int main(int argc, char **argv)
{
  if(argc < 4)
    return 2;
  const int W = atoi(argv[1]), H = atoi(argv[2]);
  const int K = argc > 4 ? atoi(argv[4]) : 3;
  const double sigma = argc > 5 ? atof(argv[5]) : 0.001;

  auto arg = [&](int i, double dflt) { return argc > i ? atof(argv[i]) : dflt; };
  /* camera 1.0 m above the ground, pitched 50 degrees down; 70 x 55 degrees field of view */
  const double pitch = arg(7, 50.0) * M_PI / 180;
  const Vec eye{ 0, 0, arg(6, 1.0) }, right{ 1, 0, 0 }, fwd{ 0, cos(pitch), -sin(pitch) }, down{ 0, -sin(pitch), -cos(pitch) };
  const double fx = (W / 2.0) / tan(35 * M_PI / 180), fy = (H / 2.0) / tan(27.5 * M_PI / 180);
  const double ppx = W / 2.0, ppy = H / 2.0;
  /* stairs: first riser 0.45 m ahead, 0.8 m wide, tread 0.28 m, rise 0.17 m, the last tread unbounded */
  const double y0 = arg(8, 0.45), tread = arg(9, 0.28), rise = arg(10, 0.17), halfW = arg(11, 0.4), outliers = arg(12, 0.0);

  std::mt19937 rng(12345);
  std::normal_distribution<double> noise(0.0, sigma);
  std::vector<float> xyz(size_t(W) * H * 3, 0.0f);

  for(int v = 0; v < H; v++)
    for(int u = 0; u < W; u++)
    {
      const double dx = (u + 0.5 - ppx) / fx, dy = (v + 0.5 - ppy) / fy;
      const Vec dir = (right * dx + down * dy) + fwd;
      double best = 1e30;
      auto consider = [&](double t, auto accept)
      {
        if(t > 1e-6 && t < best && accept(eye + dir * t))
          best = t;
      };
      if(dir.z < 0)
        consider((0 - eye.z) / dir.z, [&](Vec p) { return p.y < y0 || fabs(p.x) >= halfW; });
      for(int k = 1; k <= K; k++)
      {
        const double ya = y0 + (k - 1) * tread, yb = (k == K) ? 1e9 : y0 + k * tread, z = k * rise;
        if(dir.z < 0)
          consider((z - eye.z) / dir.z, [&](Vec p) { return p.y >= ya && p.y < yb && fabs(p.x) < halfW; });
        if(dir.y > 0)
          consider((ya - eye.y) / dir.y, [&](Vec p) { return p.z >= z - rise && p.z <= z && fabs(p.x) < halfW; });
      }
      if(best < 9.0)
      {
        double depth = best + noise(rng);
        if(outliers > 0 && std::uniform_real_distribution<double>(0, 1)(rng) < outliers)
          depth = std::uniform_real_distribution<double>(0.3, 3.0)(rng);
        float *o = &xyz[(size_t(v) * W + u) * 3];
        o[0] = float(dx * depth);
        o[1] = float(dy * depth);
        o[2] = float(depth);
      }
    }

  FILE *fp = fopen(argv[3], "wb");
  if(!fp || fwrite(xyz.data(), sizeof(float), xyz.size(), fp) != xyz.size())
    return 1;
  fclose(fp);

  const Vec world[3] = { { -0.5, 1.2, 0 }, { 0.5, 1.2, 0 }, { 0.4, 0.3, 0 } };
  for(const Vec &w : world)
  {
    const Vec q{ w.x - eye.x, w.y - eye.y, w.z - eye.z };
    printf("%.17g %.17g %.17g\n", dot(q, right), dot(q, down), dot(q, fwd));
  }
  return 0;
}
