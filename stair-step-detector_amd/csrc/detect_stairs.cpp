/*
 * detect_stairs.cpp — the detect-stairs main loop (reference detect-stairs.cpp:26-45) with the
 * camera replaced by the synthetic frame source (or a raw float32 xyz file): one
 * Stairs::serialize() line per frame on stdout, the wire format the ROS node parses
 * (ros/stair_step_detector_pkg/.../stair_step_detector.py:32-44).
 *
 *   detect-stairs-amd [--width W] [--height H] [--frames N] [--steps K] [--seed S] [--file frames.f32]
 *                     [--calibration files]   (GeometricCalibration::load() from the working directory, as detect-stairs.cpp:30)
 */
#include "../../include/stairs/stairs_api.h"
#include "../../include/ssd_source.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>
using namespace stairs;

static ssd_scene makeScene(int W, int H, int K, uint64_t seed)
{
  ssd_scene s;
  ssd_source_default_scene(&s, W, H, K, seed);
  return s;
}

int main(int argc, char **argv)
{
  int W = 1024, H = 768, frames = 1, K = 3;
  uint64_t seed = 12345;
  const char *file = nullptr;
  bool calibrationFromFiles = false;
  for(int i = 1; i + 1 < argc; i += 2)
  {
    if(!std::strcmp(argv[i], "--width")) W = std::atoi(argv[i + 1]);
    else if(!std::strcmp(argv[i], "--height")) H = std::atoi(argv[i + 1]);
    else if(!std::strcmp(argv[i], "--frames")) frames = std::atoi(argv[i + 1]);
    else if(!std::strcmp(argv[i], "--steps")) K = std::atoi(argv[i + 1]);
    else if(!std::strcmp(argv[i], "--seed")) seed = std::strtoull(argv[i + 1], nullptr, 10);
    else if(!std::strcmp(argv[i], "--file")) file = argv[i + 1];
    else if(!std::strcmp(argv[i], "--calibration")) calibrationFromFiles = !std::strcmp(argv[i + 1], "files");
  }

  Window app("stair-step-detector");

  /* calibration: three ground marks seen through the synthetic camera pose (GeometricCalibration::load) */
  const ssd_scene s0 = makeScene(W, H, K, seed);
  const double marks[3][3] = { { -0.35, 0.9, 0 }, { 0.35, 0.9, 0 }, { 0.2, 0.35, 0 } };
  GeometricTransformation::RefPoints wor, cam;
  for(int i = 0; i < 3; i++)
  {
    double c[3];
    ssd_synth_scene_to_camera(&s0, marks[i], c);
    wor[i] = Point3{ marks[i][0], marks[i][1], marks[i][2] };
    cam[i] = Point3{ c[0], c[1], c[2] };
  }
  const GeometricTransformation transSynthetic(wor, cam);
  const GeometricTransformation transFiles = GeometricCalibration::load();     /* identity when the files are missing */
  const Pointcloud pointcloud(app, calibrationFromFiles ? transFiles : transSynthetic);

  std::vector<float> xyz(static_cast<size_t>(W) * H * 3);
  FILE *fp = file ? std::fopen(file, "rb") : nullptr;
  if(file && !fp)
  {
    std::perror(file);
    return 1;
  }
  for(int f = 0; f < frames && app; f++)
  {
    if(fp)
    {
      if(std::fread(xyz.data(), sizeof(float), xyz.size(), fp) != xyz.size())
        break;
    }
    else
    {
      const ssd_scene s = makeScene(W, H, K, seed + f);
      ssd_synth_generate_host(&s, 1, xyz.data());
    }
    pointcloud.process(Camera::DepthFrame{ xyz.data(), W, H });
  }
  if(fp)
    std::fclose(fp);
  return 0;
}
