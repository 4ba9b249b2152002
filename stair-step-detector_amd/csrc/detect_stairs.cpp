/*
 * detect_stairs.cpp — the detect-stairs main loop (reference detect-stairs.cpp:26-45) with the
 * camera replaced by the synthetic frame source (or a raw float32 xyz file): one
 * Stairs::serialize() line per frame on stdout, the wire format the ROS node parses
 * (ros/stair_step_detector_pkg/.../stair_step_detector.py:32-44).
 *
 *   detect-stairs-amd [--width W] [--height H] [--frames N] [--steps K] [--seed S] [--file frames.f32]
 *                     [--calibration files]   (GeometricCalibration::load() from the working directory, as detect-stairs.cpp:30)
 *                     [--devices D | --device-list 0,1,..]   frame-sharded over D GPUs of this node:
 *                         one host thread + one handle per device, contiguous frame ranges, frames generated and kept in
 *                         HBM, processed as one batch per device through the C ABI; nothing is exchanged between devices.
 *                         Lines come out in frame order; frames/s on stderr.  A device may be listed more than once.
 */
#include "../../include/stairs/stairs_api.h"
#include "../../include/ssd_source.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <iostream>
#include <string>
#include <thread>
#include <vector>
using namespace stairs;

static ssd_scene makeScene(int W, int H, int K, uint64_t seed)
{
  ssd_scene s;
  ssd_source_default_scene(&s, W, H, K, seed);
  return s;
}

/* frames [lo, hi) of the synthetic sequence on one device: generated in HBM, one batch per kBatch frames */
static void runShard(int device, const ssd_calibration &cal, int W, int H, int K, uint64_t seed, int lo, int hi,
                     std::vector<std::string> &lines, std::string &error)
{
  const int kBatch = 256;
  ssd_config cfg;
  ssd_default_config(&cfg, W, H);
  cfg.max_frames_per_batch = hi - lo < kBatch ? hi - lo : kBatch;
  ssd_handle *h = nullptr;
  void *dFrames = nullptr;
  const size_t frameBytes = static_cast<size_t>(W) * H * 12;
  auto fail = [&](const char *what) { error = std::string(what) + ": " + ssd_last_error(); };
  if(ssd_create(&cfg, &cal, device, &h) != SSD_OK) { fail("ssd_create"); return; }
  if(ssd_device_alloc(device, frameBytes * cfg.max_frames_per_batch, &dFrames) != SSD_OK) { fail("ssd_device_alloc"); ssd_destroy(h); return; }
  std::vector<ssd_scene> scenes(cfg.max_frames_per_batch);
  std::vector<ssd_frame_result> results(cfg.max_frames_per_batch);
  std::vector<char> line(SSD_LINE_CAP);
  for(int at = lo; at < hi && error.empty(); at += cfg.max_frames_per_batch)
  {
    const int n = hi - at < cfg.max_frames_per_batch ? hi - at : cfg.max_frames_per_batch;
    for(int i = 0; i < n; i++)
      ssd_source_default_scene(&scenes[i], W, H, K, seed + static_cast<uint64_t>(at + i));
    if(ssd_synth_generate_device(scenes.data(), n, dFrames, frameBytes, device, nullptr) != SSD_OK) { error = std::string("synth: ") + ssd_source_last_error(); break; }
    if(ssd_enqueue(h, dFrames, frameBytes, n, nullptr) != SSD_OK || ssd_fetch(h, results.data(), n, nullptr) != SSD_OK) { fail("ssd_enqueue"); break; }
    for(int i = 0; i < n; i++)
    {
      ssd_serialize(&results[i], line.data(), line.size());
      lines[at + i] = line.data();
    }
  }
  ssd_device_free(device, dFrames);
  ssd_destroy(h);
}

int main(int argc, char **argv)
{
  int W = 1024, H = 768, frames = 1, K = 3;
  std::vector<int> devices;
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
    else if(!std::strcmp(argv[i], "--devices"))
      for(int d = 0; d < std::atoi(argv[i + 1]); d++)
        devices.push_back(d);
    else if(!std::strcmp(argv[i], "--device-list"))
      for(const char *p = argv[i + 1]; *p; )
      {
        devices.push_back(static_cast<int>(std::strtol(p, const_cast<char **>(&p), 10)));
        if(*p == ',') p++;
      }
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

  if(!devices.empty())
  {
    /* frame-sharded over the listed devices: shard d takes frames [d F / D, (d + 1) F / D) */
    const int D = static_cast<int>(devices.size());
    if(ssd_device_count() <= 0 || file)
    {
      std::cerr << "detect-stairs-amd: --devices needs a HIP device and the synthetic source" << std::endl;
      return 1;
    }
    for(int d : devices)
      if(d < 0 || d >= ssd_device_count())
      {
        std::cerr << "detect-stairs-amd: device " << d << " not present (" << ssd_device_count() << " visible)" << std::endl;
        return 1;
      }
    const ssd_calibration &cal = (calibrationFromFiles ? transFiles : transSynthetic).constants();
    std::vector<std::string> lines(frames), errors(D);
    std::vector<std::thread> workers;
    const auto t0 = std::chrono::steady_clock::now();
    for(int d = 0; d < D; d++)
    {
      const int lo = static_cast<int>(static_cast<long long>(frames) * d / D), hi = static_cast<int>(static_cast<long long>(frames) * (d + 1) / D);
      workers.emplace_back(runShard, devices[d], std::cref(cal), W, H, K, seed, lo, hi, std::ref(lines), std::ref(errors[d]));
    }
    for(std::thread &t : workers)
      t.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for(const std::string &e : errors)
      if(!e.empty())
      {
        std::cerr << "detect-stairs-amd: " << e << std::endl;
        return 1;
      }
    for(const std::string &l : lines)
      std::cout << l << "\n";
    std::cerr << frames << " frames on " << D << " device shard(s) in " << dt << " s (" << frames / dt << " frames/s incl. frame generation)" << std::endl;
    return 0;
  }
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
