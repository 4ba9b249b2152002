/*
 * detect_stairs.cpp — the detect-stairs main loop (reference detect-stairs.cpp:26-45) with the
 * camera replaced by the synthetic frame source (or a raw float32 xyz file): one
 * Stairs::serialize() line per frame on stdout, the wire format the ROS node parses
 * (ros/stair_step_detector_pkg/.../stair_step_detector.py:32-44).
 *
 *   detect-stairs-amd [--width W] [--height H] [--frames N] [--steps K] [--seed S] [--file frames.f32]
 *                     [--calibration files]   (GeometricCalibration::load() from the working directory, as detect-stairs.cpp:30)
 *                     [--devices D | --device-list 0,1,..] [--passes P]   frame-sharded over D GPUs of this node (SURVEY.md
 *                         section 8(e); BASELINE configs[3] is --devices 8 --frames 16384): one host thread + one handle per
 *                         device, contiguous frame ranges, every shard generated into HBM and kept there (2,048 XGA frames =
 *                         19.3 GB), then — all devices from one start line, generation outside the timed region — P passes
 *                         (default 1) over the resident frames in batches of up to 2,048 through the C ABI; nothing is exchanged
 *                         between devices.  Lines (last pass) come out in frame order; per-device and total frames/s on stderr.
 *                         A device may be listed more than once.  A frame the reference would throw on ends the output there
 *                         with exit code 1, as the reference process ends.
 */
#include "../../include/stairs/stairs_api.h"
#include "../../include/ssd_source.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <iostream>
#include <mutex>
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

/* all shard threads meet here: the timed region starts when every device holds its frames */
struct StartLine
{
  std::mutex m;
  std::condition_variable cv;
  int waiting = 0, parties = 0;
  unsigned long long turn = 0;
  void arrive()
  {
    std::unique_lock<std::mutex> lk(m);
    const unsigned long long mine = turn;
    if(++waiting == parties)
    {
      waiting = 0;
      turn++;
      cv.notify_all();
    }
    else
      cv.wait(lk, [&] { return turn != mine; });
  }
};

struct ShardReport
{
  std::string error;
  double seconds = 0.0;       /* the timed region: `passes` passes over the resident frames */
  int frames = 0;
  int cpusBound = 0;          /* CPUs of the device's NUMA node the shard's thread was bound to */
  std::string where;          /* PCI bus id, UUID and NUMA node of the device: shards on distinct GPUs show distinct ids */
};

/* Frames [lo, hi) of the synthetic sequence on one device ("2,048 per GPU, 8 host threads",
 * frames resident): the whole shard is generated into HBM first, outside the timed region; then, from a common start line,
 * `passes` passes over it through the plain handle API in batches of at most kBatch frames, as many batches ahead of the
 * fetches as the handle keeps in flight.  The lines are those of the last pass. */
static void runShard(int device, const ssd_calibration &cal, int W, int H, int K, uint64_t seed, int lo, int hi, int passes,
                     std::vector<std::string> &lines, std::vector<char> &threw, StartLine &start, ShardReport &rep)
{
  const int kBatch = 2048, kGen = 256;
  rep.frames = hi - lo;
  if(hi <= lo)
  {
    start.arrive();                                         /* an empty shard (more devices than frames) only keeps the line moving */
    return;
  }
  /* this thread feeds `device`: onto the CPUs of the GPU's NUMA node before anything is allocated (pinned result slots and
   * staging buffers are placed by first touch; 0 = the platform names no local CPUs, the affinity stays) */
  rep.cpusBound = ssd_bind_thread_to_device(device);
  ssd_device_info info;
  if(ssd_device_info_get(device, &info) == SSD_OK)
    rep.where = std::string(info.pci_bus_id) + " uuid " + info.uuid + " numa " + std::to_string(info.numa_node);
  ssd_config cfg;
  ssd_default_config(&cfg, W, H);
  cfg.max_frames_per_batch = hi - lo < kBatch ? hi - lo : kBatch;
  /* batches are enqueued ahead of their fetches and the frames never change: the overlap across workspaces is ours to ask for */
  cfg.batches_in_flight = cfg.max_frames_per_batch >= 16 ? SSD_BATCHES_IN_FLIGHT_THROUGHPUT : 1;
  ssd_handle *h = nullptr;
  void *dFrames = nullptr;
  const size_t frameBytes = static_cast<size_t>(W) * H * 12;
  auto fail = [&](const char *what) { rep.error = std::string(what) + ": " + ssd_last_error(); };
  if(ssd_create(&cfg, &cal, device, &h) != SSD_OK)
    fail("ssd_create");
  else if(ssd_device_alloc(device, frameBytes * static_cast<size_t>(hi - lo), &dFrames) != SSD_OK)
    fail("ssd_device_alloc (the shard's frames stay resident)");
  std::vector<ssd_scene> scenes(kGen);
  for(int at = lo; at < hi && rep.error.empty(); at += kGen)
  {
    const int n = hi - at < kGen ? hi - at : kGen;
    for(int i = 0; i < n; i++)
      ssd_source_default_scene(&scenes[i], W, H, K, seed + static_cast<uint64_t>(at + i));
    if(ssd_synth_generate_device(scenes.data(), n, static_cast<char *>(dFrames) + frameBytes * static_cast<size_t>(at - lo), frameBytes, device, nullptr) != SSD_OK)
      rep.error = std::string("synth: ") + ssd_source_last_error();
  }
  if(rep.error.empty() && ssd_device_sync(device) != SSD_OK)
    fail("ssd_device_sync");
  start.arrive();
  if(rep.error.empty())
  {
    std::vector<ssd_frame_result> results(hi - lo);
    struct Item { int at, n; };
    std::vector<Item> work;
    for(int p = 0; p < passes; p++)
      for(int at = 0; at < hi - lo; at += cfg.max_frames_per_batch)
        work.push_back(Item{ at, hi - lo - at < cfg.max_frames_per_batch ? hi - lo - at : cfg.max_frames_per_batch });
    const int inFlight = ssd_batches_in_flight(h);
    const int ahead = (inFlight > 2 ? inFlight : 2) - 1;
    const auto t0 = std::chrono::steady_clock::now();
    for(size_t i = 0; i < work.size() && rep.error.empty(); i++)
    {
      if(ssd_enqueue(h, static_cast<char *>(dFrames) + frameBytes * static_cast<size_t>(work[i].at), frameBytes, work[i].n, nullptr) != SSD_OK)
        fail("ssd_enqueue");
      else if(i >= static_cast<size_t>(ahead) && ssd_fetch_back(h, results.data() + work[i - ahead].at, work[i - ahead].n, ahead) != SSD_OK)
        fail("ssd_fetch_back");
    }
    const int tail = static_cast<int>(work.size()) < ahead ? static_cast<int>(work.size()) : ahead;
    for(int back = tail - 1; back >= 0 && rep.error.empty(); back--)
    {
      const Item &it = work[work.size() - 1 - back];
      if(ssd_fetch_back(h, results.data() + it.at, it.n, back) != SSD_OK)
        fail("ssd_fetch_back");
    }
    rep.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<char> line(SSD_LINE_CAP);
    for(int i = 0; i < hi - lo && rep.error.empty(); i++)
    {
      /* a frame the reference would have thrown on (QuadrilateralTest, uncaught: the process ends there) has no line */
      threw[lo + i] = (results[i].status & SSD_ST_THROW) ? 1 : 0;
      if(ssd_serialize(&results[i], line.data(), line.size()) < 0)
        fail("ssd_serialize");
      lines[lo + i] = line.data();
    }
  }
  if(dFrames) ssd_device_free(device, dFrames);
  if(h) ssd_destroy(h);
}

int main(int argc, char **argv)
{
  int W = 1024, H = 768, frames = 1, K = 3, passes = 1;
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
    else if(!std::strcmp(argv[i], "--passes")) passes = std::atoi(argv[i + 1]) > 0 ? std::atoi(argv[i + 1]) : 1;
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
    std::vector<std::string> lines(frames);
    std::vector<char> threw(frames, 0);
    std::vector<ShardReport> reports(D);
    std::vector<std::thread> workers;
    StartLine start;
    start.parties = D;
    for(int d = 0; d < D; d++)
    {
      const int lo = static_cast<int>(static_cast<long long>(frames) * d / D), hi = static_cast<int>(static_cast<long long>(frames) * (d + 1) / D);
      workers.emplace_back(runShard, devices[d], std::cref(cal), W, H, K, seed, lo, hi, passes, std::ref(lines), std::ref(threw), std::ref(start),
                           std::ref(reports[d]));
    }
    for(std::thread &t : workers)
      t.join();
    double slowest = 0.0;
    for(int d = 0; d < D; d++)
    {
      if(!reports[d].error.empty())
      {
        std::cerr << "detect-stairs-amd: shard " << d << " (device " << devices[d] << "): " << reports[d].error << std::endl;
        return 1;
      }
      if(reports[d].seconds > slowest)
        slowest = reports[d].seconds;
    }
    for(int f = 0; f < frames; f++)
    {
      if(threw[f])
      {
        std::cout.flush();
        std::cerr << "detect-stairs-amd: frame " << f << ": the reference throws std::invalid_argument here (QuadrilateralTest) and terminates" << std::endl;
        return 1;
      }
      std::cout << lines[f] << "\n";
    }
    for(int d = 0; d < D; d++)
      if(reports[d].frames > 0)
        std::cerr << "  shard " << d << " on device " << devices[d] << " [" << reports[d].where << ", thread on " << reports[d].cpusBound << " local CPUs]: " << reports[d].frames << " resident frames x " << passes << " pass(es) in "
                  << reports[d].seconds << " s = " << static_cast<double>(reports[d].frames) * passes / reports[d].seconds << " frames/s" << std::endl;
    std::cerr << frames << " frames on " << D << " device shard(s), " << passes << " pass(es): "
              << (slowest > 0.0 ? static_cast<double>(frames) * passes / slowest : 0.0)
              << " frames/s (frames resident in HBM; generation outside the timed region; slowest shard " << slowest << " s)" << std::endl;
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
