/*
 * stairs_source.cpp — stairs::Camera and stairs::Window::operator bool (include/stairs/stairs_api.h) over this
 * build's frame source: part of libssd_source.so.  Stands in for camera.cpp:27-49 (RealSense pipeline) and for the
 * GLFW window's "still open?" (example.hpp), neither of which exists on the GPU box.
 */
#include "../../include/stairs/stairs_api.h"
#include "../../include/ssd_source.h"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

namespace stairs
{

namespace
{
/* frames the open source still has to hand out; -1 = no source opened (a window without a camera stays open) */
std::atomic<long> g_framesLeft{ -1 };

long env_long(const char *name, long dflt)
{
  const char *v = std::getenv(name);
  return v && *v ? std::atol(v) : dflt;
}
} // namespace

struct Camera::Source
{
  int width = 640, height = 480, steps = 3;
  long frames = 1, produced = 0;
  unsigned long long seed = 12345;
  FILE *file = nullptr;
  ~Source()
  {
    if(file)
      std::fclose(file);
  }
};

void Camera::start()
{
  auto src = std::make_shared<Source>();
  src->width = static_cast<int>(env_long("SSD_SOURCE_WIDTH", 640));          /* configuration.h:36 */
  src->height = static_cast<int>(env_long("SSD_SOURCE_HEIGHT", 480));
  src->frames = env_long("SSD_SOURCE_FRAMES", 1);
  src->steps = static_cast<int>(env_long("SSD_SOURCE_STEPS", 3));
  src->seed = static_cast<unsigned long long>(env_long("SSD_SOURCE_SEED", 12345));
  if(src->width <= 0 || src->height <= 0 || src->frames < 0)
    throw std::runtime_error("Camera::start: bad SSD_SOURCE_* settings");
  if(const char *path = std::getenv("SSD_SOURCE_FILE"))
    if(*path)
    {
      src->file = std::fopen(path, "rb");
      if(!src->file)
        throw std::runtime_error(std::string("Camera::start: cannot open ") + path);
    }
  _source = src;
  g_framesLeft = src->frames;
}

Camera::Frameset Camera::waitForFrames()
{
  if(!_source)
    throw std::runtime_error("Camera::waitForFrames: start() was not called");
  Source &s = *_source;
  if(s.produced >= s.frames)
    throw std::runtime_error("Camera::waitForFrames: the frame source is exhausted");
  auto xyz = std::make_shared<std::vector<float>>(static_cast<size_t>(s.width) * s.height * 3);
  if(s.file)
  {
    if(std::fread(xyz->data(), sizeof(float), xyz->size(), s.file) != xyz->size())
    {
      g_framesLeft = 0;
      throw std::runtime_error("Camera::waitForFrames: the frame file ended");
    }
  }
  else
  {
    ssd_scene sc;
    if(ssd_source_default_scene(&sc, s.width, s.height, s.steps, s.seed + static_cast<unsigned long long>(s.produced)) != SSD_OK ||
       ssd_synth_generate_host(&sc, 1, xyz->data()) != SSD_OK)
      throw std::runtime_error(ssd_source_last_error());
  }
  s.produced++;
  g_framesLeft = s.frames - s.produced;
  Frameset f;
  f.depth = DepthFrame{ xyz->data(), s.width, s.height, xyz };
  return f;
}

Window::operator bool() const
{
  return g_framesLeft.load() != 0;
}

} // namespace stairs
