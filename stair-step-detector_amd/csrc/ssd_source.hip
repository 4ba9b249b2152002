/*
 * ssd_source.hip — libssd_source.so: the synthetic frame source that stands in for the camera
 * (Camera::waitForFrames + rs2::pointcloud::calculate, reference camera.cpp:46-49, pointcloud.cpp:138) in the
 * driver, the tests and bench.py.  C ABI in include/ssd_source.h.  Not part of the product library: libssd_hip.so
 * neither links nor needs it.
 */
#include "ssd_synth.h"

#include <charconv>
#include <cmath>
#include <cstdio>
#include <string>
#include <utility>

using namespace ssd;

namespace
{

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                                   \
  do                                                                                                    \
  {                                                                                                     \
    const hipError_t e_ = (expr);                                                                       \
    if(e_ != hipSuccess)                                                                                \
      return fail(SSD_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                        \
  } while(0)

int device_count()
{
  int n = 0;
  if(hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

} // namespace

namespace ssd
{

constexpr int kThreads = 256;


__global__ __launch_bounds__(kThreads) void k_synth(const ssd_scene *__restrict__ scenes, float *__restrict__ xyz,
                                                    size_t strideFloats)
{
  const int frame = blockIdx.y;
  const ssd_scene s = scenes[frame];
  const int n = s.width * s.height;
  const uint64_t key = synth_frame_key(s);
  float *out = xyz + static_cast<size_t>(frame) * strideFloats;
  for(int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads)
  {
    const int v = i / s.width, u = i - v * s.width;
    float p[3];
    synth_pixel(s, key, u, v, p);
    out[3 * static_cast<size_t>(i)] = p[0];
    out[3 * static_cast<size_t>(i) + 1] = p[1];
    out[3 * static_cast<size_t>(i) + 2] = p[2];
  }
}

/* synthetic 16-bit depth frames: the same scenes, depth quantised to depth_units as the sensor reports it */
__global__ __launch_bounds__(kThreads) void k_synth_depth(const ssd_scene *__restrict__ scenes, unsigned short *__restrict__ depth,
                                                          size_t strideElems, float depthUnits)
{
  const int frame = blockIdx.y;
  const ssd_scene s = scenes[frame];
  const int n = s.width * s.height;
  const uint64_t key = synth_frame_key(s);
  unsigned short *out = depth + static_cast<size_t>(frame) * strideElems;
  for(int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads)
  {
    const int v = i / s.width, u = i - v * s.width;
    out[i] = synth_depth_raw(s, key, u, v, depthUnits);
  }
}


} // namespace ssd

extern "C"
{

const char *ssd_source_last_error(void)
{
  return g_err.c_str();
}


int ssd_synth_generate_host(const ssd_scene *scenes, int nframes, float *xyz)
{
  if(!scenes || !xyz || nframes < 1)
    return fail(SSD_E_ARG, "ssd_synth_generate_host: bad argument");
  size_t off = 0;
  for(int f = 0; f < nframes; f++)
  {
    const ssd_scene &s = scenes[f];
    if(s.width <= 0 || s.height <= 0)
      return fail(SSD_E_ARG, "ssd_synth_generate_host: bad scene size");
    const uint64_t key = synth_frame_key(s);
    for(int v = 0; v < s.height; v++)
      for(int u = 0; u < s.width; u++)
      {
        synth_pixel(s, key, u, v, xyz + off);
        off += 3;
      }
  }
  return SSD_OK;
}

int ssd_synth_generate_device(const ssd_scene *scenes, int nframes, void *d_xyz, size_t frame_stride_bytes, int device, void *stream)
{
  if(!scenes || !d_xyz || nframes < 1 || nframes > 65535)
    return fail(SSD_E_ARG, "ssd_synth_generate_device: bad argument");
  const int nPoints = scenes[0].width * scenes[0].height;
  for(int f = 0; f < nframes; f++)
    if(scenes[f].width != scenes[0].width || scenes[f].height != scenes[0].height)
      return fail(SSD_E_ARG, "ssd_synth_generate_device: all scenes of a batch must share one resolution");
  if(frame_stride_bytes < static_cast<size_t>(nPoints) * 12 || frame_stride_bytes % 4)
    return fail(SSD_E_ARG, "ssd_synth_generate_device: bad stride");
  if(device_count() <= 0)
    return fail(SSD_E_NODEVICE, "ssd_synth_generate_device: no HIP device");
  HIP_TRY(hipSetDevice(device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  ssd_scene *dScenes = nullptr;
  HIP_TRY(hipMalloc(&dScenes, sizeof(ssd_scene) * nframes));
  hipError_t e = hipMemcpyAsync(dScenes, scenes, sizeof(ssd_scene) * nframes, hipMemcpyHostToDevice, s);
  if(e == hipSuccess)
  {
    int bx = (nPoints + kThreads * 4 - 1) / (kThreads * 4);
    bx = bx > 2048 ? 2048 : bx < 1 ? 1 : bx;
    hipLaunchKernelGGL(k_synth, dim3(bx, nframes), dim3(kThreads), 0, s, dScenes, static_cast<float *>(d_xyz), frame_stride_bytes / 4);
    e = hipGetLastError();
  }
  if(e == hipSuccess)
    e = hipStreamSynchronize(s);
  (void)hipFree(dScenes);
  if(e != hipSuccess)
    return fail(SSD_E_HIP, std::string("ssd_synth_generate_device: ") + hipGetErrorString(e));
  return SSD_OK;
}

int ssd_synth_depth_host(const ssd_scene *scenes, int nframes, float depth_units, uint16_t *depth)
{
  if(!scenes || !depth || nframes < 1 || !(depth_units > 0.0f))
    return fail(SSD_E_ARG, "ssd_synth_depth_host: bad argument");
  size_t off = 0;
  for(int f = 0; f < nframes; f++)
  {
    const ssd_scene &s = scenes[f];
    const uint64_t key = synth_frame_key(s);
    for(int v = 0; v < s.height; v++)
      for(int u = 0; u < s.width; u++)
        depth[off++] = synth_depth_raw(s, key, u, v, depth_units);
  }
  return SSD_OK;
}

int ssd_synth_depth_device(const ssd_scene *scenes, int nframes, float depth_units, void *d_depth, size_t frame_stride_bytes,
                           int device, void *stream)
{
  if(!scenes || !d_depth || nframes < 1 || nframes > 65535 || !(depth_units > 0.0f))
    return fail(SSD_E_ARG, "ssd_synth_depth_device: bad argument");
  const int nPoints = scenes[0].width * scenes[0].height;
  for(int f = 0; f < nframes; f++)
    if(scenes[f].width != scenes[0].width || scenes[f].height != scenes[0].height)
      return fail(SSD_E_ARG, "ssd_synth_depth_device: all scenes of a batch must share one resolution");
  if(frame_stride_bytes < static_cast<size_t>(nPoints) * 2 || frame_stride_bytes % 2)
    return fail(SSD_E_ARG, "ssd_synth_depth_device: bad stride");
  if(device_count() <= 0)
    return fail(SSD_E_NODEVICE, "ssd_synth_depth_device: no HIP device");
  HIP_TRY(hipSetDevice(device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  ssd_scene *dScenes = nullptr;
  HIP_TRY(hipMalloc(&dScenes, sizeof(ssd_scene) * nframes));
  hipError_t e = hipMemcpyAsync(dScenes, scenes, sizeof(ssd_scene) * nframes, hipMemcpyHostToDevice, s);
  if(e == hipSuccess)
  {
    int bx = (nPoints + kThreads * 4 - 1) / (kThreads * 4);
    bx = bx > 2048 ? 2048 : bx < 1 ? 1 : bx;
    hipLaunchKernelGGL(k_synth_depth, dim3(bx, nframes), dim3(kThreads), 0, s, dScenes, static_cast<unsigned short *>(d_depth), frame_stride_bytes / 2, depth_units);
    e = hipGetLastError();
  }
  if(e == hipSuccess)
    e = hipStreamSynchronize(s);
  (void)hipFree(dScenes);
  if(e != hipSuccess)
    return fail(SSD_E_HIP, std::string("ssd_synth_depth_device: ") + hipGetErrorString(e));
  return SSD_OK;
}

/* the scene the driver and the compat Camera generate: the SURVEY.md section 8(d) pose (camera 1.0 m above the ground,
 * pitched 50 degrees down, 70 x 55 degrees field of view) over a staircase of n_steps steps */
int ssd_source_default_scene(ssd_scene *s, int width, int height, int n_steps, uint64_t seed)
{
  if(!s || width <= 0 || height <= 0 || n_steps < 0)
    return fail(SSD_E_ARG, "ssd_source_default_scene: bad argument");
  *s = ssd_scene{};
  const double pi = 3.14159265358979323846;
  s->width = width; s->height = height;
  s->fx = (width / 2.0) / std::tan(35.0 * pi / 180.0);
  s->fy = (height / 2.0) / std::tan(27.5 * pi / 180.0);
  s->cx = (width - 1) / 2.0; s->cy = (height - 1) / 2.0;
  s->cam_height = 1.0;
  const double pitch = 50.0 * pi / 180.0;
  s->axis_right[0] = 1; s->axis_right[1] = 0; s->axis_right[2] = 0;
  s->axis_down[0] = 0; s->axis_down[1] = -std::sin(pitch); s->axis_down[2] = -std::cos(pitch);
  s->axis_fwd[0] = 0; s->axis_fwd[1] = std::cos(pitch); s->axis_fwd[2] = -std::sin(pitch);
  s->n_steps = n_steps;
  s->first_riser_y = 0.45; s->tread = 0.28; s->rise = 0.17; s->stair_width = 0.8; s->landing = 1.0;
  s->yaw_cos = 1.0; s->yaw_sin = 0.0;
  s->sigma = 0.001;
  s->outlier_frac = 0; s->outlier_min = 0.3; s->outlier_max = 3.0;
  s->invalid_frac = 0; s->max_range = 9.0;
  s->seed = seed;
  return SSD_OK;
}

/* Writes the two files GeometricCalibration::load() reads (geometricCalibration.cpp:185-203), in the formats the
 * reference's calibration step saves them (calibrationTriangle.cpp:127-146, geometricCalibration.cpp:59-71), for
 * three ground marks seen from the scene's camera pose: what `calibrate` would leave behind for this camera. */
int ssd_source_write_calibration(const ssd_scene *scene, const double world_marks[9], const char *directory)
{
  if(!scene || !world_marks || !directory)
    return fail(SSD_E_ARG, "ssd_source_write_calibration: null argument");
  /* numbers through std::to_chars: the files must read back with operator>> in the classic locale whatever LC_NUMERIC the
   * host application has set (a decimal comma would end the reference's parse) */
  auto num = [](double v, bool asFloat)
  {
    char buf[64];
    const std::to_chars_result r = asFloat ? std::to_chars(buf, buf + sizeof(buf), static_cast<float>(v))
                                           : std::to_chars(buf, buf + sizeof(buf), v);
    return std::string(buf, r.ptr);
  };
  const std::string dir(directory);
  std::string tri = "calibration triangle\n";
  for(int n = 0; n < 3; n++)
  {
    const std::string k = std::to_string(n + 1);
    tri += "x" + k + " = " + num(world_marks[3 * n], false) + ", y" + k + " = " + num(world_marks[3 * n + 1], false) + ", z" + k + " = " +
           num(world_marks[3 * n + 2], false) + "\n";
  }
  tri += "lowerQuadrant = right\n";
  std::string pts = "calibration points\n", row;
  for(int n = 0; n < 3; n++)
  {
    const double ground[3] = { world_marks[3 * n], world_marks[3 * n + 1], 0.0 };     /* the marks lie on the ground */
    double c[3];
    ssd_synth_scene_to_camera(scene, ground, c);
    row += num(c[0], true) + ", " + num(c[1], true) + ", " + num(c[2], true) + (n < 2 ? "; " : "\n");   /* MarkerPoint3_t = Point3f */
  }
  for(int k = 0; k < 10; k++)                                                         /* __numIterations = 10 */
    pts += row;
  const std::pair<const char *, const std::string *> files[2] = { { "/calibration-triangle", &tri }, { "/calibration-points", &pts } };
  for(const auto &f : files)
  {
    FILE *fp = std::fopen((dir + f.first).c_str(), "w");
    if(!fp || std::fwrite(f.second->data(), 1, f.second->size(), fp) != f.second->size())
    {
      if(fp) std::fclose(fp);
      return fail(SSD_E_ARG, std::string("ssd_source_write_calibration: cannot write ") + (dir + f.first));
    }
    std::fclose(fp);
  }
  return SSD_OK;
}

int ssd_synth_scene_to_camera(const ssd_scene *s, const double p[3], double out[3])
{
  if(!s || !p || !out)
    return fail(SSD_E_ARG, "ssd_synth_scene_to_camera: null");
  const double v[3] = { p[0], p[1], p[2] - s->cam_height };
  out[0] = v[0] * s->axis_right[0] + v[1] * s->axis_right[1] + v[2] * s->axis_right[2];
  out[1] = v[0] * s->axis_down[0] + v[1] * s->axis_down[1] + v[2] * s->axis_down[2];
  out[2] = v[0] * s->axis_fwd[0] + v[1] * s->axis_fwd[1] + v[2] * s->axis_fwd[2];
  return SSD_OK;
}


} // extern "C"
