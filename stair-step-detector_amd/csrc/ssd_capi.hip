/*
 * ssd_capi.hip — the C ABI of libssd_hip.so (see include/ssd_hip.h for the
 * reference interface each entry point replaces).  Owns the device workspace,
 * sequences the kernels of ssd_kernels.hip on one HIP stream, never touches
 * the CPU oracle: if no HIP device is present every compute entry fails with
 * SSD_E_NODEVICE.
 */
#include "ssd_launch.h"
#include "ssd_handle.h"
#include "ssd_prexy.h"

#include <charconv>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include <sched.h>

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

#ifdef SSD_TUNING
/* tools builds only (see ssd_tuning, ssd_handle.h): overrides from the environment, read once per handle */
int env_int(const char *name, int dflt)
{
  const char *v = std::getenv(name);
  return v && *v ? std::atoi(v) : dflt;
}
void tuning_from_env(ssd_tuning &t)
{
  t.chunkPoints = env_int("SSD_CHUNK_POINTS", t.chunkPoints);
  t.targetBlocks = env_int("SSD_TARGET_BLOCKS", t.targetBlocks);
  t.k1BlocksPerFrame = env_int("SSD_K1_BLOCKS_PER_FRAME", t.k1BlocksPerFrame);
  t.k24MinBlocks = env_int("SSD_K24_MIN_BLOCKS", t.k24MinBlocks);
  t.k24TallBlocks = env_int("SSD_K24_TALL_BLOCKS", t.k24TallBlocks);
  t.k2ChunkTiles = env_int("SSD_K2_CHUNK_TILES", t.k2ChunkTiles);
  t.k4ChunkTiles = env_int("SSD_K4_CHUNK_TILES", t.k4ChunkTiles);
  t.winShift = env_int("SSD_WIN_SHIFT", 0);
  t.winShiftGround = env_int("SSD_WIN_SHIFT_G", 0);
  t.recordPad = env_int("SSD_RECORD_PAD", t.recordPad);
}
#endif

/* ssd_config::batches_in_flight = 0 resolves to ONE workspace: every call on the caller's stream, in stream order — the contract
 * every caller may rely on without reading further (enqueue, then refill the frames on the same stream: ordered).  Overlap is
 * opt-in (round 4; round 3 resolved 0 to three workspaces for batches, which silently took that order away): a caller that
 * enqueues ahead of its fetches asks for kOverlapDepth and keeps its frames untouched until their results were fetched.
 * XGA, frames/s with 1 / 2 / 3 / 4 / 5 / 6 / 8 batches in flight (tools/depths.py, profiles/r03_depths.json: three and six sit
 * better than four and five; six buys 2-5 % over three for twice the memory):
 *   1024 frames per call 272 k / 290 k / 301 k / 294 k / 298 k / 307 k / 310 k      256: 237 k / 286 k / 306 k / 295 k / 303 k / 314 k / 306 k
 *     64: 188 k / 248 k / 270 k / 251 k / 268 k / 284 k / 270 k                      32: 140 k / 225 k / 253 k / 215 k / 244 k / 256 k / 246 k
 *     16:  94 k / 165 k / 214 k / 170 k / 203 k / 228 k / 206 k                       8:  59 k / 107 k / 140 k / 114 k / 130 k / 144 k / 135 k
 * bench.py and detect-stairs-amd ask for SSD_BATCHES_IN_FLIGHT_THROUGHPUT (ssd_hip.h) explicitly. */
constexpr int kOverlapDepth = SSD_BATCHES_IN_FLIGHT_THROUGHPUT;

} // namespace


extern "C"
{

const char *ssd_last_error(void)
{
  return g_err.c_str();
}

int ssd_default_config(ssd_config *cfg, int width, int height)
{
  if(!cfg || width <= 0 || height <= 0)
    return fail(SSD_E_ARG, "ssd_default_config: bad argument");
  /* configuration.h:27-52 */
  cfg->width = width;
  cfg->height = height;
  cfg->x_min = -0.6; cfg->x_max = 0.6;
  cfg->y_min = 0.1; cfg->y_max = 1.3;
  cfg->z_min = -0.1; cfg->z_max = 1.1;
  cfg->height_interval = 0.01;
  cfg->min_height_above_ground = 0.05;
  cfg->min_step_depth = 0.1;
  cfg->max_frames_per_batch = 64;
  cfg->max_step_plateaus = SSD_MAX_STEP_IMAGES;
  cfg->batches_in_flight = 0;                 /* automatic: see ssd_config */
  return SSD_OK;
}

int ssd_calibration_identity(ssd_calibration *out)
{
  if(!out)
    return fail(SSD_E_ARG, "ssd_calibration_identity: null");
  std::memset(out, 0, sizeof(*out));
  out->a[0] = out->a[4] = out->a[8] = 1.0;
  out->r2[0] = out->r2[3] = 1.0;
  return SSD_OK;
}

/* GeometricTransformation(worldPoints, cameraPoints), transformation.cpp:196-215.
 * Vector algebra as Boost.QVM evaluates it: products summed left to right,
 * normalized(v) = v * (1 / sqrt(dot(v, v))). */
int ssd_calibration_from_points(const double w[9], const double c[9], ssd_calibration *out)
{
  if(!w || !c || !out)
    return fail(SSD_E_ARG, "ssd_calibration_from_points: null");
  struct V3 { double x, y, z; };
  auto sub = [](V3 a, V3 b) { return V3{ a.x - b.x, a.y - b.y, a.z - b.z }; };
  auto cross = [](V3 a, V3 b) { return V3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; };
  auto dot = [](V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
  auto norm = [](V3 a)
  {
    const double m2 = a.x * a.x + a.y * a.y + a.z * a.z;
    const double rm = 1.0 / std::sqrt(m2);
    return V3{ a.x * rm, a.y * rm, a.z * rm };
  };
  const V3 c0{ c[0], c[1], c[2] }, c1{ c[3], c[4], c[5] }, c2{ c[6], c[7], c[8] };

  /* Transformation_<3>(triangleInPlane), transformation.cpp:108-157 */
  const V3 n0 = norm(cross(sub(c1, c0), sub(c2, c0)));
  const V3 zB{ -n0.x, -n0.y, -n0.z };
  const V3 yB = norm(V3{ 0.0, -zB.z / zB.y, 1.0 });
  const V3 xB = cross(yB, zB);
  const double dist = dot(c0, n0);
  if(!(dist > 0.0) || !std::isfinite(yB.y) || !std::isfinite(xB.x))
    return fail(SSD_E_ARG, "ssd_calibration_from_points: degenerate triangle (reference assert, transformation.cpp:153)");
  out->a[0] = xB.x; out->a[1] = xB.y; out->a[2] = xB.z;
  out->a[3] = yB.x; out->a[4] = yB.y; out->a[5] = yB.z;
  out->a[6] = zB.x; out->a[7] = zB.y; out->a[8] = zB.z;
  out->b[0] = 0.0; out->b[1] = 0.0; out->b[2] = dist;

  /* Transformation_<2>({w0, w1}, {cameraToWorld(c0), cameraToWorld(c1)}), transformation.cpp:65-106 */
  auto c2w = [&](V3 p)
  {
    V3 r;
    r.x = out->a[0] * p.x + out->a[1] * p.y + out->a[2] * p.z;
    r.y = out->a[3] * p.x + out->a[4] * p.y + out->a[5] * p.z;
    r.z = out->a[6] * p.x + out->a[7] * p.y + out->a[8] * p.z;
    r.x = r.x + out->b[0]; r.y = r.y + out->b[1]; r.z = r.z + out->b[2];
    return r;
  };
  const V3 m0 = c2w(c0), m1 = c2w(c1);
  auto norm2 = [](double x, double y, double &ox, double &oy)
  {
    const double m2 = x * x + y * y;
    const double rm = 1.0 / std::sqrt(m2);
    ox = x * rm; oy = y * rm;
  };
  double dx, dy, dxm, dym;
  norm2(w[3] - w[0], w[4] - w[1], dx, dy);
  norm2(m1.x - m0.x, m1.y - m0.y, dxm, dym);
  const double xBaseX = dx * dxm + dy * dym;
  const double xBaseY = dy * dxm - dx * dym;
  out->r2[0] = xBaseX; out->r2[1] = -xBaseY;
  out->r2[2] = xBaseY; out->r2[3] = xBaseX;
  out->t2[0] = w[0] - (out->r2[0] * m0.x + out->r2[1] * m0.y);
  out->t2[1] = w[1] - (out->r2[2] * m0.x + out->r2[3] * m0.y);
  out->world_z = w[2];
  if(!std::isfinite(xBaseX) || !std::isfinite(xBaseY))
    return fail(SSD_E_ARG, "ssd_calibration_from_points: coincident reference points");
  return SSD_OK;
}

/* ---- calibration files -------------------------------------------- */
extern "C++"
{
namespace
{

/* What the reference's readValue (calibrationTriangle.cpp:48-68) accepts, stated on the token stream: the file from the
 * current position on is a sequence of whitespace-separated tokens; the first occurrence of the token `name` that is directly
 * followed by the token "=" selects the value, which is then extracted from the SAME stream with operator>> for T (so
 * "x1 = 0.5," yields 0.5 and leaves the comma, a string value takes the next token).  An occurrence of `name` not followed by
 * "=" consumes both tokens and the search goes on behind them.  False when the stream ends first or the extraction fails. */
template<typename T>
bool read_named_value(std::ifstream &file, const std::string &name, T &value)
{
  for(std::string token; file >> token; )
  {
    if(token != name)
      continue;
    std::string next;
    if(!(file >> next))
      break;
    if(next != "=")
      continue;
    return static_cast<bool>(file >> value);
  }
  return false;
}

/* CalibrationTriangle::load + isValid, calibrationTriangle.cpp:97-125, 148-172 */
bool load_triangle(const char *path, double w[9])
{
  std::ifstream file(path);
  std::string id;
  std::getline(file, id);
  if(id != "calibration triangle")
    return false;
  bool r = true;
  for(int n = 1; n <= 3; n++)
  {
    const std::string ns = std::to_string(n);
    r = r && read_named_value(file, "x" + ns, w[3 * (n - 1)]);
    r = r && read_named_value(file, "y" + ns, w[3 * (n - 1) + 1]);
    r = r && read_named_value(file, "z" + ns, w[3 * (n - 1) + 2]);
  }
  std::string side;
  r = r && read_named_value(file, "lowerQuadrant", side);
  if(!r)
    return false;
  const double minDistQu = 0.01 * 0.01;
  auto distQu = [&](int a, int b)
  {
    const double dx = w[3 * b] - w[3 * a], dy = w[3 * b + 1] - w[3 * a + 1], dz = w[3 * b + 2] - w[3 * a + 2];
    return dx * dx + dy * dy + dz * dz;
  };
  if(distQu(0, 1) < minDistQu || distQu(1, 2) < minDistQu || distQu(2, 0) < minDistQu)
    return false;
  return side == "left" || side == "right";
}

/* loadPoints + calcAverageRefPointSet, geometricCalibration.cpp:73-98, 127-141: ten rows of three float
 * points, summed in double in file order, divided by the count */
bool load_points(const char *path, double c[9])
{
  std::ifstream file(path);
  std::string id;
  std::getline(file, id);
  if(id != "calibration points")
    return false;
  const int numIterations = 10;
  double sum[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  int n = 0;
  while(true)
  {
    float p[9];
    char ch;
    for(int k = 0; k < 3; k++)
    {
      file >> p[3 * k] >> ch >> p[3 * k + 1] >> ch >> p[3 * k + 2];
      if(k < 2)
        file >> ch;
    }
    if(!file)
      break;
    for(int k = 0; k < 9; k++)
      sum[k] += static_cast<double>(p[k]);
    if(++n == numIterations)
    {
      for(int k = 0; k < 9; k++)
        c[k] = sum[k] / static_cast<double>(static_cast<size_t>(n));
      return true;
    }
  }
  return false;
}

} // namespace
} // extern "C++"

int ssd_calibration_load(const char *triangle_path, const char *points_path, ssd_calibration *out, int *loaded,
                         double *world_points, double *camera_points)
{
  if(!triangle_path || !points_path || !out)
    return fail(SSD_E_ARG, "ssd_calibration_load: null argument");
  double w[9], c[9];
  if(loaded)
    *loaded = 0;
  if(load_triangle(triangle_path, w) && load_points(points_path, c))
  {
    if(world_points) std::memcpy(world_points, w, sizeof(w));
    if(camera_points) std::memcpy(camera_points, c, sizeof(c));
    const int rc = ssd_calibration_from_points(w, c, out);
    if(rc == SSD_OK && loaded)
      *loaded = 1;
    return rc;
  }
  return ssd_calibration_identity(out);      /* geometricCalibration.cpp:199-202: log an error, return {} */
}

int ssd_device_count(void)
{
  int n = 0;
  if(hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

static int make_params(const ssd_config &c, const ssd_calibration &k, Params &P)
{
  /* height: a pixel key's row field is 13 bits and "inside the wave's LDS window" is one subtraction on the key
   * (ssd_kernels.hip, window_hit): a window anchored within its own height (<= 128 rows) of row 8192 would take pixels of
   * the next image slot for its own */
  if(c.width <= 0 || c.height <= 0 || c.width > 8192 || c.height > 8064)
    return fail(SSD_E_ARG, "config: resolution out of range (width <= 8192, height <= 8064)");
  if(!(c.x_max > c.x_min) || !(c.y_max > c.y_min) || !(c.z_max > c.z_min) || !(c.height_interval > 0))
    return fail(SSD_E_ARG, "config: empty measuring range");
  P.W = c.width; P.H = c.height;
  P.W64 = (c.width + 63) / 64;
  P.nPoints = c.width * c.height;
  for(int i = 0; i < 9; i++) P.pt.a[i] = k.a[i];
  for(int i = 0; i < 3; i++) P.pt.b[i] = k.b[i];
  for(int i = 0; i < 4; i++) P.r2[i] = k.r2[i];
  P.t2[0] = k.t2[0]; P.t2[1] = k.t2[1];
  P.worldZ = k.world_z;
  P.xMin = c.x_min; P.xMax = c.x_max; P.yMin = c.y_min; P.yMax = c.y_max; P.zMin = c.z_min; P.zMax = c.z_max;
  /* ProcessingConfiguration / Projection2D, pointcloud.cpp:60-106 */
  P.pt.recip = 1.0 / c.height_interval;
  P.minHeight = static_cast<uint16_t>((c.min_height_above_ground - c.z_min) * P.pt.recip);
  P.minImgYExtent = static_cast<int>(c.min_step_depth * c.height / (c.y_max - c.y_min));
  P.xToImage = c.width / (c.x_max - c.x_min);
  P.yToImage = c.height / (c.y_max - c.y_min);
  P.xToWorld = 1 / P.xToImage;
  P.yToWorld = 1 / P.yToImage;
  P.xyRatio = P.xToImage / P.yToImage;
  const size_t nBins = static_cast<size_t>((c.z_max - c.z_min) * P.pt.recip) + 1;   /* pointcloud.cpp:196 */
  if(nBins < 3 || nBins > SSD_MAX_BINS)
    return fail(SSD_E_ARG, "config: histogram needs 3.." + std::to_string(SSD_MAX_BINS) + " bins");
  P.nBins = static_cast<int>(nBins);
  /* mean z is accumulated as a sum of round(z * 2^40) in int64 through a magic-constant add that is exact only for
   * |z| < 2048 (ssd_kernels.hip, z_plus_magic_bits), and the sum over a frame's points must stay below 2^63 */
  {
    const double zAbs = std::fmax(std::fabs(c.z_min), std::fabs(c.z_max));
    if(!(zAbs < 2048.0) || !(zAbs * static_cast<double>(c.width) * static_cast<double>(c.height) < 8388608.0))
      return fail(SSD_E_ARG, "config: max(|z_min|, |z_max|) must be < 2048 m and max|z| * width * height < 2^23 (fixed-point mean z)");
  }
  if(c.max_step_plateaus < 1 || c.max_step_plateaus > SSD_MAX_STEP_IMAGES)
    return fail(SSD_E_ARG, "config: max_step_plateaus out of range");
  P.maxStepImages = c.max_step_plateaus;
  P.pt.xMin = c.x_min; P.pt.xMax = c.x_max; P.pt.yMin = c.y_min; P.pt.yMax = c.y_max; P.pt.zMin = c.z_min; P.pt.zMax = c.z_max;
  P.pt.boxX = 256.0 / (c.x_max - c.x_min);
  P.pt.boxY = 256.0 / (c.y_max - c.y_min);
  P.pt.nPoints = P.nPoints;
  P.pt.nBins = P.nBins;
  P.pre = make_pre_xy(P.pt);                  /* K1's single-precision pre-filter of the x / y range test (ssd_prexy.h) */
  P.px.xToImage = P.xToImage; P.px.yToImage = P.yToImage;
  P.px.W = P.W; P.px.H = P.H; P.px.W64 = P.W64;
  P.px.maxStepImages = P.maxStepImages;
  P.px.cellCols = (c.width + kCellHost / 2) / kCellHost > 0 ? (c.width + kCellHost / 2) / kCellHost : 1;
  make_pre_pixel(P.pt, P.pre, P.px);          /* K1's candidates: the pixel in single precision first (ssd_prexy.h) */
  P.risers = 0; P.riserMinSupport = 1; P.riserTol = 0.0;
  P.heightInterval = c.height_interval;
  /* Shape of the waves' write-combining windows (ssd_kernels.hip, WaveWindow).  A wave walks down one cell column —
   * 64 camera pixels wide, which is 60 .. 110 pixels of the top-down image depending on the range — so the window is
   * tall and narrow: 64 rows x 4 words up to XGA, 32 rows x 8 words above (measured, tools/exp.sh: XGA raster 1.52 ms
   * with 32 x 8, 1.00 ms with 64 x 4, 2.23 ms with 16 x 16; FHD stress 1.04 ms with 32 x 8, 1.24 ms with 64 x 4).  The
   * ground image of k_inquad is denser (the camera looks steeply down on it): 32 x 8.  When a camera row is not a
   * whole number of cells the columns drift sideways from row to row: a window as wide as the image row then. */
  P.px.winShift = P.W64 <= 16 ? 2 : 3;
  P.px.winShiftGround = 3;
  if(c.width % kCellHost != 0 && P.W64 <= 32)
  {
    while((1 << P.px.winShift) < P.W64)
      P.px.winShift++;
    P.px.winShiftGround = P.px.winShift;
  }
  if((c.width - 1) / 25 + 2 > SSD_MAX_SCANS)
    return fail(SSD_E_ARG, "config: width needs more scan columns than SSD_MAX_SCANS");
  return SSD_OK;
}

int ssd_create(const ssd_config *cfg, const ssd_calibration *cal, int device, ssd_handle **out)
{
  if(!cfg || !cal || !out)
    return fail(SSD_E_ARG, "ssd_create: null argument");
  *out = nullptr;
  if(cfg->max_frames_per_batch < 1 || cfg->max_frames_per_batch > 65535)
    return fail(SSD_E_ARG, "ssd_create: max_frames_per_batch out of range");
  int depth = cfg->batches_in_flight;
  if(depth < 0 || depth > kMaxLanes)
    return fail(SSD_E_ARG, "ssd_create: batches_in_flight must be 0 (automatic) .. " + std::to_string(kMaxLanes));
  if(depth == 0)
    depth = 1;                                  /* strict stream order unless the caller asks for overlap (see above) */
  static_assert(kOverlapDepth >= 2 && kOverlapDepth <= kMaxLanes, "SSD_BATCHES_IN_FLIGHT_THROUGHPUT out of range");
  Params P{};
  const int rc = make_params(*cfg, *cal, P);
  if(rc)
    return rc;
  const int nDev = ssd_device_count();
  if(nDev <= 0)
    return fail(SSD_E_NODEVICE, "ssd_create: no HIP device (the HIP path is mandatory; there is no CPU fallback)");
  if(device < 0 || device >= nDev)
    return fail(SSD_E_ARG, "ssd_create: device index out of range");
  HIP_TRY(hipSetDevice(device));

  ssd_handle *h = new ssd_handle();
  h->device = device;
  h->cfg = *cfg;
  h->cfg.batches_in_flight = depth;
  h->P = P;
#ifdef SSD_TUNING
  tuning_from_env(h->tune);
  if(h->tune.winShift >= 1 && h->tune.winShift <= 6) h->P.px.winShift = h->tune.winShift;
  if(h->tune.winShiftGround >= 1 && h->tune.winShiftGround <= 6) h->P.px.winShiftGround = h->tune.winShiftGround;
#endif
  h->F = cfg->max_frames_per_batch;
  h->depth = depth;
  h->nSlots = depth > 2 ? depth : 2;
  h->imgWords = static_cast<size_t>(P.H) * P.W64;
  const size_t stepBytes = static_cast<size_t>(h->F) * P.maxStepImages * h->imgWords * 8;
  const size_t groundBytes = static_cast<size_t>(h->F) * h->imgWords * 8;
  /* the planes of the single pass, for handles whose batches can qualify (ssd_launch.h) */
  const size_t planeBytes = single_pass_geometry(P.W, P.H) && single_pass_batch(h->F, P.nPoints)
                            ? static_cast<size_t>(plane_pool_size(h->F, h->P.nPoints)) * h->imgWords * 8 : 0;
  h->planePool = static_cast<int>(plane_pool_size(h->F, h->P.nPoints));
  auto cleanup = [&]() { ssd_destroy(h); };
#define HIP_TRY_H(expr)                                                                                 \
  do                                                                                                    \
  {                                                                                                     \
    const hipError_t e_ = (expr);                                                                       \
    if(e_ != hipSuccess)                                                                                \
    {                                                                                                   \
      cleanup();                                                                                        \
      return fail(e_ == hipErrorOutOfMemory ? SSD_E_NOMEM : SSD_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    }                                                                                                   \
  } while(0)
  h->tileMaskStride = (static_cast<size_t>(P.nPoints) + kTileHost - 1) / kTileHost * (kTileHost / kCellHost)      /* cell records per frame */
                      + static_cast<size_t>(h->tune.recordPad);
  const size_t maskBytes = h->tileMaskStride * sizeof(uint2) * h->F;
  h->recordBytes = maskBytes;
  for(int k = 0; k < depth; k++)
  {
    ssd_lane &L = h->lane[k];
    HIP_TRY_H(hipMalloc(&L.dState, sizeof(FrameState) * h->F));
    HIP_TRY_H(hipMalloc(&L.dStepImg, stepBytes));
    HIP_TRY_H(hipMalloc(&L.dGroundImg, groundBytes));
    HIP_TRY_H(hipMalloc(&L.dTileMasksBase, maskBytes));
    L.dTileMasks = L.dTileMasksBase;
    HIP_TRY_H(hipEventCreateWithFlags(&L.in, hipEventDisableTiming));
    HIP_TRY_H(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
    if(depth > 1)
      HIP_TRY_H(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
    HIP_TRY_H(hipMemset(L.dState, 0, sizeof(FrameState) * h->F));
    HIP_TRY_H(hipMemset(L.dStepImg, 0, stepBytes));
    HIP_TRY_H(hipMemset(L.dGroundImg, 0, groundBytes));
  }
  const size_t resBytes = sizeof(ssd_frame_result) * h->F * h->nSlots;
  HIP_TRY_H(hipMalloc(&h->dResults, resBytes));
  HIP_TRY_H(hipHostMalloc(&h->hResults, resBytes, hipHostMallocDefault));
  std::memset(h->hResults, 0, resBytes);
  HIP_TRY_H(hipHostGetDevicePointer(reinterpret_cast<void **>(&h->hResultsDev), h->hResults, 0));
  for(int k = 0; k < h->nSlots; k++)
    HIP_TRY_H(hipEventCreateWithFlags(&h->resultsReady[k], hipEventDisableTiming));
  HIP_TRY_H(hipMemset(h->dResults, 0, resBytes));
  /* The planes of the single pass are an optimisation (results are the same without them): when they do not fit beside the
   * rest - a larger batch, more workspaces, memory shared with the caller's frames - the handle gives up the planes of ALL its
   * workspaces and runs two passes, as ssd_set_single_pass(h, 0) would; it does not fail.  They are taken LAST,
   * behind everything the handle cannot do without: a device with room for the planes but not for what
   * followed them used to fail the whole handle.  SSD_MAX_PLANE_BYTES (environment) bounds what a handle may take for the planes
   * of all its workspaces together - for a GPU shared with other tenants; a workspace whose planes would cross it counts as one
   * whose allocation failed.  Why a handle has no planes is kept for ssd_last_error() (ssd_create still returns SSD_OK). */
  size_t planeBytesHeld = 0;
  if(planeBytes)
  {
    bool ok = true;
    std::string why;
    unsigned long long cap = ~0ull;
    if(const char *e = std::getenv("SSD_MAX_PLANE_BYTES"))
      cap = std::strtoull(e, nullptr, 10);
    for(int k = 0; k < depth && ok; k++)
    {
      ssd_lane &L = h->lane[k];
      if(static_cast<unsigned long long>(k + 1) * planeBytes > cap)
      {
        ok = false;
        why = "SSD_MAX_PLANE_BYTES = " + std::to_string(cap) + " < " + std::to_string(static_cast<unsigned long long>(depth) * planeBytes);
        break;
      }
      const hipError_t e1 = hipMalloc(&L.dPlaneImg, planeBytes);
      const hipError_t e2 = e1 == hipSuccess ? hipMalloc(&L.dFallback, sizeof(int) * (kFallbackList + static_cast<size_t>(h->F))) : e1;
      if(e2 != hipSuccess)
      {
        ok = false;
        why = std::string("hipMalloc of ") + std::to_string(planeBytes) + " bytes for workspace " + std::to_string(k) + ": " + hipGetErrorString(e2);
      }
    }
    if(ok && hipHostMalloc(&h->hFallback, sizeof(int) * 2 * kMaxLanes, hipHostMallocDefault) != hipSuccess)
    {
      ok = false;
      why = "hipHostMalloc of the work lists' counters failed";
    }
    if(!ok)
    {
      (void)hipGetLastError();                   /* the failed allocation's error is not the handle's */
      g_err = "ssd_create: no planes for the single pass (" + why + "): the handle runs two passes";
      for(int k = 0; k < depth; k++)
      {
        ssd_lane &L = h->lane[k];
        if(L.dPlaneImg) { (void)hipFree(L.dPlaneImg); L.dPlaneImg = nullptr; }
        if(L.dFallback) { (void)hipFree(L.dFallback); L.dFallback = nullptr; }
      }
      if(h->hFallback) { (void)hipHostFree(h->hFallback); h->hFallback = nullptr; }
      h->singlePassMode = 0;
    }
    else
    {
      for(int k = 0; k < depth; k++)
      {
        HIP_TRY_H(hipMemset(h->lane[k].dPlaneImg, 0, planeBytes));
        HIP_TRY_H(hipMemset(h->lane[k].dFallback, 0, sizeof(int) * (kFallbackList + static_cast<size_t>(h->F))));
      }
      std::memset(h->hFallback, 0, sizeof(int) * 2 * kMaxLanes);
      planeBytesHeld = planeBytes;
    }
  }
  HIP_TRY_H(hipDeviceSynchronize());
#undef HIP_TRY_H
  h->bytes = static_cast<size_t>(depth) * (sizeof(FrameState) * h->F + stepBytes + groundBytes + planeBytesHeld + maskBytes) + resBytes;
  *out = h;
  return SSD_OK;
}

int ssd_destroy(ssd_handle *h)
{
  if(!h)
    return SSD_OK;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();               /* nothing of this handle is in flight any more (lanes run on their own streams) */
  for(ssd_lane &L : h->lane)
  {
    if(L.dState) (void)hipFree(L.dState);
    if(L.dStepImg) (void)hipFree(L.dStepImg);
    if(L.dGroundImg) (void)hipFree(L.dGroundImg);
    if(L.dPlaneImg) (void)hipFree(L.dPlaneImg);
    if(L.dFallback) (void)hipFree(L.dFallback);
    if(L.dTileMasksBase) (void)hipFree(L.dTileMasksBase);
    if(L.in) (void)hipEventDestroy(L.in);
    if(L.done) (void)hipEventDestroy(L.done);
    if(L.stream) (void)hipStreamDestroy(L.stream);
  }
  if(h->hFallback) (void)hipHostFree(h->hFallback);
  if(h->dDepthMaps) (void)hipFree(h->dDepthMaps);
  if(h->dResults) (void)hipFree(h->dResults);
  if(h->hResults) (void)hipHostFree(h->hResults);
  for(hipEvent_t e : h->resultsReady)
    if(e) (void)hipEventDestroy(e);
  for(int k = 0; k < 2; k++)
  {
    if(h->ingestBuf[k]) (void)hipFree(h->ingestBuf[k]);
    if(h->ingestCopied[k]) (void)hipEventDestroy(h->ingestCopied[k]);
    if(h->ingestCopied2[k]) (void)hipEventDestroy(h->ingestCopied2[k]);
    if(h->ingestConsumed[k]) (void)hipEventDestroy(h->ingestConsumed[k]);
  }
  if(h->ingestCopy) (void)hipStreamDestroy(h->ingestCopy);
  if(h->ingestCopy2) (void)hipStreamDestroy(h->ingestCopy2);
  if(h->ingestCompute) (void)hipStreamDestroy(h->ingestCompute);
  if(h->dRisers) (void)hipFree(h->dRisers);
  if(h->hRisers) (void)hipHostFree(h->hRisers);
  if(h->hRisersBatch) (void)hipHostFree(h->hRisersBatch);
  if(h->dDebug) (void)hipFree(h->dDebug);
  if(h->dDebugImg) (void)hipFree(h->dDebugImg);
  for(hipEvent_t e : h->evPredict)
    if(e) (void)hipEventDestroy(e);
  for(hipEvent_t e : h->ev)
    (void)hipEventDestroy(e);
  delete h;
  return SSD_OK;
}

size_t ssd_workspace_bytes(const ssd_handle *h)
{
  return h ? h->bytes : 0;
}

int ssd_set_single_pass(ssd_handle *h, int enable)
{
  if(!h)
    return fail(SSD_E_ARG, "ssd_set_single_pass: null handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());            /* the planes of batches in flight are in use */
  const size_t planeBytes = static_cast<size_t>(plane_pool_size(h->F, h->P.nPoints)) * h->imgWords * 8;
  const size_t listBytes = sizeof(int) * (kFallbackList + static_cast<size_t>(h->F));
  if(!enable)
  {
    for(int k = 0; k < h->depth; k++)
    {
      ssd_lane &L = h->lane[k];
      if(L.dPlaneImg)
      {
        (void)hipFree(L.dPlaneImg);
        L.dPlaneImg = nullptr;
        h->bytes -= planeBytes;
      }
      if(L.dFallback)
      {
        (void)hipFree(L.dFallback);
        L.dFallback = nullptr;
      }
    }
    h->singlePassMode = 0;
    return SSD_OK;
  }
  if(single_pass_geometry(h->P.W, h->P.H) && single_pass_batch(h->F, h->P.nPoints))
  {
    for(int k = 0; k < h->depth; k++)
    {
      ssd_lane &L = h->lane[k];
      if(!L.dPlaneImg)
      {
        unsigned long long cap = ~0ull;
        if(const char *ev = std::getenv("SSD_MAX_PLANE_BYTES"))
          cap = std::strtoull(ev, nullptr, 10);
        const hipError_t e = static_cast<unsigned long long>(k + 1) * planeBytes > cap ? hipErrorOutOfMemory : hipMalloc(&L.dPlaneImg, planeBytes);
        if(e != hipSuccess)
        {
          /* asked for explicitly, so said loudly - but the handle stays whole: on two passes, with no plane of any workspace */
          L.dPlaneImg = nullptr;
          (void)hipGetLastError();
          (void)ssd_set_single_pass(h, 0);
          return fail(e == hipErrorOutOfMemory ? SSD_E_NOMEM : SSD_E_HIP, std::string("ssd_set_single_pass: ") + hipGetErrorString(e) + " (the handle stays on two passes)");
        }
        HIP_TRY(hipMemset(L.dPlaneImg, 0, planeBytes));
        h->bytes += planeBytes;
      }
      if(!L.dFallback)
      {
        HIP_TRY(hipMalloc(&L.dFallback, listBytes));
        HIP_TRY(hipMemset(L.dFallback, 0, listBytes));
      }
    }
    if(!h->hFallback)
    {
      HIP_TRY(hipHostMalloc(&h->hFallback, sizeof(int) * 2 * kMaxLanes, hipHostMallocDefault));
      std::memset(h->hFallback, 0, sizeof(int) * 2 * kMaxLanes);
    }
    HIP_TRY(hipDeviceSynchronize());
  }
  h->singlePassMode = -1;
  h->singlePassBackoff = 0;
  return SSD_OK;
}

int ssd_set_debug(ssd_handle *h, int enable)
{
  if(!h)
    return fail(SSD_E_ARG, "ssd_set_debug: null handle");
  HIP_TRY(hipSetDevice(h->device));
  const bool wantImages = enable != 0 && enable != 2;
  if(enable && (!h->dDebug || (wantImages && !h->dDebugImg)))
  {
    /* the records always; the image buffer (F x (images + 1) x 2 whole images) only for capture WITH images — and then both
     * or neither: a failed second allocation must not leave the first behind as "debug is set up" */
    const size_t imgBytes = static_cast<size_t>(h->F) * (h->P.maxStepImages + 1) * 2 * h->imgWords * 8;
    DebugFrame *d = h->dDebug;
    unsigned long long *di = h->dDebugImg;
    if(!d)
      HIP_TRY(hipMalloc(&d, sizeof(DebugFrame) * h->F));
    if(wantImages && !di)
    {
      const hipError_t e = hipMalloc(&di, imgBytes);
      if(e != hipSuccess)
      {
        if(!h->dDebug)
          (void)hipFree(d);
        return fail(e == hipErrorOutOfMemory ? SSD_E_NOMEM : SSD_E_HIP, std::string("ssd_set_debug: ") + hipGetErrorString(e));
      }
      h->bytes += imgBytes;
    }
    if(!h->dDebug)
      h->bytes += sizeof(DebugFrame) * h->F;
    h->dDebug = d;
    h->dDebugImg = di;
  }
  h->debug = enable == 0 ? 0 : enable == 2 ? 2 : 1;
  return SSD_OK;
}

int ssd_set_risers(ssd_handle *h, int enable, double tolerance, int min_support)
{
  if(!h)
    return fail(SSD_E_ARG, "ssd_set_risers: null handle");
  if(enable && (!(tolerance > 0.0) || tolerance > 1.0 || min_support < 1))
    return fail(SSD_E_ARG, "ssd_set_risers: tolerance must be in (0, 1] m, min_support >= 1");
  HIP_TRY(hipSetDevice(h->device));
  if(enable && (!h->dRisers || !h->hRisers))
  {
    const size_t bytes = sizeof(ssd_frame_risers) * h->F;
    ssd_frame_risers *d = nullptr, *hh = nullptr;
    HIP_TRY(hipMalloc(&d, bytes));
    hipError_t e = hipMemset(d, 0, bytes);
    if(e == hipSuccess)
      e = hipHostMalloc(&hh, bytes, hipHostMallocDefault);
    if(e != hipSuccess)
    {
      (void)hipFree(d);
      return fail(e == hipErrorOutOfMemory ? SSD_E_NOMEM : SSD_E_HIP, std::string("ssd_set_risers: ") + hipGetErrorString(e));
    }
    h->dRisers = d;
    h->hRisers = hh;
    h->bytes += bytes;
  }
  h->P.risers = enable ? 1 : 0;
  if(enable)
  {
    h->P.riserTol = tolerance;
    h->P.riserMinSupport = min_support;
  }
  return SSD_OK;
}

int ssd_fetch_risers(ssd_handle *h, ssd_frame_risers *out, int nframes, void *stream)
{
  if(!h || !out)
    return fail(SSD_E_ARG, "ssd_fetch_risers: null argument");
  if(!h->P.risers || !h->dRisers)
    return fail(SSD_E_ARG, "ssd_fetch_risers: call ssd_set_risers(h, 1, ...) before the enqueue");
  HIP_TRY(hipSetDevice(h->device));
  if(h->hRisersBatchFrames > 0)
  {
    /* the last call was ssd_process_host / ssd_process_depth_host: its risers were collected slice by slice */
    if(nframes < 1 || nframes > h->hRisersBatchFrames)
      return fail(SSD_E_ARG, "ssd_fetch_risers: nframes exceeds what the last call processed");
    HIP_TRY(hipStreamSynchronize(h->ingestCompute));
    std::memcpy(out, h->hRisersBatch, sizeof(ssd_frame_risers) * nframes);
    return SSD_OK;
  }
  if(nframes < 1 || nframes > h->lastFrames)
    return fail(SSD_E_ARG, "ssd_fetch_risers: nframes exceeds what the last enqueue processed");
  /* the riser buffer is single: enqueues with risers on all run in lane 0, whose `done` event covers the last one */
  hipStream_t s = static_cast<hipStream_t>(stream);
  HIP_TRY(hipStreamWaitEvent(s, h->lane[h->lastLane].done, 0));
  HIP_TRY(hipMemcpyAsync(h->hRisers, h->dRisers, sizeof(ssd_frame_risers) * nframes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  std::memcpy(out, h->hRisers, sizeof(ssd_frame_risers) * nframes);
  return SSD_OK;
}

int ssd_set_timing(ssd_handle *h, int enable)
{
  if(!h)
    return fail(SSD_E_ARG, "ssd_set_timing: null handle");
  HIP_TRY(hipSetDevice(h->device));
  if(enable && h->ev.empty())
  {
    std::vector<hipEvent_t> ev;
    ev.reserve(static_cast<size_t>(SSD_TIMING_SLOTS) * 8);
    for(size_t i = 0; i < static_cast<size_t>(SSD_TIMING_SLOTS) * 8; i++)
    {
      hipEvent_t e;
      const hipError_t rc = hipEventCreate(&e);
      if(rc != hipSuccess)
      {
        for(hipEvent_t made : ev)
          (void)hipEventDestroy(made);
        return fail(SSD_E_HIP, std::string("hipEventCreate: ") + hipGetErrorString(rc));
      }
      ev.push_back(e);
    }
    /* k_predict's events before anything is handed to the handle: a failure leaves it as it was (no half-made set that a later
     * call would take for complete) */
    hipEvent_t pred[SSD_TIMING_SLOTS] = {};
    for(int i = 0; i < SSD_TIMING_SLOTS; i++)
      if(hipEventCreate(&pred[i]) != hipSuccess)
      {
        for(int k = 0; k < i; k++)
          (void)hipEventDestroy(pred[k]);
        for(hipEvent_t made : ev)
          (void)hipEventDestroy(made);
        return fail(SSD_E_HIP, "hipEventCreate (predict)");
      }
    for(int i = 0; i < SSD_TIMING_SLOTS; i++)
      h->evPredict[i] = pred[i];
    h->ev.swap(ev);
  }
  h->timing = enable != 0;
  h->timedFrom = h->enqueueCount;
  return SSD_OK;
}

static int choose_chunk(const ssd_tuning &tune, int nPoints, int nframes)
{
  const int forced = tune.chunkPoints;
  if(forced > 0)
  {
    const int c = ((forced + kTileHost - 1) / kTileHost) * kTileHost;
    return c > kMaxTilesPerBlockHost * kTileHost ? kMaxTilesPerBlockHost * kTileHost : c;
  }
  const int target = tune.targetBlocks;
  int cpf = (target + nframes - 1) / nframes;
  const int maxCpf = (nPoints + kTileHost - 1) / kTileHost;
  if(cpf > maxCpf) cpf = maxCpf;
  if(cpf < 1) cpf = 1;
  const int per = (nPoints + cpf - 1) / cpf;
  int chunk = ((per + kTileHost - 1) / kTileHost) * kTileHost;
  /* a block should stream at least 32 tiles (its prologue copies the frame's tables, its epilogue flushes the LDS
   * histogram / windows): measured +0.5 % on the XGA batch, +6.5 % on the FHD stress batch (tools/target_blocks.sh) —
   * as long as that leaves the machine at least 8192 blocks (small batches, single frames: latency first) */
  long long minChunk = static_cast<long long>(kMaxTilesPerBlockHost) * kTileHost;
  const long long total = static_cast<long long>(nPoints) * nframes;
  if(total / minChunk < 8192)
    minChunk = total / 8192 / kTileHost * kTileHost;
  if(chunk < minChunk)
    chunk = static_cast<int>(minChunk);
  if(chunk > kMaxTilesPerBlockHost * kTileHost)
    chunk = kMaxTilesPerBlockHost * kTileHost;
  return chunk;
}

static constexpr int kDirectResultFrames = 64;
static int enqueue_impl(ssd_handle *h, const void *d_xyz, size_t frame_stride_bytes, int nframes, void *stream, int stages, bool depthInput)
{
  if(!h || !d_xyz)
    return fail(SSD_E_ARG, "ssd_enqueue: null argument");
  if(nframes < 1 || nframes > h->F)
    return fail(SSD_E_ARG, "ssd_enqueue: nframes must be 1..max_frames_per_batch");
  const size_t frameBytes = static_cast<size_t>(h->P.nPoints) * (depthInput ? 2 : 12);
  if(depthInput)
  {
    if(!h->haveIntr)
      return fail(SSD_E_ARG, "ssd_enqueue_depth: call ssd_set_intrinsics first");
    if(frame_stride_bytes < frameBytes || frame_stride_bytes % 8 != 0 || (reinterpret_cast<uintptr_t>(d_xyz) & 7u) != 0 || h->P.W % 4 != 0)
      return fail(SSD_E_ARG, "ssd_enqueue_depth: frames must be 8-byte aligned, stride a multiple of 8, width a multiple of 4");
  }
  else
  {
    if(frame_stride_bytes < frameBytes || frame_stride_bytes % 4 != 0)
      return fail(SSD_E_ARG, "ssd_enqueue: frame stride smaller than a frame or not a multiple of 4");
    if((reinterpret_cast<uintptr_t>(d_xyz) & 3u) != 0)
      return fail(SSD_E_ARG, "ssd_enqueue: frame pointer must be 4-byte aligned");
  }
  HIP_TRY(hipSetDevice(h->device));
  Params P = h->P;
  P.px.groundFull = h->debug == 1 ? 1 : 0;     /* image capture compares the whole ground image; otherwise only what k_final reads is rastered */
  /* Which workspace.  A handle with one runs on the caller's stream (a switch of streams is ordered by the lane's event: the
   * workspace is single-buffered).  With several, successive batches take them in turn, each on the lane's own stream behind
   * an event recorded on the caller's stream now: the batch starts after the work the caller's stream holds, but the caller's
   * stream does not wait for the batch (ssd_fetch* / ssd_stream_wait do).  Debug capture, the riser pass and partial runs
   * (ssd_enqueue_stages) own single buffers / leave state behind for the next call: they stay in lane 0. */
  const bool pinned = h->debug != 0 || P.risers || stages != SSD_STAGE_ALL;
  const int li = h->depth == 1 || pinned ? 0 : static_cast<int>(h->laneTurn++ % static_cast<unsigned long long>(h->depth));
  ssd_lane &L = h->lane[li];
  hipStream_t s = static_cast<hipStream_t>(stream);
  if(h->depth == 1)
  {
    if(L.haveLast && s != L.lastStream)
      HIP_TRY(hipStreamWaitEvent(s, L.done, 0));
  }
  else
  {
    HIP_TRY(hipEventRecord(L.in, s));
    s = L.stream;
    HIP_TRY(hipStreamWaitEvent(s, L.in, 0));
    /* the single debug / riser buffers: a pinned call after free-running ones (or the other way round) must not overtake
     * what the other lanes still hold — rare (a switch of mode), so simply wait for all of them */
    if(pinned != h->lastPinned)
      for(int k = 0; k < h->depth; k++)
        if(k != li && h->lane[k].haveLast)
          HIP_TRY(hipStreamWaitEvent(s, h->lane[k].done, 0));
  }
  h->lastPinned = pinned;
  const float *xyz = static_cast<const float *>(d_xyz);
  const size_t strideFloats = depthInput ? frame_stride_bytes / 2 : frame_stride_bytes / 4;    /* elements of the source type */
  const DepthSrc depthSrc{ h->dDepthMaps, h->dDepthMaps ? h->dDepthMaps + P.W : nullptr, h->intr.depth_units, P.W, P.H, depth_row_magic(P.W, P.H) };
  const DepthSrc *depth = depthInput ? &depthSrc : nullptr;
  const int chunk = choose_chunk(h->tune, P.nPoints, nframes);
  /* K2 and K4 walk cell columns: the taller a block's chunk, the fewer window flushes and block starts per cell (below) */
  /* K1 ends every block with up to 121 global atomics into the frame's histogram: with one-tile chunks (single frames) 768
   * blocks queue up on the same addresses; at most ssd_tuning::k1BlocksPerFrame blocks per frame keeps that short */
  int chunkHist = chunk;
  {
    const int perFrame = h->tune.k1BlocksPerFrame;
    const int tilesPerFrame = (P.nPoints + kTileHost - 1) / kTileHost;
    int t = (tilesPerFrame + perFrame - 1) / perFrame;
    if(t > kMaxTilesPerBlockHost) t = kMaxTilesPerBlockHost;
    if(t * kTileHost > chunkHist) chunkHist = t * kTileHost;
  }
  int chunkRaster = chunk, chunkInquad = chunk;
  {
    /* Blocks of K2 / K4 by batch size (XGA, tools/exp_frames.sh, 1-128 frames swept): what a block pays per start (state,
     * LDS, cell list, window flush, final atomics) wants tall chunks, the 2048 block slots of the chip want >= ~1536 blocks,
     * and a grid of one to two rounds of slots wants more, smaller blocks for its tail.  Rule: chunks of more than 8 tiles
     * only while they leave >= 6144 blocks (three rounds); else as tall as leaves >= 1536 blocks, K2 never below 2 tiles.
     * Against the former ">= 600 blocks": 8 / 16 / 32 / 64 frames 46 -> 49 k, 73 -> 82 k, 113 -> 124 k, 159 -> 172 k frames/s. */
    const long long totalTiles = (static_cast<long long>(P.nPoints) + kTileHost - 1) / kTileHost * nframes;
    const int minBlocks = h->tune.k24MinBlocks, tallBlocks = h->tune.k24TallBlocks;
    int t2 = h->tune.k2ChunkTiles, t4 = h->tune.k4ChunkTiles;
    if(t2 > kMaxTilesPerBlockRasterHost) t2 = kMaxTilesPerBlockRasterHost;
    if(t4 > kMaxTilesPerBlockInquadHost) t4 = kMaxTilesPerBlockInquadHost;
    auto pick = [&](int t, int tMin)
    {
      while(t > 8 && totalTiles / t < tallBlocks) t /= 2;
      while(t > tMin && totalTiles / t < minBlocks) t /= 2;
      return t;
    };
    t2 = pick(t2, 2);
    t4 = pick(t4, 1);
    if(t2 * kTileHost > chunkRaster) chunkRaster = t2 * kTileHost;
    if(t4 * kTileHost > chunkInquad) chunkInquad = t4 * kTileHost;
  }
  DebugFrame *dbg = h->debug ? h->dDebug : nullptr;
  unsigned long long *dbgImg = h->debug == 1 ? h->dDebugImg : nullptr;
  /* The single pass: K1 rasters the step plateaus while it counts, into planes of the height bins k_predict expects them in;
   * k_peaks checks that against the complete histogram, frame by frame, and k_raster does the frames it does not cover.  Only
   * the whole pipeline in one call (the planes are set in the first stage and consumed in the fourth), only batches. */
  unsigned long long *planeImg = nullptr;
  /* (not on 16-bit depth input: K1 is bound by instruction issue there, not by the bytes a second pass would re-read - k_raster
   * reads a sixth of them -, and the raster's instructions cost K1 what k_raster took: 384.6 k against 386.8 k frames/s) */
  if(L.dPlaneImg && stages == SSD_STAGE_ALL && h->singlePassMode != 0 && (h->singlePassMode == 1 || (single_pass_batch(nframes, P.nPoints) && !depthInput)))
  {
    if(h->singlePassMode == 1 || h->singlePassBackoff == 0)
      planeImg = L.dPlaneImg;
    else
      h->singlePassBackoff--;                            /* see ssd_fetch_back */
  }
  /* the optional device buffers the kernels dereference without a test of their own: a handle that lacks one
   * fails here, loudly, instead of faulting on the device at a null address */
  if(planeImg && !L.dFallback)
    return fail(SSD_E_HIP, "ssd_enqueue: the single pass's planes without their work list (internal)");
  if(depthInput && (!depthSrc.xmap || !depthSrc.ymap))
    return fail(SSD_E_HIP, "ssd_enqueue_depth: the deprojection maps are missing (ssd_set_intrinsics did not complete)");
  if(!L.dState || !L.dStepImg || !L.dGroundImg || !L.dTileMasks)
    return fail(SSD_E_HIP, "ssd_enqueue: the handle's workspace is incomplete (internal)");
  h->lastSinglePass = planeImg != nullptr;
  h->predictTimed[h->enqueueCount % SSD_TIMING_SLOTS] = planeImg != nullptr;
  const bool timing = h->timing && !h->ev.empty();
  const int timingSlot = static_cast<int>(h->enqueueCount % SSD_TIMING_SLOTS);
  int evi = timingSlot * 8;

  const int slot = static_cast<int>(h->finalCount % static_cast<unsigned long long>(h->nSlots));
  const bool direct = nframes <= kDirectResultFrames && h->hResultsDev != nullptr;
  if((stages & SSD_STAGE_FINAL) && h->depth > 1 && h->resultsLane[slot] != li && h->lane[h->resultsLane[slot]].haveLast)
    HIP_TRY(hipStreamWaitEvent(s, h->lane[h->resultsLane[slot]].done, 0));     /* the slot's previous writer was another lane */

  /* the kernels of the chosen stages on stream cs, in order */
  auto chain = [&](hipStream_t cs, bool marks)
  {
    auto mk = [&]() { if(marks) (void)hipEventRecord(h->ev[evi++], cs); };
    if(planeImg)
    {
      /* k_predict in front of the seven stages, timed by itself (ssd_get_predict_time_back) */
      if(marks) (void)hipEventRecord(h->evPredict[timingSlot], cs);
      launch_predict(xyz, strideFloats, P, L.dState, nframes, depth, L.dFallback, h->planePool, h->singlePassSabotage, cs);
    }
    mk();
    if(stages & SSD_STAGE_HIST)
      launch_hist(xyz, strideFloats, P, L.dState, L.dTileMasks, h->tileMaskStride, nframes, chunkHist, depth, planeImg, cs);
    mk();
    if(stages & SSD_STAGE_PEAKS)
      launch_peaks(P, L.dState, nframes, dbg, planeImg ? L.dFallback : nullptr, cs);
    mk();
    if(stages & SSD_STAGE_RASTER)
      launch_raster(xyz, strideFloats, P, L.dState, L.dStepImg, L.dTileMasks, h->tileMaskStride, nframes, chunkRaster, depth, planeImg ? L.dFallback : nullptr, cs);
    mk();
    if(stages & SSD_STAGE_OUTLINE)
      launch_outline(P, L.dState, L.dStepImg, planeImg, nframes, dbg, dbgImg, cs);
    mk();
    if(stages & SSD_STAGE_QUADS)
      launch_quads(P, L.dState, nframes, dbg, cs);
    mk();
    if(stages & SSD_STAGE_INQUAD)
      launch_inquad(xyz, strideFloats, P, L.dState, L.dGroundImg, L.dTileMasks, h->tileMaskStride, nframes, chunkInquad, depth, cs);
    mk();
    /* The results leave with the batch, into this enqueue's pinned slot (event for ssd_fetch / ssd_fetch_back).  A few frames:
     * k_final stores them there itself — a kilobyte per frame of posted writes, visible to the host once the event has
     * fired — instead of a device-to-host copy command behind the kernel (single frame: one command less in the chain).
     * Batches go through device memory and one copy: a megabyte of scattered stores over PCIe would hold k_final's blocks. */
    if(stages & SSD_STAGE_FINAL)
    {
      ssd_frame_result *out = (direct ? h->hResultsDev : h->dResults) + static_cast<size_t>(slot) * h->F;
      launch_final(P, L.dState, L.dGroundImg, out, nframes, dbg, dbgImg, cs);
      if(P.risers)
        launch_risers(xyz, strideFloats, P, L.dState, L.dTileMasks, h->tileMaskStride, h->dRisers, nframes, chunk, depth, cs);
    }
    mk();
  };

  /* (A call of a few frames replayed from a captured HIP graph — one submission instead of seven launches — was measured in
   * round 3 and is slower on ROCm 7.2: one resident XGA frame 100 -> 200 us per ssd_enqueue + ssd_fetch, eight frames 134 ->
   * 158 us.  The code left the product in round 4; git history of this file has it.) */
  {
    /* bits a partial run (ssd_enqueue_stages) left behind: wiped before the stage that would raster on top of them — not
     * before a call that only CONSUMES them (the continuation of that partial run: k_outline / k_final read and clear) */
    if(L.stepImagesDirty && (stages & SSD_STAGE_RASTER))
    {
      HIP_TRY(hipMemsetAsync(L.dStepImg, 0, static_cast<size_t>(h->F) * P.maxStepImages * h->imgWords * 8, s));
      L.stepImagesDirty = false;
    }
    if(L.groundImageDirty && (stages & SSD_STAGE_INQUAD))
    {
      HIP_TRY(hipMemsetAsync(L.dGroundImg, 0, static_cast<size_t>(h->F) * h->imgWords * 8, s));
      L.groundImageDirty = false;
    }
    if(stages & SSD_STAGE_HIST)
    {
      /* No memset of the state in front of a call: K1 only needs its accumulators zero, and k_peaks — their one reader —
       * clears them as it takes them (everything else in FrameState is written before it is read).  Only after a call that
       * ran K1 without k_peaks (ssd_enqueue_stages), or one that failed half way, the state is zeroed here. */
      if(L.dirtyFrames > 0)
        HIP_TRY(hipMemsetAsync(L.dState, 0, sizeof(FrameState) * L.dirtyFrames, s));
      if(dbg)
        HIP_TRY(hipMemsetAsync(dbg, 0, sizeof(DebugFrame) * nframes, s));
    }
    chain(s, timing);
  }
  if(stages & SSD_STAGE_HIST)
    L.dirtyFrames = nframes;
  if(stages & SSD_STAGE_FINAL)
  {
    if(!direct)
      HIP_TRY(hipMemcpyAsync(h->hResults + static_cast<size_t>(slot) * h->F, h->dResults + static_cast<size_t>(slot) * h->F,
                             sizeof(ssd_frame_result) * nframes, hipMemcpyDeviceToHost, s));
    /* single pass: how many frames k_raster had to do travels with the results (ssd_fetch_back adapts to it) */
    h->resultsFallback[slot] = -1;
    if(planeImg && h->hFallback)
    {
      HIP_TRY(hipMemcpyAsync(h->hFallback + 2 * slot, L.dFallback, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
      h->resultsFallback[slot] = 0;
    }
    HIP_TRY(hipEventRecord(h->resultsReady[slot], s));
    h->resultsFrames[slot] = nframes;
    h->resultsLane[slot] = li;
    h->finalCount++;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(L.done, s));
  if((stages & SSD_STAGE_PEAKS) && nframes >= L.dirtyFrames)
    L.dirtyFrames = 0;
  L.lastStream = s;
  L.haveLast = true;
  /* a raster without its consumer leaves bits behind; the consumer clears what it read */
  if(stages & SSD_STAGE_RASTER) L.stepImagesDirty = true;
  if(stages & SSD_STAGE_OUTLINE) L.stepImagesDirty = false;
  if(stages & SSD_STAGE_INQUAD) L.groundImageDirty = true;
  if(stages & SSD_STAGE_FINAL) L.groundImageDirty = false;
  h->lastFrames = nframes;
  h->lastLane = li;
  h->hRisersBatchFrames = 0;
  h->enqueueCount++;
  return SSD_OK;
}

int ssd_enqueue_stages(ssd_handle *h, const void *d_xyz, size_t frame_stride_bytes, int nframes, void *stream, int stages)
{
  return enqueue_impl(h, d_xyz, frame_stride_bytes, nframes, stream, stages, false);
}

int ssd_enqueue(ssd_handle *h, const void *d_xyz, size_t frame_stride_bytes, int nframes, void *stream)
{
  return enqueue_impl(h, d_xyz, frame_stride_bytes, nframes, stream, SSD_STAGE_ALL, false);
}

int ssd_enqueue_depth(ssd_handle *h, const void *d_depth, size_t frame_stride_bytes, int nframes, void *stream)
{
  return enqueue_impl(h, d_depth, frame_stride_bytes, nframes, stream, SSD_STAGE_ALL, true);
}

/* rs2::pointcloud's maps (librealsense2 src/proc/pointcloud.cpp, pre_compute_x_y_map), float arithmetic */
static void depth_maps(const ssd_intrinsics &in, int W, int H, std::vector<float> &maps)
{
  maps.resize(static_cast<size_t>(W) + H);
  for(int u = 0; u < W; u++)
    maps[u] = (static_cast<float>(u) - in.ppx) / in.fx;
  for(int v = 0; v < H; v++)
    maps[W + v] = (static_cast<float>(v) - in.ppy) / in.fy;
}

int ssd_set_intrinsics(ssd_handle *h, const ssd_intrinsics *intr)
{
  if(!h || !intr || !(intr->fx != 0.0f) || !(intr->fy != 0.0f) || !(intr->depth_units > 0.0f))
    return fail(SSD_E_ARG, "ssd_set_intrinsics: bad argument");
  HIP_TRY(hipSetDevice(h->device));
  std::vector<float> maps;
  depth_maps(*intr, h->P.W, h->P.H, maps);
  if(!h->dDepthMaps)
    HIP_TRY(hipMalloc(&h->dDepthMaps, maps.size() * 4));
  /* depth batches still running (on the lanes' non-blocking streams, which a null-stream copy does not wait for) read the maps
   * in K1, K2 and K4: let every lane finish before they change */
  for(int k = 0; k < h->depth; k++)
    if(h->lane[k].haveLast)
      HIP_TRY(hipEventSynchronize(h->lane[k].done));
  HIP_TRY(hipMemcpy(h->dDepthMaps, maps.data(), maps.size() * 4, hipMemcpyHostToDevice));
  h->intr = *intr;
  h->haveIntr = true;
  return SSD_OK;
}

int ssd_deproject_host(const ssd_intrinsics *intr, int width, int height, const uint16_t *depth, float *xyz)
{
  if(!intr || !depth || !xyz || width <= 0 || height <= 0)
    return fail(SSD_E_ARG, "ssd_deproject_host: bad argument");
  std::vector<float> maps;
  depth_maps(*intr, width, height, maps);
  for(int v = 0; v < height; v++)
    for(int u = 0; u < width; u++)
    {
      const size_t i = static_cast<size_t>(v) * width + u;
      const float d = static_cast<float>(depth[i]) * intr->depth_units;
      xyz[3 * i] = d * maps[u];
      xyz[3 * i + 1] = d * maps[width + v];
      xyz[3 * i + 2] = d;
    }
  return SSD_OK;
}

/* milliseconds of the 7 stages of a timed enqueue; `back` = 0 is the last one, 1 the one before, ...
 * (at most SSD_TIMING_SLOTS - 1 back, and not before timing was switched on); waits for that enqueue */
int ssd_get_stage_times_back(ssd_handle *h, int back, float ms[7])
{
  if(!h || !ms)
    return fail(SSD_E_ARG, "ssd_get_stage_times: null");
  if(!h->timing || h->ev.empty() || back < 0 || back >= SSD_TIMING_SLOTS ||
     h->enqueueCount < static_cast<unsigned long long>(back) + 1 || h->enqueueCount - 1 - back < h->timedFrom)
    return fail(SSD_E_ARG, "ssd_get_stage_times: no timed enqueue at that position");
  HIP_TRY(hipSetDevice(h->device));
  const int base = static_cast<int>((h->enqueueCount - 1 - back) % SSD_TIMING_SLOTS) * 8;
  HIP_TRY(hipEventSynchronize(h->ev[base + 7]));
  for(int i = 0; i < 7; i++)
    HIP_TRY(hipEventElapsedTime(&ms[i], h->ev[base + i], h->ev[base + i + 1]));
  return SSD_OK;
}

/* k_predict's time, the kernel in front of the seven stages of a single-pass batch (0 for an enqueue that did not run it) */
int ssd_get_predict_time_back(ssd_handle *h, int back, float *ms)
{
  if(!h || !ms)
    return fail(SSD_E_ARG, "ssd_get_predict_time_back: null");
  if(!h->timing || h->ev.empty() || back < 0 || back >= SSD_TIMING_SLOTS ||
     h->enqueueCount < static_cast<unsigned long long>(back) + 1 || h->enqueueCount - 1 - back < h->timedFrom)
    return fail(SSD_E_ARG, "ssd_get_predict_time_back: no timed enqueue at that position");
  HIP_TRY(hipSetDevice(h->device));
  const int slot = static_cast<int>((h->enqueueCount - 1 - back) % SSD_TIMING_SLOTS);
  *ms = 0.0f;
  if(!h->predictTimed[slot])
    return SSD_OK;
  HIP_TRY(hipEventSynchronize(h->ev[slot * 8 + 7]));
  HIP_TRY(hipEventElapsedTime(ms, h->evPredict[slot], h->ev[slot * 8]));
  return SSD_OK;
}

int ssd_get_stage_times(ssd_handle *h, float ms[7])
{
  return ssd_get_stage_times_back(h, 0, ms);
}

int ssd_fetch_back(ssd_handle *h, ssd_frame_result *results, int nframes, int back)
{
  if(!h || !results)
    return fail(SSD_E_ARG, "ssd_fetch: null argument");
  if(back < 0 || back >= h->nSlots || h->finalCount < static_cast<unsigned long long>(back) + 1)
    return fail(SSD_E_ARG, "ssd_fetch: no enqueue at that position (back = 0: the last one, 1: the one before, .. < max(2, batches_in_flight))");
  const int slot = static_cast<int>((h->finalCount - 1 - back) % static_cast<unsigned long long>(h->nSlots));
  if(nframes < 1 || nframes > h->resultsFrames[slot])
    return fail(SSD_E_ARG, "ssd_fetch: nframes exceeds what that enqueue processed");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipEventSynchronize(h->resultsReady[slot]));
  std::memcpy(results, h->hResults + static_cast<size_t>(slot) * h->F, sizeof(ssd_frame_result) * nframes);
  /* The single pass pays when there are step plateaus to raster and its predictor covers them.  A batch of which k_raster had to
   * do more than half of the frames (scenes without a sharp peak per tread, or far more treads than planes), or of which fewer
   * than a quarter of the frames had a step plateau at all (no stairs in sight: k_predict and K1's idle raster code cost 7 % and
   * k_raster would have had nothing to do anyway), says the input is of the other kind: the next kSinglePassBackoff qualifying
   * batches run two passes, then one batch probes again.  Results do not depend on it. */
  if(h->resultsFallback[slot] == 0)
  {
    h->resultsFallback[slot] = 1;                       /* counted once */
    const int listed = h->hFallback[2 * slot], frames = h->resultsFrames[slot], withSteps = frames - h->hFallback[2 * slot + 1];
    if(2 * listed > frames || 4 * withSteps < frames)
      h->singlePassBackoff = kSinglePassBackoff;
  }
  return SSD_OK;
}

int ssd_stream_wait(ssd_handle *h, int back, void *stream)
{
  if(!h)
    return fail(SSD_E_ARG, "ssd_stream_wait: null handle");
  if(back < 0 || back >= h->nSlots || h->finalCount < static_cast<unsigned long long>(back) + 1)
    return fail(SSD_E_ARG, "ssd_stream_wait: no enqueue at that position");
  const int slot = static_cast<int>((h->finalCount - 1 - back) % static_cast<unsigned long long>(h->nSlots));
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), h->resultsReady[slot], 0));
  return SSD_OK;
}

int ssd_batches_in_flight(const ssd_handle *h)
{
  return h ? h->depth : 0;
}

int ssd_fetch(ssd_handle *h, ssd_frame_result *results, int nframes, void *stream)
{
  (void)stream;                  /* the copy was enqueued with the batch; its event is what is waited for */
  return ssd_fetch_back(h, results, nframes, 0);
}

/* ---- frames in host memory: double-buffered ingest ----------------------------------------------------------------
 * A camera-fed product is host-fed: frames arrive in host memory and cross PCIe once (9.4 MB per XGA vertex frame,
 * 1.6 MB as 16-bit depth).  The batch is cut into slices of at most kIngestFrames frames; slice c + 1 is copied into the
 * other of two device staging buffers on a copy stream while the kernels of slice c run on a compute stream, and the
 * results of slice c - 1 are read (they travelled with their enqueue).  Source memory obtained from ssd_host_alloc is
 * pinned: its copies are true DMA; pageable memory goes through the runtime's own staging (the call then blocks in
 * the copy while the GPU computes the previous slice). */
static const int kIngestFrames = 32;

static int ingest_prepare(ssd_handle *h, size_t sliceBytes)
{
  if(!h->ingestCopy)
  {
    HIP_TRY(hipStreamCreateWithFlags(&h->ingestCopy, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&h->ingestCopy2, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&h->ingestCompute, hipStreamNonBlocking));
    for(int k = 0; k < 2; k++)
    {
      HIP_TRY(hipEventCreateWithFlags(&h->ingestCopied[k], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&h->ingestCopied2[k], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&h->ingestConsumed[k], hipEventDisableTiming));
    }
  }
  if(h->ingestCap < sliceBytes)
  {
    HIP_TRY(hipDeviceSynchronize());
    for(int k = 0; k < 2; k++)
    {
      if(h->ingestBuf[k]) (void)hipFree(h->ingestBuf[k]);
      h->ingestBuf[k] = nullptr;
    }
    h->ingestCap = 0;
    HIP_TRY(hipMalloc(&h->ingestBuf[0], sliceBytes));
    HIP_TRY(hipMalloc(&h->ingestBuf[1], sliceBytes));
    h->ingestCap = sliceBytes;
  }
  return SSD_OK;
}

static int process_host_impl(ssd_handle *h, const void *src, size_t srcFrameBytes, size_t devFrameBytes, int nframes,
                             ssd_frame_result *results, bool depthInput)
{
  HIP_TRY(hipSetDevice(h->device));
  const int slice = h->F < kIngestFrames ? h->F : kIngestFrames;
  int rc = ingest_prepare(h, static_cast<size_t>(slice) * devFrameBytes);
  if(rc) return rc;
  const bool risers = h->P.risers && h->dRisers;
  if(risers && h->hRisersBatchCap < nframes)
  {
    /* the risers of the whole batch, slice by slice (the device buffer holds one enqueue's) */
    HIP_TRY(hipStreamSynchronize(h->ingestCompute));
    if(h->hRisersBatch) (void)hipHostFree(h->hRisersBatch);
    h->hRisersBatch = nullptr;
    h->hRisersBatchCap = 0;
    HIP_TRY(hipHostMalloc(&h->hRisersBatch, sizeof(ssd_frame_risers) * static_cast<size_t>(nframes), hipHostMallocDefault));
    h->hRisersBatchCap = nframes;
  }
  const unsigned char *from = static_cast<const unsigned char *>(src);
  int prevFrames = 0, prevAt = 0, c = 0;
  for(int done = 0; done < nframes; c++)
  {
    const int n = nframes - done < slice ? nframes - done : slice;
    const int k = c & 1;
    if(c >= 2)
      HIP_TRY(hipStreamWaitEvent(h->ingestCopy, h->ingestConsumed[k], 0));        /* the kernels of slice c - 2 read this buffer */
    /* A slice of 16 MB and more goes over in two halves on two copy streams: one copy of a pinned source runs on ONE copy engine, at 42 GB/s
     * on the boxes of rounds 5 and 6, while the runtime's own staging of a pageable source reached 55 (bench.py: host_fed, both
     * orders, two warm-up calls) - the pinned source was the slower one. */
    bool split = false;
    if(srcFrameBytes == devFrameBytes)
    {
      const size_t bytes = static_cast<size_t>(n) * srcFrameBytes, half = (bytes / 2) & ~static_cast<size_t>(4095);
      split = half >= (static_cast<size_t>(8) << 20);
      const unsigned char *at = from + static_cast<size_t>(done) * srcFrameBytes;
      if(split)
      {
        if(c >= 2)
          HIP_TRY(hipStreamWaitEvent(h->ingestCopy2, h->ingestConsumed[k], 0));
        HIP_TRY(hipMemcpyAsync(static_cast<unsigned char *>(h->ingestBuf[k]) + half, at + half, bytes - half, hipMemcpyHostToDevice, h->ingestCopy2));
        HIP_TRY(hipEventRecord(h->ingestCopied2[k], h->ingestCopy2));
        HIP_TRY(hipStreamWaitEvent(h->ingestCompute, h->ingestCopied2[k], 0));
      }
      HIP_TRY(hipMemcpyAsync(h->ingestBuf[k], at, split ? half : bytes, hipMemcpyHostToDevice, h->ingestCopy));
    }
    else
      HIP_TRY(hipMemcpy2DAsync(h->ingestBuf[k], devFrameBytes, from + static_cast<size_t>(done) * srcFrameBytes, srcFrameBytes, srcFrameBytes, n,
                               hipMemcpyHostToDevice, h->ingestCopy));
    HIP_TRY(hipEventRecord(h->ingestCopied[k], h->ingestCopy));
    HIP_TRY(hipStreamWaitEvent(h->ingestCompute, h->ingestCopied[k], 0));
    rc = enqueue_impl(h, h->ingestBuf[k], devFrameBytes, n, h->ingestCompute, SSD_STAGE_ALL, depthInput);
    if(rc) return rc;
    /* "Consumed" is the end of the slice's kernels — on the stream they ran on (with several workspaces the lane's own; the
     * compute stream itself only orders a slice behind its copy, so the slices of a handle with several workspaces overlap
     * like any other batches: making the compute stream wait for every slice — round 3 — had serialised them). */
    if(risers)
    {
      /* the riser buffer is single (enqueues with risers on all run in the first workspace): its copy follows the slice */
      rc = ssd_stream_wait(h, 0, h->ingestCompute);
      if(rc) return rc;
      HIP_TRY(hipMemcpyAsync(h->hRisersBatch + done, h->dRisers, sizeof(ssd_frame_risers) * n, hipMemcpyDeviceToHost, h->ingestCompute));
      HIP_TRY(hipEventRecord(h->ingestConsumed[k], h->ingestCompute));
    }
    else
      HIP_TRY(hipEventRecord(h->ingestConsumed[k], h->lane[h->lastLane].lastStream));
    if(prevFrames)
    {
      rc = ssd_fetch_back(h, results + prevAt, prevFrames, 1);
      if(rc) return rc;
    }
    prevFrames = n;
    prevAt = done;
    done += n;
  }
  rc = ssd_fetch_back(h, results + prevAt, prevFrames, 0);
  if(rc) return rc;
  h->hRisersBatchFrames = risers ? nframes : 0;
  return SSD_OK;
}

int ssd_process_host(ssd_handle *h, const float *xyz, int nframes, ssd_frame_result *results)
{
  if(!h || !xyz || !results || nframes < 1)
    return fail(SSD_E_ARG, "ssd_process_host: bad argument");
  const size_t frameBytes = static_cast<size_t>(h->P.nPoints) * 12;
  return process_host_impl(h, xyz, frameBytes, frameBytes, nframes, results, false);
}

int ssd_process_depth_host(ssd_handle *h, const uint16_t *depth, int nframes, ssd_frame_result *results)
{
  if(!h || !depth || !results || nframes < 1)
    return fail(SSD_E_ARG, "ssd_process_depth_host: bad argument");
  if(!h->haveIntr)
    return fail(SSD_E_ARG, "ssd_process_depth_host: call ssd_set_intrinsics first");
  const size_t frameElems = (static_cast<size_t>(h->P.nPoints) + 3) / 4 * 4;       /* device stride kept a multiple of 8 bytes */
  return process_host_impl(h, depth, static_cast<size_t>(h->P.nPoints) * 2, frameElems * 2, nframes, results, true);
}

/* pinned host memory for frames (DMA without a staging copy) */
int ssd_host_alloc(size_t bytes, void **ptr)
{
  if(!ptr || bytes == 0)
    return fail(SSD_E_ARG, "ssd_host_alloc: bad argument");
  if(ssd_device_count() <= 0)
    return fail(SSD_E_NODEVICE, "ssd_host_alloc: no HIP device");
  *ptr = nullptr;
  const hipError_t e = hipHostMalloc(ptr, bytes, hipHostMallocDefault);
  if(e != hipSuccess)
    return fail(e == hipErrorOutOfMemory ? SSD_E_NOMEM : SSD_E_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
  return SSD_OK;
}

int ssd_host_free(void *ptr)
{
  if(ptr)
    HIP_TRY(hipHostFree(ptr));
  return SSD_OK;
}

/* Stairs::serialize(), stairs.cpp:34-70: ["stairs",["stairSteps",N],[[["height",h],["quadrilateral",[x,y] x4]],...]]
 * fixed notation, 3 decimals; the third element is omitted when N = 0. */
int ssd_serialize(const ssd_frame_result *r, char *buf, size_t cap)
{
  if(!r || !buf || cap == 0)
    return fail(SSD_E_ARG, "ssd_serialize: bad argument");
  if(r->status & SSD_ST_THROW)
  {
    buf[0] = 0;
    return 0;
  }
  std::string s;
  s.reserve(128 + 160 * static_cast<size_t>(r->n_steps > 0 ? r->n_steps : 0));
  /* fixed notation, 3 decimals, as `os << fixed << setprecision(3)` prints in the classic locale: std::to_chars is
   * locale-independent (a host that called setlocale(LC_NUMERIC, "de_DE") must not get "0,170") and writes every
   * digit of any magnitude (DBL_MAX has 309 integral digits) */
  char tmp[384];
  auto num = [&](double v)
  {
    const std::to_chars_result r = std::to_chars(tmp, tmp + sizeof(tmp), v, std::chars_format::fixed, 3);
    s.append(tmp, r.ptr);
  };
  s += "[\"stairs\",[\"stairSteps\",";
  s += std::to_string(r->n_steps);
  s += ']';
  if(r->n_steps > 0)
  {
    s += ",[";
    for(int i = 0; i < r->n_steps; i++)
    {
      const ssd_step &st = r->steps[i];
      if(i) s += ',';
      s += "[[\"height\",";
      num(st.height);
      s += "],[\"quadrilateral\",";
      for(int k = 0; k < 4; k++)
      {
        if(k) s += ',';
        s += '[';
        num(st.quad[2 * k]);
        s += ',';
        num(st.quad[2 * k + 1]);
        s += ']';
      }
      s += "]]";
    }
    s += ']';
  }
  s += ']';
  if(s.size() + 1 > cap)
    return fail(SSD_E_CAP, "ssd_serialize: buffer too small");
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return static_cast<int>(s.size());
}

int ssd_get_debug(ssd_handle *h, int frame, ssd_debug_frame *out)
{
  if(!h || !out)
    return fail(SSD_E_ARG, "ssd_get_debug: null");
  if(!h->dDebug || frame < 0 || frame >= h->lastFrames)
    return fail(SSD_E_ARG, "ssd_get_debug: debug capture off or frame out of range");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, &h->dDebug[frame].d, sizeof(ssd_debug_frame), hipMemcpyDeviceToHost));
  return SSD_OK;
}

int ssd_get_debug_image(ssd_handle *h, int frame, int step_slot, int closed, uint8_t *out)
{
  if(!h || !out)
    return fail(SSD_E_ARG, "ssd_get_debug_image: null");
  if(!h->dDebugImg || h->debug != 1 || frame < 0 || frame >= h->lastFrames || step_slot < -1 || step_slot >= h->P.maxStepImages)
    return fail(SSD_E_ARG, "ssd_get_debug_image: image capture off (ssd_set_debug(h, 1)) or index out of range");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  const int slot = step_slot < 0 ? h->P.maxStepImages : step_slot;
  const unsigned long long *src = h->dDebugImg +
    ((static_cast<size_t>(frame) * (h->P.maxStepImages + 1) + slot) * 2 + (closed ? 1 : 0)) * h->imgWords;
  std::vector<unsigned long long> words(h->imgWords);
  HIP_TRY(hipMemcpy(words.data(), src, h->imgWords * 8, hipMemcpyDeviceToHost));
  const int W = h->P.W, H = h->P.H, W64 = h->P.W64;
  for(int y = 0; y < H; y++)
    for(int x = 0; x < W; x++)
      out[static_cast<size_t>(y) * W + x] = ((words[static_cast<size_t>(y) * W64 + (x >> 6)] >> (x & 63)) & 1ull) ? 0xff : 0;
  return SSD_OK;
}

/* ---- plain device-memory helpers --------------------------------------------- */

int ssd_device_alloc(int device, size_t bytes, void **d_ptr)
{
  if(!d_ptr)
    return fail(SSD_E_ARG, "ssd_device_alloc: null");
  if(ssd_device_count() <= 0)
    return fail(SSD_E_NODEVICE, "ssd_device_alloc: no HIP device");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMalloc(d_ptr, bytes));
  return SSD_OK;
}
int ssd_device_free(int device, void *d_ptr)
{
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipFree(d_ptr));
  return SSD_OK;
}
int ssd_device_upload(int device, void *d_dst, const void *src, size_t bytes)
{
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
  return SSD_OK;
}
int ssd_device_download(int device, void *dst, const void *d_src, size_t bytes)
{
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
  return SSD_OK;
}
int ssd_device_sync(int device)
{
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipDeviceSynchronize());
  return SSD_OK;
}

extern "C++"
{
namespace
{
/* first line of a sysfs attribute, without the newline; empty when it cannot be read */
std::string sysfs_line(const std::string &path)
{
  std::ifstream f(path);
  std::string s;
  if(f)
    std::getline(f, s);
  return s;
}
/* "0-15,128-143" -> CPU numbers; anything malformed ends the list there */
std::vector<int> parse_cpu_list(const std::string &list)
{
  std::vector<int> cpus;
  const char *p = list.c_str();
  while(*p)
  {
    char *end = nullptr;
    const long a = std::strtol(p, &end, 10);
    if(end == p || a < 0)
      break;
    long b = a;
    p = end;
    if(*p == '-')
    {
      b = std::strtol(p + 1, &end, 10);
      if(end == p + 1 || b < a)
        break;
      p = end;
    }
    for(long c = a; c <= b && cpus.size() < 4096; c++)
      cpus.push_back(static_cast<int>(c));
    if(*p == ',')
      p++;
    else
      break;
  }
  return cpus;
}
} // namespace
} // extern "C++"

/* the device's report, and (full != nullptr) its local CPU list as sysfs spells it - the report's field holds 255 characters of it */
static int device_info_impl(int device, ssd_device_info *out, std::string *full)
{
  if(!out)
    return fail(SSD_E_ARG, "ssd_device_info_get: null");
  const int n = ssd_device_count();
  if(n <= 0)
    return fail(SSD_E_NODEVICE, "ssd_device_info_get: no HIP device");
  if(device < 0 || device >= n)
    return fail(SSD_E_ARG, "ssd_device_info_get: device index out of range");
  std::memset(out, 0, sizeof(*out));
  out->numa_node = -1;
  HIP_TRY(hipDeviceGetPCIBusId(out->pci_bus_id, static_cast<int>(sizeof(out->pci_bus_id)), device));
  for(char *c = out->pci_bus_id; *c; c++)
    if(*c >= 'A' && *c <= 'F')
      *c = static_cast<char>(*c - 'A' + 'a');                 /* sysfs spells bus ids in lower case */
  hipUUID uuid;
  hipDevice_t dev;
  if(hipDeviceGet(&dev, device) == hipSuccess && hipDeviceGetUuid(&uuid, dev) == hipSuccess)
  {
    static const char hex[] = "0123456789abcdef";
    for(int i = 0; i < 16; i++)
    {
      const unsigned char b = static_cast<unsigned char>(uuid.bytes[i]);
      out->uuid[2 * i] = hex[b >> 4];
      out->uuid[2 * i + 1] = hex[b & 15];
    }
  }
  else
    (void)hipGetLastError();
  const std::string dir = std::string("/sys/bus/pci/devices/") + out->pci_bus_id + "/";
  const std::string node = sysfs_line(dir + "numa_node");
  if(!node.empty())
    out->numa_node = std::atoi(node.c_str());
  const std::string cpus = sysfs_line(dir + "local_cpulist");
  std::snprintf(out->cpu_list, sizeof(out->cpu_list), "%s", cpus.c_str());
  out->n_local_cpus = static_cast<int32_t>(parse_cpu_list(cpus).size());
  if(full)
    *full = cpus;
  return SSD_OK;
}

int ssd_device_info_get(int device, ssd_device_info *out)
{
  return device_info_impl(device, out, nullptr);
}

int ssd_bind_thread_to_device(int device)
{
  ssd_device_info info;
  std::string list;               /* the whole sysfs line: the report's copy may end in the middle of a number ("128-143" cut to "12") */
  const int rc = device_info_impl(device, &info, &list);
  if(rc != SSD_OK)
    return rc;
  const std::vector<int> cpus = parse_cpu_list(list);
  if(cpus.empty())
    return 0;
  cpu_set_t *set = CPU_ALLOC(4096);
  if(!set)
    return fail(SSD_E_NOMEM, "ssd_bind_thread_to_device: CPU_ALLOC");
  const size_t bytes = CPU_ALLOC_SIZE(4096);
  /* only CPUs this thread may run on (a container's cpuset can be narrower than the node): an empty intersection changes nothing */
  cpu_set_t *allowed = CPU_ALLOC(4096);
  int bound = 0;
  if(allowed && sched_getaffinity(0, bytes, allowed) == 0)
  {
    CPU_ZERO_S(bytes, set);
    for(int c : cpus)
      if(c < 4096 && CPU_ISSET_S(c, bytes, allowed))
      {
        CPU_SET_S(c, bytes, set);
        bound++;
      }
    if(bound > 0 && sched_setaffinity(0, bytes, set) != 0)
      bound = 0;
  }
  if(allowed) CPU_FREE(allowed);
  CPU_FREE(set);
  return bound;
}

} // extern "C"
