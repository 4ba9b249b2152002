/*
 * ssd_testhooks.hip — libssd_testhooks.so: entry points that exist only so that tests can run pieces of the
 * kernels in isolation (std::hypot and std::sort as restated for the device, QuadrilateralTest as the kernels build
 * and evaluate it, a frame's raw device state).  C ABI in include/ssd_testhooks.h.  Not part of the product ABI.
 */
#include "ssd_handle.h"
#include "ssd_launch.h"
#include "ssd_predict.h"
#include "ssd_prexy.h"
#include "ssd_math.h"
#include "ssd_quadtest.h"
#include "ssd_closing.h"
#include "ssd_bestline.h"
#include "ssd_sort.h"
#include "../../include/ssd_testhooks.h"

#include <cstddef>
#include <cstring>
#include <string>
#include <vector>

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

/* test hook: gnu_sort on the device */
__global__ void k_sorttest(double *dist, int *idx, int n)
{
  if(blockIdx.x == 0 && threadIdx.x == 0)
  {
    for(int i = 0; i < n; i++)
      idx[i] = i;
    gnu_sort(SortKeys{ dist, idx }, n);
  }
}

/* test hook: k_inquad's "is this box of K1's grid wholly inside the quadrilateral?" (build_grid_segs + grid_box_inside) for one
 * quadrilateral and n boxes (x0, x1, y0, y1 in grid cells); usable = QuadGridSegs::ok (0 also when the test would throw) */
__global__ void k_gridboxes(const double *__restrict__ quad, double xMin, double yMin, double boxX, double boxY,
                            const int *__restrict__ boxes, int n, unsigned char *__restrict__ inside, int *__restrict__ usable)
{
  __shared__ QuadTest t;
  __shared__ QuadGridSegs sg;
  if(threadIdx.x == 0)
  {
    double q[8];
    for(int k = 0; k < 8; k++)
      q[k] = quad[k];
    QuadTest local;
    build_quad_test(q, local);
    t = local;
    build_grid_segs(t, xMin, yMin, boxX, boxY, sg);
    *usable = sg.ok;
  }
  __syncthreads();
  for(int i = threadIdx.x; i < n; i += blockDim.x)
    inside[i] = grid_box_inside(sg, boxes[4 * i], boxes[4 * i + 1], boxes[4 * i + 2], boxes[4 * i + 3]) ? 1 : 0;
}

/* test hook: QuadrilateralTest as the kernels build and evaluate it (build_quad_test + the constant cell + quad_test),
 * one quadrilateral, n points; err = the negative code of the reference's throw or 0 */
__global__ void k_quadtest(const double *__restrict__ quad, const double *__restrict__ pts, int n, unsigned char *__restrict__ inside,
                           int *__restrict__ err)
{
  __shared__ QuadTest t;
  if(threadIdx.x == 0)
  {
    double q[8];
    for(int k = 0; k < 8; k++)
      q[k] = quad[k];
    QuadTest local;
    build_quad_test(q, local);
    t = local;
    if(blockIdx.x == 0)
      *err = local.err;
  }
  __syncthreads();
  if(t.err)
    return;
  for(int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
  {
    const double x = pts[2 * i], y = pts[2 * i + 1];
    const bool fast = x >= t.fx0 && x < t.fx1 && y >= t.fy0 && y < t.fy1;
    inside[i] = (fast || quad_test(t, x, y)) ? 1 : 0;
  }
}

/* test hook: k_inquad's edge table as k_quads builds it (ssd_quadtest.h: quad_edges_coeffs, quad_cell_unsound dealt out to the lanes,
 * quad_edges_margin) for n quadrilaterals, one per block: out = gx[4], gy[4], g2[4], m per quadrilateral */
__global__ __launch_bounds__(64) void k_quadedges(const double *__restrict__ quads, int n, double xMin, double xMax, double yMin, double yMax,
                                                  double boxX, double boxY, float *__restrict__ out)
{
  __shared__ QuadTest t;
  __shared__ QuadEdgesD w;
  __shared__ unsigned int bad;
  __shared__ QuadEdgesF ef;
  const int i = blockIdx.x;
  if(i >= n)
    return;
  if(threadIdx.x == 0)
  {
    double q[8];
    for(int k = 0; k < 8; k++)
      q[k] = quads[8 * i + k];
    QuadTest local;
    build_quad_test(q, local);
    QuadGridSegs sg;
    build_grid_segs(local, xMin, yMin, boxX, boxY, sg);
    QuadEdgesD lw;
    QuadEdgesF lf;
    quad_edges_coeffs(local, sg.ok, xMin, xMax, yMin, yMax, lw, lf);
    t = local; w = lw; ef = lf; bad = 0u;
  }
  __syncthreads();
  if(threadIdx.x < 9 && w.fine != 0 && quad_cell_unsound(t, w, static_cast<int>(threadIdx.x) / 3, static_cast<int>(threadIdx.x) % 3))
    atomicOr(&bad, 1u);
  __syncthreads();
  if(threadIdx.x == 0)
  {
    for(int s = 0; s < 4; s++) { out[13 * i + s] = ef.gx[s]; out[13 * i + 4 + s] = ef.gy[s]; out[13 * i + 8 + s] = ef.g2[s]; }
    out[13 * i + 12] = quad_edges_margin(w, bad != 0u);
  }
}

/* test hook: hypot_ref on the device */
__global__ void k_hypot(const double *a, const double *b, double *out, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n)
    out[i] = hypot_ref(a[i], b[i]);
}

/* measurement hook: a plain read stream (tools/loadbench.hip, variant C: every wave reads 3 KiB contiguously as three 16-byte
 * loads per lane, 1024 "points" of 12 bytes per block iteration, chunk bytes per block) — what the memory system of this GPU
 * delivers right now, to put beside K1's time on the same buffer (bench.py: roofline.k1_over_plain_stream) */
__global__ __launch_bounds__(256) void k_stream_read(const float4 *__restrict__ p, size_t nVec, int chunkVec, float *out)
{
  const size_t begin = static_cast<size_t>(blockIdx.x) * chunkVec, end = begin + chunkVec < nVec ? begin + chunkVec : nVec;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float acc = 0.0f;
  for(size_t i0 = begin; i0 < end; i0 += 768)                 /* 768 float4 = 1024 points of 12 bytes */
  {
    const size_t q = i0 + 192 * wave;
    float4 a = { 0, 0, 0, 0 }, b = a, c = a;
    if(q + 192 <= end)
    {
      a = p[q + lane]; b = p[q + 64 + lane]; c = p[q + 128 + lane];
    }
    acc += a.x * 1.0001f + a.y * 0.5f + a.z + a.w * 1.0001f + b.x * 0.5f + b.y + b.z * 1.0001f + b.w * 0.5f + c.x + c.y * 1.0001f + c.z * 0.5f + c.w;
  }
  if(acc == 1234.5f)
    out[0] = acc;
}

} // namespace ssd

extern "C"
{

const char *ssd_testhooks_last_error(void)
{
  return g_err.c_str();
}


/* hypot as the kernels compute it: on the host (no GPU needed) and on the device */
double ssd_test_hypot_host(double a, double b)
{
  return hypot_ref(a, b);
}
long long ssd_test_frame_state(ssd_handle *h, int frame, void *out, size_t cap, long long layout[8])
{
  if(!h || !out || frame < 0 || frame >= h->F)
    return fail(SSD_E_ARG, "ssd_test_frame_state: bad argument");
  if(layout)
  {
    layout[0] = sizeof(FrameState);
    layout[1] = offsetof(FrameState, hist);
    layout[2] = offsetof(FrameState, lut);
    layout[3] = offsetof(FrameState, imgYMin);
    layout[4] = offsetof(FrameState, pl);
    layout[5] = offsetof(FrameState, qtLive);
    layout[6] = offsetof(FrameState, sumZ);
    layout[7] = offsetof(FrameState, cnt);
  }
  const size_t n = cap < sizeof(FrameState) ? cap : sizeof(FrameState);
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, h->lane[h->lastLane].dState + frame, n, hipMemcpyDeviceToHost));
  return static_cast<long long>(n);
}

int ssd_test_record_offset(ssd_handle *h, size_t offset_bytes)
{
  if(!h || offset_bytes % 8 != 0 || offset_bytes > h->lane[0].recordSlack)
    return fail(SSD_E_ARG, "ssd_test_record_offset: bad argument (the offset must lie within the extra bytes of ssd_test_record_realloc_sized)");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  h->lane[0].dTileMasks = h->lane[0].dTileMasksBase + offset_bytes / 8;
  return SSD_OK;
}

static std::vector<void *> g_recordKeep;
unsigned long long ssd_test_record_realloc_sized(ssd_handle *h, size_t extra_bytes, size_t offset_bytes);
unsigned long long ssd_test_record_realloc(ssd_handle *h)
{
  return ssd_test_record_realloc_sized(h, 0, 0);
}
unsigned long long ssd_test_record_realloc_sized(ssd_handle *h, size_t extra_bytes, size_t offset_bytes)
{
  if(!h || offset_bytes % 8 != 0 || offset_bytes > extra_bytes)
    return 0ull;
  if(hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
    return 0ull;
  const size_t bytes = h->recordBytes + extra_bytes;
  void *p = nullptr;
  if(hipMalloc(&p, bytes) != hipSuccess)
    return 0ull;
  g_recordKeep.push_back(h->lane[0].dTileMasksBase);        /* the old one stays allocated: the next hipMalloc cannot reuse its place */
  h->lane[0].dTileMasksBase = static_cast<uint2 *>(p);
  h->lane[0].recordSlack = extra_bytes;
  h->lane[0].dTileMasks = h->lane[0].dTileMasksBase + offset_bytes / 8;
  /* the other workspaces of the handle likewise (plain allocations) */
  for(int k = 1; k < h->depth; k++)
  {
    void *q = nullptr;
    if(hipMalloc(&q, h->recordBytes) != hipSuccess)
      return 0ull;
    g_recordKeep.push_back(h->lane[k].dTileMasksBase);
    h->lane[k].dTileMasksBase = static_cast<uint2 *>(q);
    h->lane[k].dTileMasks = h->lane[k].dTileMasksBase;
  }
  return reinterpret_cast<unsigned long long>(p) + offset_bytes;
}
int ssd_test_record_release(void)
{
  for(void *p : g_recordKeep)
    (void)hipFree(p);
  g_recordKeep.clear();
  return SSD_OK;
}

int ssd_test_empty_quadrilateral(ssd_handle *h, int frame, int surface)
{
  if(!h || frame < 0 || frame >= h->F || surface < -1 || surface >= kMaxPlateaus)
    return fail(SSD_E_ARG, "ssd_test_empty_quadrilateral: bad argument");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  FrameState *d = h->lane[h->lastLane].dState + frame;
  FrameState fs;
  HIP_TRY(hipMemcpy(&fs, d, sizeof(fs), hipMemcpyDeviceToHost));
  const int acc = surface < 0 ? kGroundAcc : surface;
  if(surface < 0)
  {
    fs.sumZ[acc] = 0;
    fs.cnt[acc] = 0u;
  }
  else
  {
    const int slot = surface - fs.firstStep;
    if(surface >= fs.nPlateaus || slot < 0 || slot >= fs.nStepImages)
      return fail(SSD_E_ARG, "ssd_test_empty_quadrilateral: not a step plateau of this frame");
    fs.sumZ[acc] = fs.totZ[slot];
    fs.cnt[acc] = static_cast<unsigned int>(fs.pl[surface].nPoints);
  }
  HIP_TRY(hipMemcpy(&d->sumZ[acc], &fs.sumZ[acc], sizeof(fs.sumZ[acc]), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(&d->cnt[acc], &fs.cnt[acc], sizeof(fs.cnt[acc]), hipMemcpyHostToDevice));
  return SSD_OK;
}

int ssd_test_single_pass(ssd_handle *h, int mode, int sabotage)
{
  if(!h || mode < -1 || mode > 1 || sabotage < 0 || sabotage > 2)
    return fail(SSD_E_ARG, "ssd_test_single_pass: bad argument");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  if(mode == 1)
  {
    if(!single_pass_geometry(h->P.W, h->P.H))
      return fail(SSD_E_ARG, "ssd_test_single_pass: a tile of 1024 points is not a whole number of this geometry's camera rows");
    const size_t planeBytes = static_cast<size_t>(plane_pool_size(h->F, h->P.nPoints)) * h->imgWords * 8;
    h->planePool = static_cast<int>(plane_pool_size(h->F, h->P.nPoints));
    for(int k = 0; k < h->depth; k++)
      if(!h->lane[k].dPlaneImg)
      {
        HIP_TRY(hipMalloc(&h->lane[k].dPlaneImg, planeBytes));
        HIP_TRY(hipMemset(h->lane[k].dPlaneImg, 0, planeBytes));
        HIP_TRY(hipMalloc(&h->lane[k].dFallback, sizeof(int) * (kFallbackList + static_cast<size_t>(h->F))));
        HIP_TRY(hipMemset(h->lane[k].dFallback, 0, sizeof(int) * (kFallbackList + static_cast<size_t>(h->F))));
      }
  }
  h->singlePassMode = mode;
  h->singlePassSabotage = sabotage;
  h->singlePassBackoff = 0;
  if(mode == 1 && !h->hFallback)
  {
    HIP_TRY(hipHostMalloc(&h->hFallback, sizeof(int) * 2 * kMaxLanes, hipHostMallocDefault));
    std::memset(h->hFallback, 0, sizeof(int) * 2 * kMaxLanes);
  }
  return SSD_OK;
}

int ssd_test_plane_pool(ssd_handle *h, int planes)
{
  if(!h)
    return fail(SSD_E_ARG, "ssd_test_plane_pool: null handle");
  const int size = static_cast<int>(plane_pool_size(h->F, h->P.nPoints));
  if(planes > size)
    return fail(SSD_E_ARG, "ssd_test_plane_pool: more planes than the pool holds");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  h->planePool = planes < 0 ? size : planes;
  return size;
}

int ssd_test_single_pass_stats(ssd_handle *h, int frames, int scan_planes, long long counts[4])
{
  if(!h || !counts || frames < 0 || frames > h->F)
    return fail(SSD_E_ARG, "ssd_test_single_pass_stats: bad argument");
  counts[0] = counts[1] = counts[2] = counts[3] = 0;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  const ssd_lane &L = h->lane[h->lastLane];
  if(!scan_planes)
    counts[3] = -1;
  else if(L.dPlaneImg)
  {
    const size_t words = static_cast<size_t>(plane_pool_size(h->F, h->P.nPoints)) * h->imgWords;
    std::vector<unsigned long long> img(words);
    HIP_TRY(hipMemcpy(img.data(), L.dPlaneImg, words * 8, hipMemcpyDeviceToHost));
    for(const unsigned long long w : img)
      counts[3] += w != 0ull;
  }
  if(!h->lastSinglePass)
    return 0;
  std::vector<FrameState> st(static_cast<size_t>(frames));
  HIP_TRY(hipMemcpy(st.data(), L.dState, sizeof(FrameState) * st.size(), hipMemcpyDeviceToHost));
  for(const FrameState &fs : st)
  {
    counts[0] += fs.nStepImages > 0 && fs.specOk;
    counts[1] += fs.nStepImages > 0;
    counts[2] += fs.nPlanes;
  }
  return 1;
}

int ssd_test_single_pass_frame(ssd_handle *h, int frame, uint8_t *plane_of_bin, int info[3])
{
  if(!h || !plane_of_bin || !info || frame < 0 || frame >= h->F)
    return fail(SSD_E_ARG, "ssd_test_single_pass_frame: bad argument");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  FrameState fs;
  HIP_TRY(hipMemcpy(&fs, h->lane[h->lastLane].dState + frame, sizeof(fs), hipMemcpyDeviceToHost));
  std::memcpy(plane_of_bin, fs.specPlane, kMaxBins);
  info[0] = fs.nPlanes; info[1] = fs.specOk; info[2] = fs.nStepImages;
  return SSD_OK;
}

int ssd_test_single_pass_sample(ssd_handle *h, int frame, uint32_t *sample)
{
  if(!h || !sample || frame < 0 || frame >= h->F)
    return fail(SSD_E_ARG, "ssd_test_single_pass_sample: bad argument");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(sample, h->lane[h->lastLane].dState[frame].predSample, sizeof(unsigned int) * kMaxBins, hipMemcpyDeviceToHost));
  return SSD_OK;
}

int ssd_test_predict_table_host(const uint32_t *sample, int n_bins, int min_height, int sabotage, uint8_t *plane_of_bin)
{
  if(!sample || !plane_of_bin || n_bins < 2 || n_bins > kMaxBins || sabotage < 0 || sabotage > 2)
    return fail(SSD_E_ARG, "ssd_test_predict_table_host: bad argument");
  return predict_table(sample, n_bins, min_height, sabotage, plane_of_bin);
}

int ssd_test_prexy_host(const double range[6], const double a[9], const double b[3], float out[14])
{
  if(!range || !a || !b || !out)
    return fail(SSD_E_ARG, "ssd_test_prexy_host: null");
  PointParams P{};
  for(int i = 0; i < 9; i++) P.a[i] = a[i];
  for(int i = 0; i < 3; i++) P.b[i] = b[i];
  P.xMin = range[0]; P.xMax = range[1]; P.yMin = range[2]; P.yMax = range[3]; P.zMin = range[4]; P.zMax = range[5];
  const PreXY Q = make_pre_xy(P);
  for(int i = 0; i < 4; i++) { out[2 * i] = Q.c[i][0]; out[2 * i + 1] = Q.c[i][1]; }
  out[8] = Q.lo; out[9] = Q.hi; out[10] = Q.maxInput; out[11] = Q.boxLo; out[12] = Q.boxHi; out[13] = static_cast<float>(Q.checkInput);
  return SSD_OK;
}

int ssd_test_prez_host(const double range[6], const double a[9], const double b[3], double height_interval, int width, int height, float out[16])
{
  if(!range || !a || !b || !out || !(height_interval > 0.0) || width <= 0 || height <= 0)
    return fail(SSD_E_ARG, "ssd_test_prez_host: bad argument");
  PointParams P{};
  for(int i = 0; i < 9; i++) P.a[i] = a[i];
  for(int i = 0; i < 3; i++) P.b[i] = b[i];
  P.xMin = range[0]; P.xMax = range[1]; P.yMin = range[2]; P.yMax = range[3]; P.zMin = range[4]; P.zMax = range[5];
  P.recip = 1.0 / height_interval;                       /* as make_params (ssd_capi.hip) */
  const PreXY Q = make_pre_xy(P);
  PixelParams X{};
  X.W = width; X.H = height;
  X.xToImage = width / (P.xMax - P.xMin);
  X.yToImage = height / (P.yMax - P.yMin);
  make_pre_pixel(P, Q, X);
  for(int i = 0; i < 4; i++) out[i] = Q.zc[i];
  out[4] = Q.zNegK; out[5] = Q.zH0; out[6] = Q.zTop; out[7] = static_cast<float>(Q.zCheckTop);
  out[8] = X.fW; out[9] = X.fHalfW; out[10] = X.fNegH; out[11] = X.fHalfH; out[12] = X.pxNegK; out[13] = X.pxH0;
  out[14] = static_cast<float>(P.recip); out[15] = 0.0f;
  return SSD_OK;
}

/* k_inquad's single-precision edge tests (csrc/ssd_quadtest.h: build_quad_edges, quad_edges_classify) compiled for the host, on the d
 * and the bound K1's pre-filter would hand them (csrc/ssd_prexy.h) for n camera points */
int ssd_test_quad_edges_host(const double quad[8], const double range[6], const double a[9], const double b[3], const float *pts_xyz, int n,
                             float consts[15], int8_t *cls, double *world_xy, uint8_t *in_range_xy, int *err)
{
  if(!quad || !range || !a || !b || !pts_xyz || !consts || !cls || !world_xy || !in_range_xy || !err || n < 0)
    return fail(SSD_E_ARG, "ssd_test_quad_edges_host: bad argument");
  PointParams P{};
  for(int i = 0; i < 9; i++) P.a[i] = a[i];
  for(int i = 0; i < 3; i++) P.b[i] = b[i];
  P.xMin = range[0]; P.xMax = range[1]; P.yMin = range[2]; P.yMax = range[3]; P.zMin = range[4]; P.zMax = range[5];
  P.recip = 100.0;
  P.boxX = 256.0 / (P.xMax - P.xMin); P.boxY = 256.0 / (P.yMax - P.yMin);
  const PreXY Q = make_pre_xy(P);
  ssd::QuadTest t;
  ssd::build_quad_test(quad, t);
  *err = t.err;
  ssd::QuadGridSegs sg;
  ssd::build_grid_segs(t, P.xMin, P.yMin, P.boxX, P.boxY, sg);
  ssd::QuadEdgesF E;
  ssd::build_quad_edges(t, sg.ok, P.xMin, P.xMax, P.yMin, P.yMax, E);
  for(int s = 0; s < 4; s++) { consts[s] = E.gx[s]; consts[4 + s] = E.gy[s]; consts[8 + s] = E.g2[s]; }
  consts[12] = E.m; consts[13] = Q.dK; consts[14] = Q.dE0;
  for(int i = 0; i < n; i++)
  {
    const float x = pts_xyz[3 * i], y = pts_xyz[3 * i + 1], z = pts_xyz[3 * i + 2];
    /* the reference's rows (transformation.h:59-64) */
    const double wx = ((P.a[0] * x + P.a[1] * y) + P.a[2] * z) + P.b[0];
    const double wy = ((P.a[3] * x + P.a[4] * y) + P.a[5] * z) + P.b[1];
    world_xy[2 * i] = wx; world_xy[2 * i + 1] = wy;
    in_range_xy[i] = (wx > P.xMin && wx < P.xMax && wy > P.yMin && wy < P.yMax) ? 1 : 0;
    /* K1's d (pre_xy: z, then y, then x) and the magnitude its bound follows */
    float dx = std::fmaf(Q.c[2][0], z, Q.c[3][0]), dy = std::fmaf(Q.c[2][1], z, Q.c[3][1]);
    dx = std::fmaf(Q.c[1][0], y, dx); dy = std::fmaf(Q.c[1][1], y, dy);
    dx = std::fmaf(Q.c[0][0], x, dx); dy = std::fmaf(Q.c[0][1], x, dy);
    const float M = std::fmax(std::fabs(x), std::fmax(std::fabs(y), std::fabs(z)));
    cls[i] = static_cast<int8_t>(ssd::quad_edges_classify(E, dx, dy, std::fmaf(M, Q.dK, Q.dE0)));
  }
  return SSD_OK;
}

/* the kernels' line helpers (csrc/ssd_math.h: line_through_i / line_through_d = LineCoordinates(p, q), types.h:140-158; intersect60 =
 * Line<double>::intersection, segmentation.cpp:344-362, whose numerators are LineCoordinates::det / detx / dety) compiled for the host */
int ssd_test_line_host(const double pq[4], double abc_d[3], int32_t abc_i[3])
{
  if(!pq || !abc_d || !abc_i)
    return fail(SSD_E_ARG, "ssd_test_line_host: null");
  const LineD d = line_through_d(pq[0], pq[1], pq[2], pq[3]);
  abc_d[0] = d.a; abc_d[1] = d.b; abc_d[2] = d.c;
  const LineI i = line_through_i(static_cast<int>(pq[0]), static_cast<int>(pq[1]), static_cast<int>(pq[2]), static_cast<int>(pq[3]));
  abc_i[0] = i.a; abc_i[1] = i.b; abc_i[2] = i.c;
  return SSD_OK;
}
/* returns 1 and the intersection point when the lines meet at more than 60 degrees, else 0 */
int ssd_test_intersect_host(const double l[3], const double o[3], double xy[2])
{
  if(!l || !o || !xy)
    return fail(SSD_E_ARG, "ssd_test_intersect_host: null");
  double x = 0.0, y = 0.0;
  const bool ok = intersect60(LineD{ l[0], l[1], l[2] }, LineD{ o[0], o[1], o[2] }, x, y);
  xy[0] = x; xy[1] = y;
  return ok ? 1 : 0;
}

int ssd_test_ground_image(ssd_handle *h, int frame, uint8_t *out)
{
  if(!h || !out || frame < 0 || frame >= h->F)
    return fail(SSD_E_ARG, "ssd_test_ground_image: bad argument");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  std::vector<unsigned long long> words(h->imgWords);
  HIP_TRY(hipMemcpy(words.data(), h->lane[h->lastLane].dGroundImg + static_cast<size_t>(frame) * h->imgWords, h->imgWords * 8, hipMemcpyDeviceToHost));
  const int W = h->P.W, H = h->P.H, W64 = h->P.W64;
  for(int y = 0; y < H; y++)
    for(int x = 0; x < W; x++)
      out[static_cast<size_t>(y) * W + x] = ((words[static_cast<size_t>(y) * W64 + (x >> 6)] >> (x & 63)) & 1ull) ? 0xff : 0;
  return SSD_OK;
}

int ssd_test_sort_host(const double *dist, int n, int32_t *perm)
{
  if(!dist || !perm || n < 0 || n > 32768)
    return fail(SSD_E_ARG, "ssd_test_sort_host: bad argument");
  std::vector<double> d(dist, dist + n);
  std::vector<int> idx(static_cast<size_t>(n));
  for(int i = 0; i < n; i++)
    idx[static_cast<size_t>(i)] = i;
  gnu_sort(SortKeys{ d.data(), idx.data() }, n);
  for(int i = 0; i < n; i++)
    perm[i] = idx[static_cast<size_t>(i)];
  return SSD_OK;
}

int ssd_test_sort_device(int device, const double *dist, int n, int32_t *perm)
{
  if(!dist || !perm || n < 1 || n > 32768)
    return fail(SSD_E_ARG, "ssd_test_sort_device: bad argument");
  if(device_count() <= 0)
    return fail(SSD_E_NODEVICE, "ssd_test_sort_device: no HIP device");
  HIP_TRY(hipSetDevice(device));
  double *dd = nullptr;
  int *di = nullptr;
  HIP_TRY(hipMalloc(&dd, static_cast<size_t>(n) * sizeof(double)));
  HIP_TRY(hipMalloc(&di, static_cast<size_t>(n) * sizeof(int)));
  HIP_TRY(hipMemcpy(dd, dist, static_cast<size_t>(n) * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_sorttest, dim3(1), dim3(64), 0, nullptr, dd, di, n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(perm, di, static_cast<size_t>(n) * sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(dd); (void)hipFree(di);
  return SSD_OK;
}

int ssd_test_quad_edges_device(int device, const double *quads, int n, const double range_xy[4], float *out)
{
  if(!quads || !range_xy || !out || n < 1)
    return fail(SSD_E_ARG, "ssd_test_quad_edges_device: bad argument");
  if(device_count() <= 0)
    return fail(SSD_E_NODEVICE, "ssd_test_quad_edges_device: no HIP device");
  HIP_TRY(hipSetDevice(device));
  double *dq = nullptr;
  float *dout = nullptr;
  HIP_TRY(hipMalloc(&dq, static_cast<size_t>(n) * 8 * sizeof(double)));
  HIP_TRY(hipMalloc(&dout, static_cast<size_t>(n) * 13 * sizeof(float)));
  HIP_TRY(hipMemcpy(dq, quads, static_cast<size_t>(n) * 8 * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_quadedges, dim3(n), dim3(64), 0, nullptr, dq, n, range_xy[0], range_xy[1], range_xy[2], range_xy[3],
                     256.0 / (range_xy[1] - range_xy[0]), 256.0 / (range_xy[3] - range_xy[2]), dout);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout, static_cast<size_t>(n) * 13 * sizeof(float), hipMemcpyDeviceToHost));
  (void)hipFree(dq); (void)hipFree(dout);
  return SSD_OK;
}

int ssd_test_quad_device(int device, const double quad[8], const double *pts_xy, int n, uint8_t *inside, int *err)
{
  if(!quad || !pts_xy || !inside || !err || n < 1)
    return fail(SSD_E_ARG, "ssd_test_quad_device: bad argument");
  if(device_count() <= 0)
    return fail(SSD_E_NODEVICE, "ssd_test_quad_device: no HIP device");
  HIP_TRY(hipSetDevice(device));
  double *dq = nullptr, *dp = nullptr;
  unsigned char *di = nullptr;
  int *de = nullptr;
  HIP_TRY(hipMalloc(&dq, 8 * sizeof(double)));
  HIP_TRY(hipMalloc(&dp, static_cast<size_t>(n) * 2 * sizeof(double)));
  HIP_TRY(hipMalloc(&di, static_cast<size_t>(n)));
  HIP_TRY(hipMalloc(&de, sizeof(int)));
  HIP_TRY(hipMemcpy(dq, quad, 8 * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dp, pts_xy, static_cast<size_t>(n) * 2 * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(di, 0, static_cast<size_t>(n)));
  hipLaunchKernelGGL(k_quadtest, dim3(n > 4096 ? 16 : 1), dim3(256), 0, nullptr, dq, dp, n, di, de);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(inside, di, static_cast<size_t>(n), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(err, de, sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(dq); (void)hipFree(dp); (void)hipFree(di); (void)hipFree(de);
  return SSD_OK;
}

/* the kernels' closing compiled for the host (ssd_closing.h is host + device code): a W x H byte image (non-zero = lit) is
 * packed into the kernels' bit image; closed_out (W x H bytes, may be null) = the closed image word by word (closed_word);
 * first / last [n_cols] = the first and last closed row of the pixel columns x0, x0 + x_step, .. (closed_scan_column over
 * rows [y_from, H), in bands of band_rows rows as the kernels cut them), -1 where none */
int ssd_test_closing_host(const uint8_t *img, int width, int height, int x0, int x_step, int y_from, int band_rows, uint8_t *closed_out,
                          int32_t *first, int32_t *last, int n_cols)
{
  if(!img || width < 1 || height < 1 || x_step < 1 || x0 < 0 || y_from < 0 || band_rows < 1 || (n_cols > 0 && (!first || !last)))
    return fail(SSD_E_ARG, "ssd_test_closing_host: bad argument");
  const int W64 = (width + 63) / 64;
  std::vector<unsigned long long> bits(static_cast<size_t>(W64) * height, 0ull);
  for(int y = 0; y < height; y++)
    for(int x = 0; x < width; x++)
      if(img[static_cast<size_t>(y) * width + x])
        bits[static_cast<size_t>(y) * W64 + (x >> 6)] |= 1ull << (x & 63);
  const ssd::BitImg im{ bits.data(), width, height, W64 };
  if(closed_out)
    for(int y = 0; y < height; y++)
      for(int c = 0; c < W64; c++)
      {
        const unsigned long long w = ssd::closed_word(im, y, c);
        for(int b = 0; b < 64 && 64 * c + b < width; b++)
          closed_out[static_cast<size_t>(y) * width + 64 * c + b] = ((w >> b) & 1ull) ? 255 : 0;
      }
  for(int j = 0; j < n_cols; j++)
  {
    const int x = x0 + x_step * j;
    first[j] = last[j] = -1;
    if(x >= width)
      continue;
    for(int yA = y_from; yA < height; yA += band_rows)
      ssd::closed_scan_column(im, x, yA, yA + band_rows < height ? yA + band_rows : height, [&](int y)
      {
        if(first[j] < 0)
          first[j] = y;
        last[j] = y;
      });
  }
  return SSD_OK;
}

/* BestLine (segmentation.cpp:409-487) over a point list with the kernels' residual code compiled for the host
 * (ssd_bestline.h): every pair in the reference's order, the first of the smallest residuals wins.
 * form 0: line_residual_generic, 1: line_residual_keys (n <= 128), 2: line_residual_onepass (n <= 64) */
int ssd_test_best_line_host(const int32_t *pts_xy, int n, int form, int32_t line[3])
{
  if(!pts_xy || !line || n < 2 || form < 0 || form > 2 || (form == 1 && n > 128) || (form == 2 && n > 64))
    return fail(SSD_E_ARG, "ssd_test_best_line_host: bad argument");
  std::vector<int> px(n), py(n);
  for(int i = 0; i < n; i++)
  {
    px[i] = pts_xy[2 * i];
    py[i] = pts_xy[2 * i + 1];
  }
  double bestRes = 0.0;
  bool have = false;
  ssd::LineI best{ 0, 0, 0 };
  for(int p = 0; p < n; p++)
    for(int q = p + 1; q < n; q++)
    {
      ssd::LineI l;
      double r;
      if(form == 0)
        r = ssd::line_residual_generic(px.data(), py.data(), n, p, q, l);
      else if(form == 1)
        r = ssd::line_residual_keys(px.data(), py.data(), n, p, q, l);
      else
      {
        l = ssd::line_through_i(px[p], py[p], px[q], py[q]);
        r = ssd::line_residual_onepass([&](int i, int &x, int &y) { x = px[i]; y = py[i]; }, n, l);
      }
      if(!have || r < bestRes)
      {
        have = true;
        bestRes = r;
        best = l;
      }
    }
  line[0] = best.a; line[1] = best.b; line[2] = best.c;
  return SSD_OK;
}

/* the same builder and evaluator compiled for the host (ssd_quadtest.h is host + device code): no GPU needed */
int ssd_test_quad_host(const double quad[8], const double *pts_xy, int n, uint8_t *inside, int *err)
{
  if(!quad || !pts_xy || !inside || !err || n < 1)
    return fail(SSD_E_ARG, "ssd_test_quad_host: bad argument");
  ssd::QuadTest t;
  ssd::build_quad_test(quad, t);
  *err = t.err;
  for(int i = 0; i < n; i++)
  {
    const double x = pts_xy[2 * i], y = pts_xy[2 * i + 1];
    const bool fast = x >= t.fx0 && x < t.fx1 && y >= t.fy0 && y < t.fy1;
    inside[i] = (t.err == 0 && (fast || ssd::quad_test(t, x, y))) ? 1 : 0;
  }
  return SSD_OK;
}

int ssd_test_grid_boxes_device(int device, const double quad[8], double x_min, double y_min, double box_x, double box_y,
                               const int32_t *boxes, int n, uint8_t *inside, int *usable)
{
  if(!quad || !boxes || !inside || !usable || n < 1)
    return fail(SSD_E_ARG, "ssd_test_grid_boxes_device: bad argument");
  if(device_count() <= 0)
    return fail(SSD_E_NODEVICE, "ssd_test_grid_boxes_device: no HIP device");
  HIP_TRY(hipSetDevice(device));
  double *dq = nullptr;
  int *db = nullptr, *du = nullptr;
  unsigned char *di = nullptr;
  HIP_TRY(hipMalloc(&dq, 8 * sizeof(double)));
  HIP_TRY(hipMalloc(&db, static_cast<size_t>(n) * 4 * sizeof(int)));
  HIP_TRY(hipMalloc(&di, static_cast<size_t>(n)));
  HIP_TRY(hipMalloc(&du, sizeof(int)));
  HIP_TRY(hipMemcpy(dq, quad, 8 * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db, boxes, static_cast<size_t>(n) * 4 * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(di, 0, static_cast<size_t>(n)));
  hipLaunchKernelGGL(k_gridboxes, dim3(1), dim3(256), 0, nullptr, dq, x_min, y_min, box_x, box_y, db, n, di, du);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(inside, di, static_cast<size_t>(n), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(usable, du, sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(dq); (void)hipFree(db); (void)hipFree(di); (void)hipFree(du);
  return SSD_OK;
}

int ssd_test_hypot_device(int device, const double *a, const double *b, double *out, int n)
{
  if(device_count() <= 0)
    return fail(SSD_E_NODEVICE, "ssd_test_hypot_device: no HIP device");
  HIP_TRY(hipSetDevice(device));
  double *da = nullptr, *db = nullptr, *dout = nullptr;
  HIP_TRY(hipMalloc(&da, 8 * static_cast<size_t>(n)));
  HIP_TRY(hipMalloc(&db, 8 * static_cast<size_t>(n)));
  HIP_TRY(hipMalloc(&dout, 8 * static_cast<size_t>(n)));
  HIP_TRY(hipMemcpy(da, a, 8 * static_cast<size_t>(n), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db, b, 8 * static_cast<size_t>(n), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_hypot, dim3((n + 255) / 256), dim3(256), 0, nullptr, da, db, dout, n);
  HIP_TRY(hipMemcpy(out, dout, 8 * static_cast<size_t>(n), hipMemcpyDeviceToHost));
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
  return SSD_OK;
}

int ssd_test_stream_read(int device, const void *d_ptr, size_t bytes, int reps, void *stream, float *ms_avg)
{
  if(!d_ptr || !ms_avg || reps < 1 || bytes < 12288 || (reinterpret_cast<uintptr_t>(d_ptr) & 15u) != 0)
    return fail(SSD_E_ARG, "ssd_test_stream_read: bad argument (16-byte aligned buffer of >= 12288 bytes)");
  if(device_count() <= 0)
    return fail(SSD_E_NODEVICE, "ssd_test_stream_read: no HIP device");
  HIP_TRY(hipSetDevice(device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t nVec = bytes / 12288 * 768;                    /* whole block iterations only */
  const int chunkVec = 768 * 32;                              /* 32 iterations per block, as K1's chunks */
  const unsigned int blocks = static_cast<unsigned int>((nVec + chunkVec - 1) / chunkVec);
  float *out = nullptr;
  HIP_TRY(hipMalloc(&out, 16));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_stream_read, dim3(blocks), dim3(256), 0, s, static_cast<const float4 *>(d_ptr), nVec, chunkVec, out);
  HIP_TRY(hipEventRecord(e0, s));
  for(int r = 0; r < reps; r++)
    hipLaunchKernelGGL(k_stream_read, dim3(blocks), dim3(256), 0, s, static_cast<const float4 *>(d_ptr), nVec, chunkVec, out);
  HIP_TRY(hipEventRecord(e1, s));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.0f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  *ms_avg = ms / static_cast<float>(reps);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(out);
  return SSD_OK;
}

} // extern "C"
