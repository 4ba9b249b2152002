/*
 * ssd_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the per-frame point-cloud path.
 *
 * Pipeline over a batch of F frames resident in HBM (grid.x = frame everywhere: see the launch note on XCD balance):
 *   K1 k_hist      transform + crop + 1 cm height bin + histogram      (pointcloud.cpp:122-204)
 *   K1b k_peaks    peaks, filter, plateau pairs, bin->plateau LUT      (pointcloud.cpp:214-343, 399-418)
 *   K2 k_raster    top-down bit images of the step plateaus            (pointcloud.cpp:458-471)
 *   K3 k_outline   3x3 close + scans + best lines + corners            (segmentation.cpp:919-971)
 *   K3b k_quads    ground quadrilateral, point-in-quad tests           (pointcloud.cpp:431-443,489-512; quadrilateralTest.cpp:275-443)
 *   K4 k_inquad    in-quad filter, z sums, ground image                (pointcloud.cpp:560-581, 530-531)
 *                  (range, bin and quadrilateral in single precision first, round 6: ssd_prexy.h, ssd_quadtest.h build_quad_edges)
 *   K5 k_final     ground front edge, mean z, ToExternalWorld, result  (segmentation.cpp:879-917; pointcloud.cpp:532-547, 370-383)
 *
 * All floating-point work is fp64 with contraction off (the file is compiled with
 * -ffp-contract=off): the reference is built without FMA (CMakeLists.txt:23-28) and one
 * flipped bin or pixel moves a corner by more than the parity bar.
 * K1 is HBM-read bound (12 B per raw point), K2 / K4 re-read a third of the points and are VALU-issue bound; no MFMA anywhere.
 *
 * Batches of 64 XGA frames' worth of points and more (vertex input, the whole pipeline in one call) run the SINGLE PASS instead:
 *   K0 k_predict      histogram of a sample of the frame -> the height bins that may hold a step plateau, a plane (bit image) each
 *   K1 k_hist_planes  k_hist, and the raster of those bins' points into their planes
 *   K1b k_peaks       ... and which step plateaus the planes cover; the frames with others on k_raster's work list
 *   K2 k_raster       only the listed frames' uncovered plateaus
 *   K3 k_outline      reads a covered plateau's plane(s) in place of its step image
 * Nothing the predictor says can change a result ("The single pass").
 */
#include "ssd_device.h"
#include "ssd_math.h"
#include "ssd_quadtest.h"
#include "ssd_closing.h"
#include "ssd_bestline.h"
#include "ssd_sort.h"

namespace ssd
{

#include "ssd_phase.h"      /* tools builds: clocks at the marks below; product build: empty macros */

/* Bounds-checked tools build (-DSSD_CHECKED, tools/checked.sh; GPU sanitizers are not available on this pool): the index of every
 * store or atomic whose address comes from a point, a pixel, a window or a list is compared with the extent of what it writes
 * into FIRST; a violation is reported from the device (one line, "SSD_CHECK site ...") and the access is dropped.  In the
 * product build the macro is the constant `true`. */
#ifdef SSD_CHECKED
__device__ __noinline__ void ssd_chk_report(int site, unsigned long long idx, unsigned long long limit)
{
  printf("SSD_CHECK site %d index %llu limit %llu block (%u, %u) thread %u\n", site, idx, limit, blockIdx.x, blockIdx.y, threadIdx.x);
}
__device__ __forceinline__ bool ssd_chk(int site, unsigned long long idx, unsigned long long limit)
{
  if(idx < limit)
    return true;
  ssd_chk_report(site, idx, limit);
  return false;
}
#define SSD_CHK(site, idx, limit) ssd_chk((site), static_cast<unsigned long long>(idx), static_cast<unsigned long long>(limit))
#else
#define SSD_CHK(site, idx, limit) (true)
#endif

/* Tools build (-DSSD_COUNT, tools/k1count.py): what K1's windows do - tiles with candidate points, window moves, words flushed,
 * pixels that missed the window.  In the product build the macros are empty. */
#ifdef SSD_COUNT
__device__ unsigned long long ssd_k1_counters[8];
#define SSD_CNT(i, n) do { if(threadIdx.x % 64 == 0) atomicAdd(&ssd_k1_counters[i], static_cast<unsigned long long>(n)); } while(0)
#define SSD_CNT_LANES(i, pred) do { const unsigned long long cnt_m = __ballot(pred); if(threadIdx.x % 64 == 0 && cnt_m) atomicAdd(&ssd_k1_counters[i], static_cast<unsigned long long>(__popcll(cnt_m))); } while(0)
#else
#define SSD_CNT(i, n) do { } while(0)
#define SSD_CNT_LANES(i, pred) do { } while(0)
#endif

/* ========================================================================= */
/* shared per-point arithmetic                                                */

struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };

/* CameraToWorld (transformation.h:59-64, 79-87): a*x + b, float promoted to double, row sums left to right, then the
 * translation; followed by the non-zero test (pointcloud.cpp:143-146) and the six strict range compares of
 * getPointsInRange (pointcloud.cpp:150-165).  Flat code: all three rows, then the seven tests combined without short
 * circuit.  Early exits only pay when all 64 lanes of a wave take them, which a camera image next to never offers; as nested
 * branches they cost the default values of everything the point contributes, re-materialised at every level (9 moves per
 * point in K1's ISA).  (The throughput of K1 / K2 on float3 input did not move with it — they sit on the memory roof
 * of the GPU's current clock state — but the instruction count did: what 16-bit depth input, VALU-bound, runs on.) */
__device__ __forceinline__ bool world_point_flat(const PointParams &P, const F3 &v, double &wx, double &wy, double &wz)
{
  const double x = v.x, y = v.y, z = v.z;
  wx = (P.a[0] * x + P.a[1] * y) + P.a[2] * z;
  wy = (P.a[3] * x + P.a[4] * y) + P.a[5] * z;
  wz = (P.a[6] * x + P.a[7] * y) + P.a[8] * z;
  wx = wx + P.b[0];
  wy = wy + P.b[1];
  wz = wz + P.b[2];
  return (v.z > 0.0f) & (wx > P.xMin) & (wx < P.xMax) & (wy > P.yMin) & (wy < P.yMax) & (wz > P.zMin) & (wz < P.zMax);
}

/* The same decisions taken height first, for the passes that drop most points on their height bin: the z row
 * and the z tests, then (only for points whose bin matters) the x and y rows and their tests.  The
 * conjunction of tests and every operation are those of world_point_flat. */
__device__ __forceinline__ bool world_z(const PointParams &P, const F3 &v, double &wz)
{
  if(!(v.z > 0.0f))
    return false;
  wz = (P.a[6] * static_cast<double>(v.x) + P.a[7] * static_cast<double>(v.y)) + P.a[8] * static_cast<double>(v.z);
  wz = wz + P.b[2];
  return wz > P.zMin && wz < P.zMax;
}
/* world_z without the early exit, for flat code (wz is meaningless when false is returned) */
__device__ __forceinline__ bool world_z_flat(const PointParams &P, const F3 &v, double &wz)
{
  wz = (P.a[6] * static_cast<double>(v.x) + P.a[7] * static_cast<double>(v.y)) + P.a[8] * static_cast<double>(v.z);
  wz = wz + P.b[2];
  return v.z > 0.0f && wz > P.zMin && wz < P.zMax;
}
__device__ __forceinline__ bool world_xy(const PointParams &P, const F3 &v, double &wx, double &wy)
{
  const double x = v.x, y = v.y, z = v.z;
  wx = (P.a[0] * x + P.a[1] * y) + P.a[2] * z;
  wy = (P.a[3] * x + P.a[4] * y) + P.a[5] * z;
  wx = wx + P.b[0];
  wy = wy + P.b[1];
  return wx > P.xMin && wx < P.xMax && wy > P.yMin && wy < P.yMax;
}

/* calcHeights (pointcloud.cpp:175): truncating conversion, value is in [0, nBins) */
__device__ __forceinline__ int height_bin(const PointParams &P, double wz)
{
  return static_cast<int>((wz - P.zMin) * P.recip);
}

/* ========================================================================= */
/* K1: histogram                                                              */

/* k_raster is built for 6 waves per SIMD: 75 VGPRs, no scratch spills (round 2: 7 waves, 72 VGPRs, 3 spills).  Measured in
 * round 3, same box, alternating runs, XGA batch: 8 / 7 / 6 / 5 / 4 waves 0.848 / 0.823 / 0.800 / 0.801 / 0.799 ms; FHD stress:
 * 7 / 6 / 5 within 1 % — the walk is bound by instruction issue, not by latency: residency beyond four waves buys nothing */
/* k_inquad (round 6: every decision in single precision first, the doubles out of line): 5 waves per SIMD - 92 vector registers,
 * no spill of either kind in the XGA instantiation; 6 waves spill 3 vector registers and run a third slower (0.64 -> 0.84 ms), 4 waves
 * the same as 5 (profiles/r06_k4_edges.txt).  Until round 6: 8 waves at 48 scalar spills in the loop. */
#ifndef SSD_K4_WAVES
#define SSD_K4_WAVES 5
#endif
#ifndef SSD_K2_WAVES
#define SSD_K2_WAVES 6
#endif
constexpr int kThreads = 256;
constexpr int kPts = 4;                 /* points per thread per iteration: four CONSECUTIVE points (48 B) */
constexpr int kTile = kThreads * kPts;  /* 1024 points per block iteration */
#ifndef SSD_HIST_COPIES
#define SSD_HIST_COPIES 32
#endif
constexpr int kHistCopies = SSD_HIST_COPIES;         /* LDS histogram privatised by lane & 31: bank = copy, no conflicts */

/* Point sources of the streaming kernels */
constexpr int kSrcF3 = 0;          /* float xyz, 12-byte loads (unaligned frames, or a point count not divisible by 4) */
constexpr int kSrcF3Aligned = 1;   /* float xyz, 16-byte loads */
constexpr int kSrcDepth16 = 2;     /* 16-bit depth image + intrinsics: deprojected on the fly */

/* Four consecutive points of one lane.  kSrcF3Aligned: three 16-byte loads (lanes 48 B apart; the three
 * instructions of a wave together cover 3 KiB contiguously — measured 6.2-6.3 TB/s on MI355X, the same as
 * a plain float4 stream, tools/loadbench.hip).  kSrcF3 (frame base/stride not 16-byte aligned, or the
 * tail of a frame whose point count is not a multiple of 4): 12-byte loads.  kSrcDepth16: four 16-bit depth
 * values (one 8-byte load) turned into points exactly as librealsense's pointcloud block does in float:
 * d = raw * depth_units; point = (d * xmap[u], d * ymap[v], d); raw = 0 gives the invalid point (0,0,0). */
/* kSrcDepth16: the lane's four depth pixels and the maps' values for them -> four points, in float as librealsense does
 * (d = raw * depth_units; point = (d * xmap[u], d * ymap[v], d); raw = 0 gives the invalid point (0,0,0)) */
__device__ __forceinline__ void deproject4(const uint2 raw, const float4 xm, const float ym, const float units, F3 (&v)[kPts])
{
  const float d0 = static_cast<float>(raw.x & 0xffffu) * units, d1 = static_cast<float>(raw.x >> 16) * units;
  const float d2 = static_cast<float>(raw.y & 0xffffu) * units, d3 = static_cast<float>(raw.y >> 16) * units;
  v[0] = F3{ d0 * xm.x, d0 * ym, d0 };
  v[1] = F3{ d1 * xm.y, d1 * ym, d1 };
  v[2] = F3{ d2 * xm.z, d2 * ym, d2 };
  v[3] = F3{ d3 * xm.w, d3 * ym, d3 };
}
/* image row of point index idx (DepthSrc::rowMagic) */
__device__ __forceinline__ int depth_row(const DepthSrc &D, int idx)
{
  return D.rowMagic ? static_cast<int>(__umulhi(static_cast<unsigned int>(idx), D.rowMagic) >> 7) : idx / D.W;
}

template<int SRC>
__device__ __forceinline__ void load_points(const float *__restrict__ base, int idx0, int end, F3 (&v)[kPts], const DepthSrc &D)
{
  if(SRC == kSrcDepth16)
  {
    /* width % 4 == 0 and idx0 % 4 == 0 (ssd_enqueue_depth checks the first, every caller provides the second): the four
     * pixels share the image row, lie wholly inside or wholly outside the frame, and their x-map entries are one aligned
     * 16-byte load (round 3: four dependent 4-byte loads and a division per call) */
    const unsigned short *depth = reinterpret_cast<const unsigned short *>(base);
    const bool inside = idx0 + kPts <= end;
    const uint2 raw = inside ? *reinterpret_cast<const uint2 *>(depth + idx0) : make_uint2(0u, 0u);
    const int row = depth_row(D, idx0), col = idx0 - row * D.W;
    const float ym = D.ymap[min(row, D.H - 1)];               /* lanes past the end of the frame (raw = 0) stay inside the map */
    const float4 xm = *reinterpret_cast<const float4 *>(D.xmap + col);
    deproject4(raw, xm, ym, D.depthUnits, v);
    return;
  }
  if(SRC == kSrcF3Aligned && idx0 + kPts <= end)
  {
    const float4 *q = reinterpret_cast<const float4 *>(base + 3 * static_cast<size_t>(idx0));
    const float4 a = q[0], b = q[1], c = q[2];
    v[0] = F3{ a.x, a.y, a.z };
    v[1] = F3{ a.w, b.x, b.y };
    v[2] = F3{ b.z, b.w, c.x };
    v[3] = F3{ c.y, c.z, c.w };
    return;
  }
#pragma unroll
  for(int j = 0; j < kPts; j++)
  {
    const int idx = idx0 + j;
    v[j] = F3{ 0.0f, 0.0f, 0.0f };
    if(idx < end)
      v[j] = *reinterpret_cast<const F3 *>(base + 3 * static_cast<size_t>(idx));
  }
}

/* The lane's four points of a tile that lies wholly inside the frame: no bounds, no branches, nothing between the loads and their
 * first use.  Round 5: load_points() has two paths (this one, and point by point at the frame's end); where they join the compiler
 * puts copies of the loaded registers INTO the fast path - an s_waitcnt vmcnt right behind the loads - and the "prefetch" of the
 * next tile waited for its data before the current tile was touched (found in the ISA of the single pass's K1, which a build fed
 * from the Infinity Cache had shown to be waiting for memory: 1.69 ms against 2.12 from HBM; tools/mkvariant.sh's SED_EXPR). */
template<int SRC>
__device__ __forceinline__ void load_points_full(const float *__restrict__ base, int idx0, F3 (&v)[kPts])
{
  if(SRC == kSrcDepth16)
    return;                                     /* vertex input only: the depth stream has a loop of its own */
  if(SRC == kSrcF3Aligned)
  {
    const float4 *q = reinterpret_cast<const float4 *>(base + 3 * static_cast<size_t>(idx0));
#ifdef SSD_NT_LOADS           /* tools: the frames read as a stream that is not to be kept in the caches */
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v *qv = reinterpret_cast<const f4v *>(q);
    const f4v av = __builtin_nontemporal_load(qv), bv = __builtin_nontemporal_load(qv + 1), cv = __builtin_nontemporal_load(qv + 2);
    const float4 a = make_float4(av.x, av.y, av.z, av.w), b = make_float4(bv.x, bv.y, bv.z, bv.w), c = make_float4(cv.x, cv.y, cv.z, cv.w);
#else
    const float4 a = q[0], b = q[1], c = q[2];
#endif
    v[0] = F3{ a.x, a.y, a.z };
    v[1] = F3{ a.w, b.x, b.y };
    v[2] = F3{ b.z, b.w, c.x };
    v[3] = F3{ c.y, c.z, c.w };
    return;
  }
#pragma unroll
  for(int j = 0; j < kPts; j++)
    v[j] = *reinterpret_cast<const F3 *>(base + 3 * static_cast<size_t>(idx0 + j));
}

/* The same without knowing that the tile is whole: the loads go to addresses clamped into [0, end) - no bounds, no branches, and so
 * nothing that uses the data behind the loads - and zero_beyond(), called where the points are first used, blanks the points at or
 * beyond `end` (z = 0: "no measurement", dropped by every consumer first).  end >= 4; vertex input only.  kSrcF3Aligned: idx0 and end
 * are multiples of 4 (points_aligned()), so a lane's four points lie wholly inside or wholly beyond. */
template<int SRC>
__device__ __forceinline__ void load_points_clamped(const float *__restrict__ base, int idx0, int end, F3 (&v)[kPts])
{
  if(SRC == kSrcDepth16)
    return;
  if(SRC == kSrcF3Aligned)
  {
    load_points_full<SRC>(base, min(idx0, end - kPts), v);
    return;
  }
#pragma unroll
  for(int j = 0; j < kPts; j++)
    v[j] = *reinterpret_cast<const F3 *>(base + 3 * static_cast<size_t>(min(idx0 + j, end - 1)));
}
__device__ __forceinline__ void zero_beyond(F3 (&v)[kPts], int idx0, int end)
{
#pragma unroll
  for(int j = 0; j < kPts; j++)
    if(idx0 + j >= end)
      v[j] = F3{ 0.0f, 0.0f, 0.0f };
}

/* "The loaded registers are first looked at HERE": an empty asm that takes and returns the twelve registers of a lane's four points.
 * Placed behind the tile the loads were issued in front of: without it the compiler builds the (x, y) register pairs of the next
 * tile's points (v_pk_fma_f32 and the fp64 conversions want them aligned) right behind the loads - and waits for the data there. */
__device__ __forceinline__ void first_use(F3 (&v)[kPts])
{
  asm volatile("" : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[0].z), "+v"(v[1].x), "+v"(v[1].y), "+v"(v[1].z),
                    "+v"(v[2].x), "+v"(v[2].y), "+v"(v[2].z), "+v"(v[3].x), "+v"(v[3].y), "+v"(v[3].z));
}

/* The streaming kernels (K1, K2, K4) share one loop shape: a block owns a contiguous chunk of a frame and
 * walks it in tiles of 1024 points; the loads of the next tile are issued before the current tile is
 * processed (register double buffer), so that HBM requests stay in flight while the SIMDs do the fp64 work. */
/* body(v) is called once per tile with the lane's four points.  Two register buffers used alternately (the loop is
 * unrolled by two tiles) instead of one copied into the other after every tile: 12 moves fewer per tile. */
#define SSD_STREAM_LOOP(body)                                                                       \
  {                                                                                                 \
    F3 va[kPts], vb[kPts];                                                                          \
    const int fullEnd = begin + (end - begin) / kTile * kTile;  /* tiles wholly inside the chunk */ \
    if(begin < fullEnd)                                                                             \
      load_points_full<SRC>(base, begin + kPts * tid, va);                                          \
    for(int i0 = begin; i0 < fullEnd; i0 += 2 * kTile)                                              \
    {                                                                                               \
      const bool more1 = i0 + kTile < fullEnd;                                                      \
      if(more1)                                                                                     \
        load_points_full<SRC>(base, i0 + kTile + kPts * tid, vb);                                   \
      body(va);                                                                                     \
      if(!more1)                                                                                    \
        break;                                                                                      \
      first_use(vb);                                                                                \
      const bool more2 = i0 + 2 * kTile < fullEnd;                                                  \
      if(more2)                                                                                     \
        load_points_full<SRC>(base, i0 + 2 * kTile + kPts * tid, va);                               \
      body(vb);                                                                                     \
      if(more2)                                                                                     \
        first_use(va);                                                                              \
    }                                                                                               \
    if(fullEnd < end)                                                                               \
    {                                                                                               \
      load_points<SRC>(base, fullEnd + kPts * tid, end, va, D);       /* the frame's last, partial tile */ \
      body(va);                                                                                     \
    }                                                                                               \
  }

/* K2 and K4 run after frame-wide decisions (plateau table, outlines) and need only the points of a few
 * height bins.  Camera rows sweep one plateau at a time, so whole runs of consecutive points are irrelevant
 * to them.  K1 therefore leaves one 32-bit mask per CELL — 64 consecutive points = the 4 points of each of
 * the 16 lanes of one DPP row — saying which groups of 4 height bins occur in it (4 B per 768 B of input:
 * no measurable traffic, unlike a byte per point which costs K1 a third of its bandwidth,
 * tools/storebench.hip).  The later passes gather the cells that hold a bin they care about into a compact
 * list per block and walk only those (cell_list_build / SSD_CELL_LOOP below): nothing else is loaded, and
 * every wave iteration works on four wanted cells instead of on 256 consecutive points of which half are
 * somebody else's (wave-tile gating: 52 % of the tiles at 52 % lane use in K2; cells: 34 % at 79 %; tools/cell_analysis.py). */
constexpr int kBinsPerGroup = 4;      /* 128 bins -> 32 groups: one 32-bit mask per cell */
constexpr int kWavesPerBlock = kThreads / 64;
constexpr int kCell = 64;                         /* points per cell */
constexpr int kCellsPerTile = kTile / kCell;      /* 16 */
/* a block's chunk is at most this many tiles (choose_chunk; ssd_launch.h holds the same numbers for the host): K1 keeps the
 * chunk's cell masks in LDS, K2 / K4 / K6 the list of its wanted cells.  K2 gets tall chunks: its waves walk down cell
 * columns and pay a window flush at every column change (measured, XGA batch: 2.5 ms with 8-row chunks, 1.5 with 16,
 * 1.0 with 32) */
#ifndef SSD_MAX_TILES
#define SSD_MAX_TILES 32
#endif
constexpr int kMaxTilesPerBlock = SSD_MAX_TILES;
constexpr int kMaxCellsPerBlock = kMaxTilesPerBlock * kCellsPerTile;
constexpr int kMaxTilesPerBlockRaster = 128;
constexpr int kMaxCellsPerBlockRaster = kMaxTilesPerBlockRaster * kCellsPerTile;
constexpr int kMaxTilesPerBlockInquad = 128;
constexpr int kMaxCellsPerBlockInquad = kMaxTilesPerBlockInquad * kCellsPerTile;

/* OR over the 16 lanes of a DPP row (row_ror 8, 4, 2, 1): every lane of the row gets the row's result */
__device__ __forceinline__ unsigned int row_or_u32(unsigned int v)
{
  v |= static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x128, 0xf, 0xf, false));
  v |= static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x124, 0xf, 0xf, false));
  v |= static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x122, 0xf, 0xf, false));
  v |= static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x121, 0xf, 0xf, false));
  return v;
}

/* K1 also leaves, per cell, the bounding box of its in-range points in world x / y on a 256 x 256 grid over the measuring
 * range (one byte per bound: x0, x1, y0, y1; a cell with no in-range point has an empty mask and a meaningless box).
 * k_inquad decides from it, without loading the cell, that all its points lie outside the ground quadrilateral, or all
 * inside a tread's (its thresholds on this grid carry a margin far above any rounding: InquadLds::liveBox).
 *
 * Round 4: the extremes are tracked per point on the HIGH dwords of d = w - min (doubles, d > 0 for a point in range): for
 * positive doubles the high dword (exponent + 20 mantissa bits) orders like an unsigned integer, so a point costs two
 * subtractions and four 32-bit min / max (round 3: two subtractions, two multiplications, two conversions, a pack and two
 * packed min / max), and the reduction over the 16 lanes of a DPP row is one v_min_u32_dpp / v_max_u32_dpp per step and
 * value.  Only the cell's four results are turned into grid cells (lanes 0, 16, 32, 48): the minimum with the low dword 0
 * (rounded down), the maximum with the high dword + 1 (rounded up) - the box can only grow by that, never lose a point:
 * "a cell's box [x0, x1] x [y0, y1] holds points with xMin + x0 / boxX <= x < xMin + (x1 + 1) / boxX" (live_box_thresholds)
 * stays true. */
__device__ __forceinline__ unsigned int row_min_u32(unsigned int v)
{
  v = min(v, static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x128, 0xf, 0xf, false)));
  v = min(v, static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x124, 0xf, 0xf, false)));
  v = min(v, static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x122, 0xf, 0xf, false)));
  v = min(v, static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x121, 0xf, 0xf, false)));
  return v;
}
__device__ __forceinline__ unsigned int row_max_u32(unsigned int v)
{
  v = max(v, static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x128, 0xf, 0xf, false)));
  v = max(v, static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x124, 0xf, 0xf, false)));
  v = max(v, static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x122, 0xf, 0xf, false)));
  v = max(v, static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x121, 0xf, 0xf, false)));
  return v;
}
/* The box of a cell from the four extremes' high dwords: a minimum with the low dword 0 is the lowest the distance can have
 * been (its cell is below 256: the distance is below the range), a maximum with the NEXT high dword the highest (clamped). */
__device__ __forceinline__ unsigned int cell_box_from_high_dwords(unsigned int x0, unsigned int x1, unsigned int y0, unsigned int y1,
                                                                  double boxX, double boxY)
{
  const unsigned int cx0 = static_cast<unsigned int>(__hiloint2double(static_cast<int>(x0), 0) * boxX);
  const unsigned int cy0 = static_cast<unsigned int>(__hiloint2double(static_cast<int>(y0), 0) * boxY);
  const unsigned int cx1 = min(static_cast<unsigned int>(__hiloint2double(static_cast<int>(x1 + 1u), 0) * boxX), 255u);
  const unsigned int cy1 = min(static_cast<unsigned int>(__hiloint2double(static_cast<int>(y1 + 1u), 0) * boxY), 255u);
  return (cx0 | (cx1 << 8)) | ((cy0 | (cy1 << 8)) << 16);
}

/* The list of the cells of a block's chunk whose mask meets `wanted`, in LDS, COLUMN-major: the cells of one
 * cell column (`cols` cells apart: vertically adjacent in the camera image) follow each other, so a wave that walks
 * the list stays on one patch of the top-down image that creeps down row by row — what its LDS write-combining
 * window needs — and jumps only at a column change.  Entries are cell indices relative to the chunk's first cell.
 * wantCell(record) decides from the cell's record (x = mask of the groups of 4 height bins that occur, y = bounding box,
 * x0 | x1 << 8 | y0 << 16 | y1 << 24 on the 256 x 256 grid) whether the cell is walked.
 * All threads of the block call it; returns the number of entries (block-uniform).  scratch: 2 * kWavesPerBlock words. */
template<typename Want>
__device__ __forceinline__ int cell_list_build(const uint2 *__restrict__ cellInfo, int nCells, int cols, Want wantCell,
                                               unsigned short *list, unsigned int *scratch)
{
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rows = (nCells + cols - 1) / cols;
  constexpr int kFastPasses = 2;                 /* chunks of up to 32 tiles: the product's geometry for batches */
  if(rows * cols <= kFastPasses * kThreads)
  {
    /* All records of the chunk requested at once, every pass's ballot taken, ONE barrier for the counts, one for the list:
     * two barriers and one memory round trip, where the loop below pays two of each per 256 cells (measured: the list was
     * 11 % of a block's life in k_raster and k_inquad). */
    int cidx[kFastPasses];
    uint2 rec[kFastPasses];
#pragma unroll
    for(int p = 0; p < kFastPasses; p++)
    {
      const int i = p * kThreads + tid;
      const int cx = i / rows, r = i - cx * rows;
      const int c = r * cols + cx;
      cidx[p] = (cx < cols && c < nCells) ? c : -1;
      rec[p] = cidx[p] >= 0 ? cellInfo[c] : make_uint2(0u, 0u);
    }
    bool want[kFastPasses];
    int pos[kFastPasses];
#pragma unroll
    for(int p = 0; p < kFastPasses; p++)
    {
      want[p] = cidx[p] >= 0 && wantCell(rec[p]);
      const unsigned long long b = __ballot(want[p]);
      pos[p] = __popcll(b & ((1ull << lane) - 1ull));
      if(lane == 0)
        scratch[p * kWavesPerBlock + wave] = static_cast<unsigned int>(__popcll(b));
    }
    __syncthreads();
    int total = 0;
#pragma unroll
    for(int p = 0; p < kFastPasses; p++)
    {
      int before = total;
#pragma unroll
      for(int w = 0; w < kWavesPerBlock; w++)
      {
        const int n = static_cast<int>(scratch[p * kWavesPerBlock + w]);
        before += w < wave ? n : 0;
        total += n;
      }
      if(want[p])
        list[before + pos[p]] = static_cast<unsigned short>(cidx[p]);
    }
    __syncthreads();
    return total;
  }
  int total = 0;
  for(int i0 = 0; i0 < rows * cols; i0 += kThreads)
  {
    const int i = i0 + tid;
    const int cx = i / rows, r = i - cx * rows;
    const int c = r * cols + cx;
    const bool want = cx < cols && c < nCells && wantCell(cellInfo[c]);
    const unsigned long long b = __ballot(want);
    if(lane == 0)
      scratch[wave] = static_cast<unsigned int>(__popcll(b));
    __syncthreads();
    int before = total;
#pragma unroll
    for(int w = 0; w < kWavesPerBlock; w++)
    {
      const int n = static_cast<int>(scratch[w]);
      before += w < wave ? n : 0;
      total += n;
    }
    if(want)
      list[before + __popcll(b & ((1ull << lane) - 1ull))] = static_cast<unsigned short>(c);
    __syncthreads();
  }
  return total;
}

/* A wave takes a contiguous run of GROUPS of four list entries; DPP row q (lanes 16q .. 16q+15) takes entry 4g + q:
 * its 16 lanes read the cell's 768 bytes contiguously, four consecutive points per lane.  Rows beyond the end of the
 * list get invalid points (z = 0), which every consumer drops first. */
template<int SRC>
__device__ __forceinline__ void load_cell(const float *__restrict__ base, int cell0, const unsigned short *list, int entry, int count,
                                          int lane, int nPoints, F3 (&v)[kPts], const DepthSrc &D)
{
  if(entry < count)
    load_points<SRC>(base, (cell0 + static_cast<int>(list[entry])) * kCell + kPts * (lane & 15), nPoints, v, D);
  else
  {
#pragma unroll
    for(int j = 0; j < kPts; j++)
      v[j] = F3{ 0.0f, 0.0f, 0.0f };
  }
}

/* ========================================================================= */
/* LDS image windows, pixel keys, fixed-point z: shared by K1 (single pass), K2 and K4 */

/* Write-combining LDS windows for the rasterising kernels — one per WAVE, no barriers.
 *
 * Setting one bit per point with global atomics costs more than the whole fp64 path (memory-side atomics:
 * ~1.9 ms of K2's 3.8 ms per 1024 XGA frames, profiles/r01).  A wave owns 256 consecutive camera pixels of
 * every camera row of its block's chunk; on one plateau those land on a patch of a few 64-bit word columns
 * that creeps down the top-down image by one or two rows per camera row.  So each wave keeps a window of
 * kWinRows x kWinCols words of ONE image (slot, row0, col0 — wave-uniform) in LDS, ORs bits into it with LDS
 * atomics (no global traffic, no vmcnt) and writes the non-zero words out with global atomics only when the
 * window moves and at the end.  Bits outside the window go straight to memory, so the result never depends
 * on where the window is; when more lanes missed than hit during a tile the wave flushes and re-anchors at
 * the lowest missing (image, row).  Everything is decided with ballots and shuffles inside the wave. */
constexpr int kWinWords = 256;      /* 2 KiB per wave; shaped (256 >> winShift) rows x (1 << winShift) word columns, see PixelParams::winShift */

struct ImageBox { int yMin, yMax, xMin, xMax; };

struct WaveWindow
{
  int slot = -1, row0 = 0, col0 = 0;     /* wave-uniform */
  unsigned int limitWords = 0xffffffffu; /* 64-bit words of the frame's images behind `images` (SSD_CHECKED builds compare) */
};

/* Bits that miss the window go straight to memory; their bounding box is kept per WAVE in LDS (wm[2..5] = row min /
 * max, word column min / max; wm[6] = the image slots touched; wm[0..1], wm[7] unused): four LDS
 * min/max per miss, no per-lane state.  One box for all slots of the wave: a wave that misses in several images
 * (outliers) widens each of them to the union — boxes only bound the region K3 / K5 visit and clear. */
constexpr int kWaveMissWords = 8;
__device__ __forceinline__ void wavemiss_init(unsigned int *wm)
{
  wm[0] = 0xffffffffu; wm[1] = 0u;
  wm[2] = 0x7fffffffu; wm[3] = 0xffffffffu;              /* as int: INT_MAX, -1 */
  wm[4] = 0x7fffffffu; wm[5] = 0xffffffffu;
  wm[6] = 0u; wm[7] = 0u;
}
/* end of the kernel, whole wave: the wave's miss box into the block's per-slot boxes */
__device__ __forceinline__ void wavemiss_flush(unsigned int *wm, ImageBox *boxes, int lane)
{
  const unsigned int slots = wm[6];
  if(lane < 32 && ((slots >> lane) & 1u))
  {
    const int *b = reinterpret_cast<const int *>(wm);
    atomicMin(&boxes[lane].yMin, b[2]); atomicMax(&boxes[lane].yMax, b[3]);
    atomicMin(&boxes[lane].xMin, b[4]); atomicMax(&boxes[lane].xMax, b[5]);
  }
}

/* signed minimum / maximum over the wave: rows by DPP, the four rows by v_readlane */
__device__ __forceinline__ int wave_min_i(int v)
{
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x128, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x124, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x122, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x121, 0xf, 0xf, false));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_i(int v)
{
  v = max(v, __builtin_amdgcn_update_dpp(static_cast<int>(0x80000000u), v, 0x128, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(static_cast<int>(0x80000000u), v, 0x124, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(static_cast<int>(0x80000000u), v, 0x122, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(static_cast<int>(0x80000000u), v, 0x121, 0xf, 0xf, false));
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

/* writes the wave's window out (non-zero words only), clears it, extends the image's bounding box; all 64 lanes */
__device__ __forceinline__ void wavewin_flush(unsigned long long *ww, const WaveWindow &w, unsigned long long *__restrict__ images,
                                              unsigned int imgWords, int W64, int winShift, ImageBox *boxes, int lane)
{
  if(w.slot < 0)
    return;
  unsigned long long *img = images + static_cast<size_t>(w.slot) * imgWords;
  int y0 = 0x7fffffff, y1 = -1, x0 = 0x7fffffff, x1 = -1;
#pragma unroll
  for(int k = 0; k < kWinWords / 64; k++)
  {
    const int i = k * 64 + lane;
    const unsigned long long v = ww[i];
    if(v)
    {
      const int y = w.row0 + (i >> winShift), x = w.col0 + (i & ((1 << winShift) - 1));
      if(SSD_CHK(1, static_cast<size_t>(w.slot) * imgWords + static_cast<size_t>(y) * W64 + x, w.limitWords) && SSD_CHK(2, x, W64))
        atomicOr(img + static_cast<size_t>(y) * W64 + x, v);
      ww[i] = 0ull;
      y0 = min(y0, y); y1 = max(y1, y);
      x0 = min(x0, x); x1 = max(x1, x);
    }
  }
  y1 = wave_max_i(y1);
  if(y1 >= 0)
  {
    y0 = wave_min_i(y0); x0 = wave_min_i(x0); x1 = wave_max_i(x1);
    if(lane == 0)
    {
      atomicMin(&boxes[w.slot].yMin, y0); atomicMax(&boxes[w.slot].yMax, y1);
      atomicMin(&boxes[w.slot].xMin, x0); atomicMax(&boxes[w.slot].xMax, x1);
    }
  }
}

/* A lit pixel as one sortable word: image slot (5 bits) | row (13) | column (13); width and height <= 8192 (ssd_create).
 * kNoPixel sorts last.  (key >> 6) names the pixel's 64-bit image word. */
constexpr unsigned int kNoPixel = 0xffffffffu;
__device__ __forceinline__ unsigned int pixel_key(int slot, int iy, int ix)
{
  return (static_cast<unsigned int>(slot) << 26) | (static_cast<unsigned int>(iy) << 13) | static_cast<unsigned int>(ix);
}

/* minimum over the wave, the same in every lane (and uniform to the compiler): the rows by DPP, the four rows by v_readlane
 * (round 4: six ds_bpermute steps per value) */
__device__ __forceinline__ unsigned int wave_min_u32(unsigned int v)
{
  v = row_min_u32(v);
  return min(min(static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(v), 0)), static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(v), 16))),
             min(static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(v), 32)), static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(v), 48))));
}

/* The window's origin as a pixel_key (slot, row0, 64 * col0): for a pixel key k of the same image at or beyond the origin,
 * d = k - base is (row - row0) << 13 | (column - 64 * col0); a pixel of another image, a higher row, or a column left of
 * the origin borrows into the upper fields and fails the bounds below (64 * col0 + window width <= 8192), so one
 * subtraction and two compares decide "inside the window".  kNoWindow (bit 31, which no pixel key has) fails for all. */
constexpr unsigned int kNoWindow = 0x80000000u;
__device__ __forceinline__ unsigned int window_base(const WaveWindow &w)
{
  return w.slot < 0 ? kNoWindow : pixel_key(w.slot, w.row0, w.col0 << 6);
}
/* rows (kWinWords >> winShift) and pixel columns (64 << winShift) of a window are powers of two, so "row offset below the
 * rows and column offset below the columns" is "d has no bit outside the two offset fields": one AND and one compare with
 * zero (round 3: two shifts / masks and two compares) */
__device__ __forceinline__ bool window_hit(unsigned int d, int winShift)
{
  const unsigned int inside = (((static_cast<unsigned int>(kWinWords) >> winShift) - 1u) << 13) | ((64u << winShift) - 1u);
  return (d & ~inside) == 0u;
}

/* Before a tile's pixels go out, all 64 lanes; `first` = the lane's lowest pixel_key of this tile (kNoPixel: none).
 * When more than half of the lanes that have pixels would miss the window, it is flushed and re-anchored at the lowest
 * such key FIRST — deciding only after the tile had gone out (round 1 / 2a) sent every wave's first tile and the tile
 * after each change of cell column straight to memory: 10.5 % of all words, 2.2 % now (tools/window_sim.py). */
__device__ __forceinline__ void wavewin_prepare(unsigned long long *ww, WaveWindow &w, unsigned long long *__restrict__ images,
                                                unsigned int imgWords, int W64, int winShift, ImageBox *boxes, unsigned int first, int lane)
{
  const bool has = first != kNoPixel;
  const bool miss = has && !window_hit(first - window_base(w), winShift);
  const unsigned long long missing = __ballot(miss);
  if(missing == 0ull || 2 * __popcll(missing) <= __popcll(__ballot(has)))
    return;
  const unsigned int lowest = __builtin_amdgcn_readfirstlane(wave_min_u32(miss ? first : kNoPixel));
  wavewin_flush(ww, w, images, imgWords, W64, winShift, boxes, lane);
  w.slot = static_cast<int>(lowest >> 26);
  w.row0 = static_cast<int>((lowest >> 13) & 0x1fffu);
  w.col0 = max(0, min(static_cast<int>((lowest & 0x1fffu) >> 6) - 1, W64 - (1 << winShift)));
}

/* A lane's (up to four) pixels of one tile, as keys; all 64 lanes (wavewin_prepare votes).  One 32-bit LDS atomic per pixel
 * (the window's 64-bit words as pairs of halves), no merging of the lane's neighbouring pixels first: with range noise
 * they shared a word only 1.4 to 1, and the bookkeeping for it cost more instructions than the atomics it saved.  A pixel
 * outside the window goes straight to memory (the bounding box of those is kept per wave, wavemiss_*), so the result never
 * depends on where the window is. */
__device__ __forceinline__ void wavewin_emit(unsigned long long *ww, unsigned int *wm, WaveWindow &w, unsigned long long *__restrict__ images,
                                             unsigned int imgWords, int W64, int winShift, ImageBox *boxes, const unsigned int (&key)[4], int lane)
{
  wavewin_prepare(ww, w, images, imgWords, W64, winShift, boxes, min(min(key[0], key[1]), min(key[2], key[3])), lane);
  const unsigned int base = window_base(w);
  unsigned int *ww32 = reinterpret_cast<unsigned int *>(ww);
#pragma unroll
  for(int j = 0; j < 4; j++)
  {
    const unsigned int k = key[j], d = k - base;
    const unsigned int bit = 1u << (k & 31u);
    if(window_hit(d, winShift))
    {
      if(SSD_CHK(3, ((d >> 13) << (winShift + 1)) + ((d & 0x1fffu) >> 5), 2 * kWinWords))
        atomicOr(&ww32[((d >> 13) << (winShift + 1)) + ((d & 0x1fffu) >> 5)], bit);
    }
    else if(k != kNoPixel)
    {
      const unsigned int slot = k >> 26, iy = (k >> 13) & 0x1fffu, ix = k & 0x1fffu, xw = ix >> 6;
      if(SSD_CHK(4, static_cast<unsigned long long>(slot) * imgWords + iy * static_cast<unsigned int>(W64) + xw, w.limitWords) && SSD_CHK(5, xw, W64))
        atomicOr(reinterpret_cast<unsigned int *>(images + (slot * imgWords + iy * static_cast<unsigned int>(W64) + xw)) + ((ix >> 5) & 1u), bit);
      int *b = reinterpret_cast<int *>(wm);
      atomicMin(&b[2], static_cast<int>(iy)); atomicMax(&b[3], static_cast<int>(iy));
      atomicMin(&b[4], static_cast<int>(xw)); atomicMax(&b[5], static_cast<int>(xw));
      atomicOr(&wm[6], 1u << slot);
    }
  }
}

/* Projection2D::worldToImage (pointcloud.cpp:79-83); false = outside the image (quirk Q5) */
__device__ __forceinline__ bool image_pixel(const PointParams &P, const PixelParams &X, double wx, double wy, int &ix, int &iy)
{
  ix = static_cast<int>((wx - P.xMin) * X.xToImage);
  iy = static_cast<int>((P.yMax - wy) * X.yToImage);
  /* two unsigned compares: a negative coordinate is a huge unsigned one */
  return (static_cast<unsigned int>(ix) < static_cast<unsigned int>(X.W)) & (static_cast<unsigned int>(iy) < static_cast<unsigned int>(X.H));
}

/* round(z * 2^40) without a double->int64 conversion (a long software sequence on this ISA): adding
 * 1.5 * 2^12 puts z (|z| < 2048) on the 2^-40 grid of [4096, 8192), rounded to nearest even by the add;
 * the mantissa difference to the constant is the integer.  Same value as llrint(z * 2^40). */
__device__ __forceinline__ long long z_to_fixed(double z)
{
  const double magic = 6144.0;
  return __double_as_longlong(z + magic) - __double_as_longlong(magic);
}
/* the same in two halves for running sums: n values of z_plus_magic_bits minus n * kMagicBits */
__device__ __forceinline__ long long z_plus_magic_bits(double z)
{
  return __double_as_longlong(z + 6144.0);
}
constexpr long long kMagicBits = 0x40B8000000000000ll;       /* bits of 6144.0 */

/* ---- single pass: K1's window spans FOUR planes ----
 * K1 rasters the points of the bins k_predict chose, one bit image ("plane") per height bin: which two adjacent bins make a
 * plateau is only known once the histogram is complete (k_peaks), and range noise spreads a tread over two or three bins
 * whose pixels interleave.  A window over one image would send every pixel of the minority bins to memory; this one
 * holds the same patch of 2^kSpecPlaneBits consecutive planes, 2^kSpecRowBits rows x 8 words each (2 x 16 x 8: 2 KiB per wave;
 * 4 planes x 16 rows, 2 x 32 and 1 x 32 measured the same or slower, 4 x 32 - three blocks per CU - much slower:
 * profiles/r04_single_pass_ab.txt).  Keys as pixel_key with the plane in the slot field, "inside" by one subtraction and one
 * AND as window_hit.  Since k_predict gives a tread whose fuller neighbour bin is beyond doubt ONE plane for both bins, most
 * treads live in a single plane anyway. */
#ifndef SSD_SPEC_PLANE_BITS
#define SSD_SPEC_PLANE_BITS 1
#endif
#ifndef SSD_SPEC_ROW_BITS
#define SSD_SPEC_ROW_BITS 4
#endif
constexpr int kSpecPlaneBits = SSD_SPEC_PLANE_BITS, kSpecRowBits = SSD_SPEC_ROW_BITS;
constexpr int kSpecWinPlanes = 1 << kSpecPlaneBits, kSpecWinRows = 1 << kSpecRowBits, kSpecWinCols = 8;
constexpr int kSpecWinWords = kSpecWinPlanes * kSpecWinRows * kSpecWinCols;
constexpr unsigned int kSpecInside = (static_cast<unsigned int>(kSpecWinPlanes - 1) << 26) | (static_cast<unsigned int>(kSpecWinRows - 1) << 13)
                                     | static_cast<unsigned int>(64 * kSpecWinCols - 1);
static_assert(kSpecWinCols == 8, "the index arithmetic below is written for eight word columns");
static_assert(kMaxPlanes <= 32 - kSpecWinPlanes, "a plane offset that borrows must fall outside the window's planes");

struct SpecWindow
{
  int plane0 = -1, row0 = 0, col0 = 0;   /* wave-uniform */
};
__device__ __forceinline__ unsigned int specwin_base(const SpecWindow &w)
{
  return w.plane0 < 0 ? kNoWindow : pixel_key(w.plane0, w.row0, w.col0 << 6);
}
__device__ __forceinline__ bool specwin_hit(unsigned int d)
{
  return (d & ~kSpecInside) == 0u;
}
/* OR over the wave, result in every lane's SGPR copy: rows by DPP, the four rows by readlane */
__device__ __forceinline__ unsigned int wave_or_u32(unsigned int v)
{
  v = row_or_u32(v);
  return static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(v), 0) | __builtin_amdgcn_readlane(static_cast<int>(v), 16)
                                   | __builtin_amdgcn_readlane(static_cast<int>(v), 32) | __builtin_amdgcn_readlane(static_cast<int>(v), 48));
}
/* Round 6: the window's rows are a RING - image row r of the window's range [row0, row0 + kSpecWinRows) lives in slot
 * r & (kSpecWinRows - 1) - so that the window can SLIDE down the image by half its height, writing out only the rows it leaves,
 * instead of standing still until more than half of a tile's pixels miss it and then being written out whole: measured on the
 * bench's frames, 98 % of the wave-tiles' pixels span fewer than 16 rows, yet 5 % of all pixels missed the standing window
 * "below" - in the tiles before each move - and went to memory one by one (tools/k1count.py, profiles/r06_k1_windows.txt).
 *
 * NROWS rows from image row rBegin (within the window's range) out - non-zero words only - and cleared.  One pass per plane
 * and 8 rows: lane = (row, word column).  The rows and columns that held bits come from the ballot of the non-zero words on the
 * scalar unit (no reductions over the wave), and extend the plane's box.  All 64 lanes. */
template<int NROWS>
__device__ __forceinline__ void specwin_flush_rows(unsigned long long *ww, const SpecWindow &w, const int rBegin, unsigned long long *__restrict__ planes,
                                                   unsigned int imgWords, int W64, ImageBox *boxes, int lane)
{
  static_assert(NROWS % 8 == 0 && NROWS <= kSpecWinRows, "whole passes of 8 rows");
  if(w.plane0 < 0)
    return;
#pragma unroll
  for(int pl = 0; pl < kSpecWinPlanes; pl++)
  {
    unsigned long long *img = planes + static_cast<size_t>(w.plane0 + pl) * imgWords;
#pragma unroll
    for(int k = 0; k < NROWS / 8; k++)
    {
      const int y = rBegin + 8 * k + (lane >> 3), x = w.col0 + (lane & 7);
      const int at = pl * (kSpecWinRows * kSpecWinCols) + ((y & (kSpecWinRows - 1)) << 3) + (lane & 7);
      const unsigned long long v = ww[at];
      const unsigned long long nz = __ballot(v != 0ull);         /* bit 8 r + c: row rBegin + 8 k + r, word column col0 + c */
      SSD_CNT(3, __popcll(nz));
      if(v)
      {
#if !(defined(SSD_ABL) && SSD_ABL == 3)
        if(SSD_CHK(6, static_cast<size_t>(w.plane0 + pl) * imgWords + static_cast<size_t>(y) * W64 + x, static_cast<size_t>(kMaxPlanes) * imgWords) && SSD_CHK(7, x, W64))
          atomicOr(img + static_cast<size_t>(y) * W64 + x, v);
#endif
        ww[at] = 0ull;
      }
      if(nz != 0ull)
      {
        unsigned long long r = nz | (nz >> 4);
        r |= r >> 2;
        r |= r >> 1;
        r &= 0x0101010101010101ull;                               /* bit 8 r: row r held bits */
        unsigned long long c = nz | (nz >> 32);
        c |= c >> 16;
        c |= c >> 8;
        c &= 0xffull;                                             /* bit c: column c held bits */
        const int rLo = (__ffsll(static_cast<long long>(r)) - 1) >> 3, rHi = (63 - __clzll(static_cast<long long>(r))) >> 3;
        const int cLo = __ffsll(static_cast<long long>(c)) - 1, cHi = 63 - __clzll(static_cast<long long>(c));
        if(lane == 0)
        {
          ImageBox &bx = boxes[w.plane0 + pl];
          atomicMin(&bx.yMin, rBegin + 8 * k + rLo); atomicMax(&bx.yMax, rBegin + 8 * k + rHi);
          atomicMin(&bx.xMin, w.col0 + cLo); atomicMax(&bx.xMax, w.col0 + cHi);
        }
      }
    }
  }
}
/* the whole window out (before it is re-anchored, and at the end of the block's loop) */
__device__ __forceinline__ void specwin_flush(unsigned long long *ww, const SpecWindow &w, unsigned long long *__restrict__ planes,
                                              unsigned int imgWords, int W64, ImageBox *boxes, int lane)
{
  specwin_flush_rows<kSpecWinRows>(ww, w, w.row0, planes, imgWords, W64, boxes, lane);
}
/* Before a tile's pixels go out, all 64 lanes; first / last: the lane's lowest and highest pixel_key of this tile (kNoPixel: none;
 * last as a signed number: kNoPixel is -1 and sorts below every key).
 *   - enough lanes (kSpecSlideLanes) whose last pixel lies in the half window BELOW the window - same planes, same columns -
 *     while no more than half of the lanes miss it altogether: the window slides down by half its height (the rows it leaves
 *     are written out);
 *   - more than half of the lanes that have pixels miss it: written out whole and re-anchored - one plane below the lowest
 *     plane of the tile's pixels (a tread's minority bin may lie on either side of the bin seen first), the lowest missing row,
 *     one word left of the lowest missing column. */
constexpr int kSpecSlideLanes = 6;
constexpr int kSpecSlideRows = kSpecWinRows / 2;
static_assert(kSpecSlideRows % 8 == 0, "a slide writes whole passes of 8 rows");
__device__ __forceinline__ void specwin_prepare(unsigned long long *ww, SpecWindow &w, unsigned long long *__restrict__ planes,
                                                unsigned int imgWords, int W64, ImageBox *boxes, unsigned int first, int last, int lane)
{
  const unsigned int base = specwin_base(w);
  const bool has = first != kNoPixel;
  const bool miss = has && !specwin_hit(first - base);
  /* "below": the difference to the origin has the row offset's next bit set and nothing else outside the window's fields, with
   * the offset below 1.5 windows: rows [kSpecWinRows, kSpecWinRows + kSpecSlideRows) */
  const unsigned int dl = static_cast<unsigned int>(last) - base;
  const bool below = last >= 0 && (dl & ~(kSpecInside & ~(static_cast<unsigned int>(kSpecSlideRows) << 13))) == (static_cast<unsigned int>(kSpecWinRows) << 13);
  const unsigned long long missing = __ballot(miss);
  const int nMissing = __popcll(missing), nHas = __popcll(__ballot(has));
  if(2 * nMissing <= nHas)
  {
    if(__popcll(__ballot(below)) >= kSpecSlideLanes)
    {
      SSD_CNT(2, 1);
      specwin_flush_rows<kSpecSlideRows>(ww, w, w.row0, planes, imgWords, W64, boxes, lane);
      w.row0 += kSpecSlideRows;
    }
    return;
  }
  const unsigned int loPlane = __builtin_amdgcn_readfirstlane(wave_min_u32(has ? first >> 26 : 31u));     /* of ALL the tile's pixels: the planes of a tread alternate */
  const unsigned int loRow = __builtin_amdgcn_readfirstlane(wave_min_u32(miss ? (first >> 13) & 0x1fffu : 0x1fffu));
  const unsigned int loCol = __builtin_amdgcn_readfirstlane(wave_min_u32(miss ? first & 0x1fffu : 0x1fffu));
  SSD_CNT(2, 1);
  specwin_flush(ww, w, planes, imgWords, W64, boxes, lane);
  w.plane0 = max(0, min(static_cast<int>(loPlane) - (kSpecWinPlanes > 2 ? 1 : 0), kMaxPlanes - kSpecWinPlanes));
  w.row0 = static_cast<int>(loRow);
  w.col0 = max(0, min(static_cast<int>(loCol >> 6) - 1, W64 - kSpecWinCols));
}
/* The pixels a lane sent straight to memory because they missed the window: their bounding box and planes, in the lane's
 * registers (K2 / K4 keep them per wave in LDS with five LDS atomics per miss, wavemiss_*; here two or three times as many
 * pixels miss - the points of a candidate bin that lie elsewhere in the image, on a wall behind the stairs - and the lane
 * has registers to spare).  specmiss_flush, end of the block's loop: reduced over the wave, into the block's boxes. */
struct SpecMiss
{
  int y0 = 0x7fffffff, y1 = -1, x0 = 0x7fffffff, x1 = -1;
  unsigned int planes = 0u;
};
__device__ __forceinline__ void specmiss_flush(const SpecMiss &m, ImageBox *boxes, int lane)
{
  const unsigned int planes = wave_or_u32(m.planes);
  if(planes == 0u)
    return;
  const int y0 = wave_min_i(m.y0), y1 = wave_max_i(m.y1), x0 = wave_min_i(m.x0), x1 = wave_max_i(m.x1);
  if(lane < 32 && ((planes >> lane) & 1u))
  {
    atomicMin(&boxes[lane].yMin, y0); atomicMax(&boxes[lane].yMax, y1);
    atomicMin(&boxes[lane].xMin, x0); atomicMax(&boxes[lane].xMax, x1);
  }
}
/* as wavewin_emit; the window's word of a pixel: (plane offset, row slot = row & (rows - 1), half word) */
__device__ __forceinline__ void specwin_emit(unsigned long long *ww, SpecMiss &miss, SpecWindow &w, unsigned long long *__restrict__ planes,
                                             unsigned int imgWords, int W64, ImageBox *boxes, const unsigned int (&key)[4], int lane)
{
  specwin_prepare(ww, w, planes, imgWords, W64, boxes, min(min(key[0], key[1]), min(key[2], key[3])),
                  max(max(static_cast<int>(key[0]), static_cast<int>(key[1])), max(static_cast<int>(key[2]), static_cast<int>(key[3]))), lane);
  const unsigned int base = specwin_base(w);
  unsigned int *ww32 = reinterpret_cast<unsigned int *>(ww);
#pragma unroll
  for(int j = 0; j < 4; j++)
  {
    const unsigned int k = key[j], d = k - base;
    const unsigned int bit = 1u << (k & 31u);
    SSD_CNT_LANES(4, k != kNoPixel);
    SSD_CNT_LANES(5, k != kNoPixel && !specwin_hit(d));
    if(specwin_hit(d))
    {
      const unsigned int at = ((d >> 26) << (kSpecRowBits + 4)) | (((k >> 13) & (kSpecWinRows - 1u)) << 4) | ((d & 0x1fffu) >> 5);
      if(SSD_CHK(8, at, 2 * kSpecWinWords))
        atomicOr(&ww32[at], bit);
    }
    else if(k != kNoPixel)
    {
      const unsigned int slot = k >> 26, iy = (k >> 13) & 0x1fffu, ix = k & 0x1fffu, xw = ix >> 6;
      if(SSD_CHK(9, static_cast<unsigned long long>(slot) * imgWords + iy * static_cast<unsigned int>(W64) + xw, static_cast<unsigned long long>(kMaxPlanes) * imgWords) && SSD_CHK(10, xw, W64))
        atomicOr(reinterpret_cast<unsigned int *>(planes + (slot * imgWords + iy * static_cast<unsigned int>(W64) + xw)) + ((ix >> 5) & 1u), bit);
      miss.y0 = min(miss.y0, static_cast<int>(iy)); miss.y1 = max(miss.y1, static_cast<int>(iy));
      miss.x0 = min(miss.x0, static_cast<int>(xw)); miss.x1 = max(miss.x1, static_cast<int>(xw));
      miss.planes |= 1u << slot;
    }
  }
}

/* The streaming kernels are written as block bodies over an explicit LDS struct, (frame, chunk) given by the caller:
 * the kernels below pass blockIdx (tools and experiments have paired two bodies in one launch: DESIGN.md section 3). */

/* ---- round 5: K1's x / y range test in single precision first (PreXY, ssd_device.h; the bound: ssd_prexy.h) ---- */
typedef float f32x2 __attribute__((ext_vector_type(2)));
/* (d.x, d.y): the point's world x / y, centred on the measuring range and divided by its extent - three packed FMAs */
__device__ __forceinline__ f32x2 pre_xy(const PreXY &Q, const f32x2 c3, float x, float y, float z)
{
  f32x2 d = __builtin_elementwise_fma(f32x2{ Q.c[2][0], Q.c[2][1] }, f32x2{ z, z }, c3);
  d = __builtin_elementwise_fma(f32x2{ Q.c[1][0], Q.c[1][1] }, f32x2{ y, y }, d);
  d = __builtin_elementwise_fma(f32x2{ Q.c[0][0], Q.c[0][1] }, f32x2{ x, x }, d);
  return d;
}
/* single instructions with |.| on the operands (as builtins the compiler canonicalises fmaxf's operands first: v_max x, x) */
__device__ __forceinline__ float absmax2(float a, float b)
{
  float r;
  asm("v_max_f32 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float absmax3(float a, float b, float c)
{
  float r;
  asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
/* truncating conversion that saturates (negative and NaN: 0) instead of being undefined */
__device__ __forceinline__ unsigned int cvt_u32_f32(float a)
{
  unsigned int r;
  asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(a));
  return r;
}
__device__ __forceinline__ float min3_f32(float a, float b, float c)
{
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float min_f32(float a, float b)
{
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float max_f32(float a, float b)
{
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
/* the cell's five reductions over the 16 lanes of a DPP row (groups: OR; the box: two minima, two maxima of the lanes' d), in one
 * block of twenty instructions behind a two-cycle no-op, the five chains interleaved so that no DPP operand is read within two instructions of its write
 * (the wait states the hardware asks for; the compiler does not see into the block) */
__device__ __forceinline__ void row_reduce_cell(unsigned int &groups, float &x0, float &x1, float &y0, float &y1)
{
  /* s_nop 1 first: the hazard recogniser does not look into the block, and nothing else guarantees that the instruction in front of
   * it has not just written one of the first two operands (a DPP read within two wait states of the write takes stale lanes) */
  asm("s_nop 1\n\t"
      "v_or_b32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_or_b32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %4, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_or_b32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %2, %2, %2 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_f32_dpp %3, %3, %3 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %4, %4, %4 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_or_b32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_f32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %2, %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_f32_dpp %3, %3, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %4, %4, %4 row_ror:1 row_mask:0xf bank_mask:0xf"
      : "+v"(groups), "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1));
}
/* the cell's box from the extremes of d: rn(d * 256 + boxLo / boxHi) saturated to 0 .. 255, one byte each (ssd_prexy.h) */
__device__ __forceinline__ unsigned int cell_box_from_d(const PreXY &Q, float x0, float x1, float y0, float y1)
{
  /* (x0, x1) and (y0, y1) scaled and shifted by one packed FMA each */
  const f32x2 scale = f32x2{ 256.0f, 256.0f }, shift = f32x2{ Q.boxLo, Q.boxHi };
  const f32x2 bx = __builtin_elementwise_fma(f32x2{ x0, x1 }, scale, shift), by = __builtin_elementwise_fma(f32x2{ y0, y1 }, scale, shift);
  unsigned int box = __builtin_amdgcn_cvt_pk_u8_f32(bx.x, 0u, 0u);
  box = __builtin_amdgcn_cvt_pk_u8_f32(bx.y, 1u, box);
  box = __builtin_amdgcn_cvt_pk_u8_f32(by.x, 2u, box);
  box = __builtin_amdgcn_cvt_pk_u8_f32(by.y, 3u, box);
  return box;
}

/* what the single pass adds to K1's LDS */
struct SpecLds
{
  unsigned long long wins[kThreads / 64][kSpecWinWords];
  ImageBox boxes[kMaxPlanes];
  unsigned char plane[kMaxBins];                 /* FrameState::specPlane */
  unsigned char order[kMaxTilesPerBlock * (kTile / 256)];   /* STRIPS: the chunk's 256-point strips by (column band, index) */
  unsigned int oob[kMaxPlanes];
  unsigned long long ltot[kMaxPlanes][8];        /* sum of round(z * 2^40) per plane (this block's share) */
};
struct NoSpecLds {};

/* The constants of K1's seldom-run piece - the double-precision x / y rows of a point the pre-filter cannot call - live in LDS,
 * copied there once per block: as kernel arguments of the plain k_hist they would sit in scalar registers through the whole
 * point loop for one lane in thousands (with them the loop's own constants were spilled and came back through a dozen
 * v_readlane per point). */
struct K1Consts
{
  double a[9], b[3];                              /* CameraToWorld, all three rows */
  double xMin, xMax, yMin, yMax, zMin, zMax;
  double boxX, boxY;
  double recip;
  double xToImage, yToImage;                      /* SPEC: Projection2D, for the candidate whose pixel single precision cannot call */
};
/* the address of the block's copy, opaque to the compiler at every use: loads from it stay where they are written (hoisted
 * out of the point loop they would occupy thirty-two vector registers for its whole length) */
typedef const K1Consts __attribute__((address_space(3))) *K1ConstsLds;
__device__ __forceinline__ K1ConstsLds k1_consts(const K1Consts &c)
{
  K1ConstsLds p = (K1ConstsLds)(&c);
  asm volatile("" : "+v"(p));
  return p;
}

struct HistLds
{
  K1Consts kc;
  uint2 lInfo[kMaxCellsPerBlock];
  /* [bin][copy]: a vote goes to copy = lane & 31, i.e. LDS bank = copy: the 32 lanes the LDS serves per
   * cycle never collide, whatever their bins (a camera row sweeping one plateau puts all 64 lanes in one bin) */
  alignas(16) unsigned int lh[kMaxBins * kHistCopies];      /* 16-byte aligned: cleared and summed four copies at a time */
  unsigned int lNonZero;
};
static_assert(kHistCopies % 8 == 0 && (kMaxBins * kHistCopies / 4) % kThreads == 0 && kMaxBins * 2 == kThreads,
              "hist_block clears the LDS histogram in 16-byte pieces and sums a bin's copies with two threads");

/* SPEC (single pass): the block also rasters the points of the bins that have a plane (FrameState::specPlane, k_predict) into
 * the frame's planes, as k_raster does for the plateaus' bins: pixel (image_pixel), the plane's z sum and out-of-image count. */
template<int SRC, bool SPEC, bool STRIPS, bool CHECKS, typename SPECLDS>
__device__ __forceinline__ void hist_block(HistLds &L, SPECLDS &SL, const float *__restrict__ xyz, size_t strideFloats, const PointParams &P,
                                           const PreXY &Q, const PixelParams &X, FrameState *__restrict__ st, uint2 *__restrict__ tileMasks,
                                           unsigned long long *__restrict__ planeImg,
                                           size_t tileMaskStride, int chunkPoints, const DepthSrc &D, const int frame, const int chunkIdx)
{
  uint2 (&lInfo)[kMaxCellsPerBlock] = L.lInfo;
  unsigned int (&lh)[kMaxBins * kHistCopies] = L.lh;
  unsigned int &lNonZero = L.lNonZero;

  const int tid = threadIdx.x, lane = tid & 63;
  SpecWindow win;
  SpecMiss missed;
  int curT = -1;
  unsigned long long accT = 0;
  const unsigned int imgWords = static_cast<unsigned int>(X.H) * static_cast<unsigned int>(X.W64);
  unsigned long long *frameImg = nullptr;
  if constexpr(SPEC)
  {
    frameImg = planeImg + static_cast<size_t>(st[frame].planeBase) * imgWords;          /* the frame's planes in the pool (k_predict) */
    const unsigned char planeMine = tid < kMaxBins ? st[frame].specPlane[tid] : static_cast<unsigned char>(0xff);
    if(tid < kMaxBins)
      SL.plane[tid] = planeMine;
    if(tid < kMaxPlanes)
    {
      SL.boxes[tid] = ImageBox{ 0x7fffffff, -1, 0x7fffffff, -1 };
      SL.oob[tid] = 0u;
    }
    if(tid < kMaxPlanes * 8)
      (&SL.ltot[0][0])[tid] = 0ull;
    for(int i = tid; i < (kThreads / 64) * kSpecWinWords; i += kThreads)
      (&SL.wins[0][0])[i] = 0ull;
  }
  /* strideFloats counts floats, or 16-bit depth values for kSrcDepth16 */
  const float *base = SRC == kSrcDepth16
    ? reinterpret_cast<const float *>(reinterpret_cast<const unsigned short *>(xyz) + static_cast<size_t>(frame) * strideFloats)
    : xyz + static_cast<size_t>(frame) * strideFloats;
  const int begin = chunkIdx * chunkPoints;
  const int end = min(begin + chunkPoints, P.nPoints);

  /* (16 bytes per store: a block's prologue and epilogue are ~14 % of K1's vector instructions, round 6) */
  for(int i = tid; i < kMaxBins * kHistCopies / 4; i += kThreads)
    reinterpret_cast<uint4 *>(lh)[i] = make_uint4(0u, 0u, 0u, 0u);
  if(tid == 0)
  {
    lNonZero = 0;
    K1Consts &c = L.kc;
#pragma unroll
    for(int i = 0; i < 9; i++)
      c.a[i] = P.a[i];
    c.b[0] = P.b[0]; c.b[1] = P.b[1]; c.b[2] = P.b[2];
    c.xMin = P.xMin; c.xMax = P.xMax; c.yMin = P.yMin; c.yMax = P.yMax; c.zMin = P.zMin; c.zMax = P.zMax;
    c.boxX = P.boxX; c.boxY = P.boxY;
    c.recip = P.recip;
    if constexpr(SPEC)
    {
      c.xToImage = X.xToImage; c.yToImage = X.yToImage;
    }
  }
  __syncthreads();

  unsigned int *mine = lh + (lane & (kHistCopies - 1));
  unsigned int nz = 0;                                               /* wave-uniform: the count of the whole wave (scalar popcounts) */
  int it = 0;
  int nStore = -1;                                                    /* cell records to store (STRIPS), else it * kCellsPerTile */

  /* the lane's copies of constants that are the SECOND scalar operand of an instruction (one is allowed): the z row's fourth
   * coefficient, the threshold's offset, the x / y rows' fourth pair - without them a v_mov per use and point */
  float zc3 = Q.zc[3], zh0 = Q.zH0;
  f32x2 c3xy = f32x2{ Q.c[3][0], Q.c[3][1] };
  asm volatile("" : "+v"(zc3), "+v"(zh0), "+v"(c3xy));

  /* SPEC: a point of a candidate bin (in range, its bin has plane `plane`): its share of the plane's z sum, its pixel key.
   * projectToBinaryImage (pointcloud.cpp:458-471) for a bin that may turn out a plateau's.  The z sum takes the reference's doubles
   * (world_point_flat's z row).  The pixel (Projection2D::worldToImage, pointcloud.cpp:79-83) comes from the single-precision d of
   * the range test where that is certain (round 6, make_pre_pixel()): px = (d.x + 1/2) W and py = (1/2 - d.y) H farther from every
   * integer than the bound for this point's magnitude truncate to the reference's pixel - and lie inside the image, 0 and W / H
   * being integers; the others take the reference's rows and pixel in doubles. */
  auto candidatePoint = [&](const F3 &q, const f32x2 dj, const int plane) -> unsigned int
  {
    if constexpr(!SPEC)
      return kNoPixel;
    else
    {
    const double x = q.x, y = q.y, z = q.z;
    double wz = (P.a[6] * x + P.a[7] * y) + P.a[8] * z;
    wz = wz + P.b[2];
    const float px = __builtin_fmaf(dj.x, X.fW, X.fHalfW), py = __builtin_fmaf(dj.y, X.fNegH, X.fHalfH);
    const f32x2 gg = f32x2{ __builtin_amdgcn_fractf(px), __builtin_amdgcn_fractf(py) } + f32x2{ -0.5f, -0.5f };      /* one packed add */
    const float hp = __builtin_fmaf(absmax3(q.x, q.y, q.z), X.pxNegK, X.pxH0);
    unsigned int ix = cvt_u32_f32(px), iy = cvt_u32_f32(py);
    bool inside = true;
#if defined(SSD_SABOTAGE_PRE) && (SSD_SABOTAGE_PRE & 2)   /* tools: the band around the pixel edges NOT handed to the doubles */
    if(false)
#else
    if(!(absmax2(gg.x, gg.y) < hp))
#endif
    {
      /* rare: the doubles, and the image's bounds (quirk Q5) */
      const K1ConstsLds c = k1_consts(L.kc);
      double wx = (c->a[0] * x + c->a[1] * y) + c->a[2] * z;
      double wy = (c->a[3] * x + c->a[4] * y) + c->a[5] * z;
      wx = wx + c->b[0];
      wy = wy + c->b[1];
      ix = static_cast<unsigned int>(static_cast<int>((wx - c->xMin) * c->xToImage));
      iy = static_cast<unsigned int>(static_cast<int>((c->yMax - wy) * c->yToImage));
      inside = (ix < static_cast<unsigned int>(X.W)) & (iy < static_cast<unsigned int>(X.H));
      if(!inside && SSD_CHK(14, plane, kMaxPlanes))
        atomicAdd(&SL.oob[plane], 1u);                   /* quirk Q5 */
    }
    if(plane != curT)
    {
      if(curT >= 0 && SSD_CHK(13, curT, kMaxPlanes))
        atomicAdd(&SL.ltot[curT][lane & 7], accT);
      curT = plane;
      accT = 0;
    }
    accT += static_cast<unsigned long long>(z_plus_magic_bits(wz));      /* the bits of z + 6144; k_peaks takes the constant's share off (count x kMagicBits) */
    return inside ? pixel_key(plane, static_cast<int>(iy), static_cast<int>(ix)) : kNoPixel;
    }
  };

  auto tileBody = [&](const F3 (&v)[kPts], const int cellAt)
  {
    unsigned int groups = 0u;
    /* extremes of d (pre_xy) over the lane's in-range points: the cell's box */
    float x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
    unsigned int planes[kPts] = { 0xffu, 0xffu, 0xffu, 0xffu };     /* SPEC: the plane of the point's bin (0xff: none, or the point is out of range) */
    /* SPEC: the points' d, for their pixels - kept where the registers are there (the tile loop on vertex input), made again otherwise */
    constexpr bool kKeepD = SPEC && !STRIPS && SRC != kSrcDepth16;
    f32x2 dk[kPts];
    unsigned int keys[kPts] = { kNoPixel, kNoPixel, kNoPixel, kNoPixel };
#pragma unroll
    for(int j = 0; j < kPts; j++)
    {
      const F3 p{ v[j].x, v[j].y, v[j].z };
      /* The tests' outcomes are kept as the wave's 64-bit lane masks (one v_cmp each, straight into a scalar register pair) and combined
       * on the scalar unit; as `bool`s the compiler evaluated two of them in BOTH polarities - a second v_cmp each, 25 M vector
       * instructions per launch - where one s_andn2 does.  __builtin_amdgcn_inverse_ballot_w64 turns a mask back into the lanes' branch. */
      /* pointcloud.cpp:143-146, counted per wave on the scalar unit */
      const unsigned long long mValid = __ballot(p.z > 0.0f);
      nz += static_cast<unsigned int>(__popcll(mValid));
      /* x / y in single precision (round 5): inside for sure, outside for sure (M > hi), or the band between them (and NaNs) */
      f32x2 d = pre_xy(Q, c3xy, p.x, p.y, p.z);
      const float M = absmax2(d.x, d.y);
      unsigned long long mInxy = __ballot(M < Q.lo);
      unsigned long long mMaybexy = ~__ballot(M > Q.hi);
      const float M3 = absmax3(p.x, p.y, p.z);
      if constexpr(CHECKS)
      {
        /* unless make_pre_xy() showed that larger inputs cannot read "inside" */
        const unsigned long long mFar = Q.checkInput ? ~__ballot(M3 <= Q.maxInput) : 0ull;
        mInxy &= ~mFar;
        mMaybexy |= mFar;
      }
      /* z in single precision (round 6, make_pre_z()): t = the height above zMin in bins.  Farther from every integer than the
       * bound for this point's magnitude: the bin is floor(t) and the z range is 0 < t < zTop, as the doubles would say */
      const float t = __builtin_fmaf(Q.zc[0], p.x, __builtin_fmaf(Q.zc[1], p.y, __builtin_fmaf(Q.zc[2], p.z, zc3)));
      const float g = __builtin_amdgcn_fractf(t) - 0.5f;
      const float h = __builtin_fmaf(M3, Q.zNegK, zh0);
      unsigned long long mSurez = __ballot(__builtin_fabsf(g) < h);      /* not for a NaN, nor for a magnitude whose bound exceeds half a bin */
      const unsigned long long mInz = __ballot(__float_as_uint(t) < Q.zTopBits);     /* +0 <= t < zTop on the bits (a negative t has the sign bit) */
      if constexpr(CHECKS)
      {
        if(Q.zCheckTop)
          mSurez &= __ballot(__builtin_fabsf(t - Q.zTop) > 0.5f - h);    /* the range's upper end is no bin edge: its own band */
      }
#if defined(SSD_SABOTAGE_PRE) && (SSD_SABOTAGE_PRE & 1)   /* tools: the band around the bin edges NOT handed to the doubles - the tests built for it must fail */
      mSurez = ~0ull;
#endif
      unsigned int b = cvt_u32_f32(t);
      /* in range for sure; or possibly in range - neither test says "outside for sure" - with a test unsure: those take the doubles */
      const unsigned long long mInSure = mValid & mInz & mInxy & mSurez;
      const unsigned long long mSlow = mValid & mMaybexy & (~mSurez | (mInz & ~mInxy));
      unsigned long long mIn = mInSure;
      if(mSlow != 0ull)
      {
        /* rare (one wave slot in fifty; a wave-uniform branch, so that the mask stays a scalar): the reference's arithmetic, all of it -
         * world_point_flat's rows and compares, height_bin - evaluated by the whole wave, taken by the lanes it is for */
        const K1ConstsLds c = k1_consts(L.kc);
        const double x = p.x, y = p.y, z = p.z;
        double wx = (c->a[0] * x + c->a[1] * y) + c->a[2] * z;
        double wy = (c->a[3] * x + c->a[4] * y) + c->a[5] * z;
        double wz = (c->a[6] * x + c->a[7] * y) + c->a[8] * z;
        wx = wx + c->b[0];
        wy = wy + c->b[1];
        wz = wz + c->b[2];
        mIn |= mSlow & __ballot((wx > c->xMin) & (wx < c->xMax) & (wy > c->yMin) & (wy < c->yMax) & (wz > c->zMin) & (wz < c->zMax));
        const bool mine = __builtin_amdgcn_inverse_ballot_w64(mSlow);
        const unsigned int bD = static_cast<unsigned int>(static_cast<int>((wz - c->zMin) * c->recip));      /* height_bin; meaningless for a point out of range, as is d */
        const float dxD = static_cast<float>((wx - c->xMin) * c->boxX * 0.00390625 - 0.5);
        const float dyD = static_cast<float>((wy - c->yMin) * c->boxY * 0.00390625 - 0.5);
        b = mine ? bD : b;
        d.x = mine ? dxD : d.x;
        d.y = mine ? dyD : d.y;
      }
      if(__builtin_amdgcn_inverse_ballot_w64(mIn))
      {
        if(SSD_CHK(11, b, P.nBins))
          atomicAdd(mine + b * kHistCopies, 1u);                          /* ++hist[bin], pointcloud.cpp:199-202 */
        groups |= 1u << (b / static_cast<unsigned int>(kBinsPerGroup));
        x0 = min_f32(x0, d.x); x1 = max_f32(x1, d.x);
        y0 = min_f32(y0, d.y); y1 = max_f32(y1, d.y);
        if constexpr(SPEC)
        {
          const unsigned int plane = SSD_CHK(12, b, kMaxBins) ? SL.plane[b] : 0xffu;
          if constexpr(STRIPS)
          {
            if(plane != 0xffu)
              keys[j] = candidatePoint(p, d, static_cast<int>(plane));
          }
          else
            planes[j] = plane;
        }
      }
      if constexpr(kKeepD)
        dk[j] = d;
    }
    if constexpr(SPEC && !STRIPS)
    {
      /* The points of the candidate bins, behind the tile's other work: one wave-uniform test per TILE skips all of it where no lane
       * holds one - most tiles (ground, risers, background).  A plane number is below 0xff, so the AND of the four is 0xff only
       * if all are.  (The sorted strips' walk - FHD stress: nearly every strip holds some - takes them point by point above: it
       * has no registers for a second loop.) */
      const unsigned int all4 = planes[0] & planes[1] & planes[2] & planes[3];
      SSD_CNT(0, 1);
      SSD_CNT(1, __ballot(all4 != 0xffu) != 0ull ? 1 : 0);
      SSD_CNT_LANES(6, planes[0] != 0xffu); SSD_CNT_LANES(6, planes[1] != 0xffu); SSD_CNT_LANES(6, planes[2] != 0xffu); SSD_CNT_LANES(6, planes[3] != 0xffu);
#if defined(SSD_ABL) && SSD_ABL == 2                  /* tools: timing without the candidates' work (results are wrong) */
      if(false)
#else
      if(__ballot(all4 != 0xffu) != 0ull)
#endif
      {
#pragma unroll
        for(int j = 0; j < kPts; j++)
        {
          const int plane = static_cast<int>(planes[j]);
          unsigned int key = kNoPixel;
          if(plane != 0xff)
            key = candidatePoint(v[j], kKeepD ? dk[j] : pre_xy(Q, c3xy, v[j].x, v[j].y, v[j].z), plane);
          keys[j] = key;
        }
#if defined(SSD_ABL) && SSD_ABL == 1                  /* tools: timing without the window (results are wrong) */
        asm volatile("" :: "v"(keys[0]), "v"(keys[1]), "v"(keys[2]), "v"(keys[3]));
#else
        specwin_emit(SL.wins[tid >> 6], missed, win, frameImg, imgWords, X.W64, SL.boxes, keys, lane);
#endif
      }
    }
    if constexpr(SPEC && STRIPS)
      specwin_emit(SL.wins[tid >> 6], missed, win, frameImg, imgWords, X.W64, SL.boxes, keys, lane);
    row_reduce_cell(groups, x0, x1, y0, y1);
    if((lane & 15) == 0 && SSD_CHK(15, cellAt, kMaxCellsPerBlock))    /* cell = 64 consecutive points = lanes 16q .. 16q+15 */
      lInfo[cellAt] = make_uint2(groups, cell_box_from_d(Q, x0, x1, y0, y1));
  };
  if constexpr(STRIPS)
  {
    /* Single pass at a width that does not divide a tile (FHD, VGA ..).  Walking the chunk tile by tile, a wave's 256 points
     * would sit in another band of camera columns every tile - its image window would be flushed and re-anchored every tile.
     * So the chunk's strips of 256 points are sorted by (column band of 256 pixels, index) and each wave takes a quarter of
     * the sorted list: it walks down one band, then part of the next.  Sorted by one wave with ballots (<= 128 strips,
     * <= 32 bands).  A strip that crosses the end of a camera row puts its tail into the first band: those pixels miss the
     * window and go to memory one by one - one strip in W / 256. */
    const int nStrips = (end - begin + 255) >> 8;
    if(tid < 64)
    {
      const int sLo = tid, sHi = tid + 64;
      const int bLo = sLo < nStrips ? ((begin + (sLo << 8)) % X.W) >> 8 : -1, bHi = sHi < nStrips ? ((begin + (sHi << 8)) % X.W) >> 8 : -1;
      const int nBands = (X.W + 255) >> 8;
      const unsigned long long below = (1ull << tid) - 1ull;
      int placed = 0;
      for(int b = 0; b < nBands; b++)
      {
        const unsigned long long mLo = __ballot(bLo == b), mHi = __ballot(bHi == b);
        if(bLo == b)
          SL.order[placed + __popcll(mLo & below)] = static_cast<unsigned char>(sLo);
        if(bHi == b)
          SL.order[placed + __popcll(mLo) + __popcll(mHi & below)] = static_cast<unsigned char>(sHi);
        placed += __popcll(mLo) + __popcll(mHi);
      }
    }
    __syncthreads();
    const int wave = tid >> 6;
    int k = wave * nStrips / kWavesPerBlock;
    const int kEnd = (wave + 1) * nStrips / kWavesPerBlock;
    if(k < kEnd)
    {
      /* loads by clamped addresses and the first look at them behind the strip before (first_use): the next strip's loads stay
       * in flight while this one is worked on; only a frame's last strip can reach beyond `end` (block-uniform test) */
      F3 va[kPts], vb[kPts];
      int sa = SL.order[k], sb = 0;
      const int lastStrip = SRC != kSrcDepth16 && ((end - begin) & 255) != 0 ? nStrips - 1 : -1;
      auto issue = [&](const int strip, F3 (&v)[kPts])
      {
        if constexpr(SRC == kSrcDepth16)
          load_points<SRC>(base, begin + (strip << 8) + kPts * lane, end, v, D);         /* deprojected where it is loaded */
        else
          load_points_clamped<SRC>(base, begin + (strip << 8) + kPts * lane, end, v);
      };
      issue(sa, va);
      while(true)
      {
        const bool more = k + 1 < kEnd;
        if(more)
        {
          sb = SL.order[k + 1];
          issue(sb, vb);
        }
        if(sa == lastStrip)
          zero_beyond(va, begin + (sa << 8) + kPts * lane, end);
        tileBody(va, sa * 4 + (lane >> 4));
        if(!more)
          break;
        first_use(vb);
        k++;
        sa = sb;
#pragma unroll
        for(int j = 0; j < kPts; j++)
          va[j] = vb[j];
      }
    }
    nStore = nStrips * (256 / kCell);           /* every strip's four cells were written */
  }
  else if(SRC == kSrcDepth16)
  {
    /* The depth stream: 8 bytes per lane and tile, and the maps.  A tile is 1024 consecutive pixels, so from tile to tile a
     * lane's row advances by 1024 / W and its column by 1024 % W (wrapping once at most): no division in the loop, and when
     * 1024 % W == 0 (XGA: a tile is one image row) the lane's four x-map entries never change - loaded once per block.  The
     * next tile's pixels and map values are requested before the current tile is processed, as in SSD_STREAM_LOOP.
     * (Round 3 went through load_points: per tile a division, five dependent map loads, ~90 instructions; K1 on depth input is
     * bound by instruction issue, not by its 2 bytes per point.) */
    const unsigned short *depth = reinterpret_cast<const unsigned short *>(base);
    const int dRow = kTile / D.W, dCol = kTile - dRow * D.W;
    int idx = begin + kPts * tid;
    int row = depth_row(D, idx), col = idx - row * D.W;
    float4 xm = *reinterpret_cast<const float4 *>(D.xmap + col);
    float ym = D.ymap[min(row, D.H - 1)];
    uint2 raw = idx < end ? *reinterpret_cast<const uint2 *>(depth + idx) : make_uint2(0u, 0u);
    while(true)
    {
      const bool more = idx - kPts * tid + kTile < end;                    /* block-uniform */
      uint2 rawN = make_uint2(0u, 0u);
      float4 xmN = xm;
      float ymN = ym;
      if(more)
      {
        idx += kTile;
        row += dRow; col += dCol;
        if(col >= D.W) { col -= D.W; row++; }
        if(idx < end)
          rawN = *reinterpret_cast<const uint2 *>(depth + idx);
        ymN = D.ymap[min(row, D.H - 1)];
        if(dCol != 0)
          xmN = *reinterpret_cast<const float4 *>(D.xmap + col);
      }
      F3 v[kPts];
      deproject4(raw, xm, ym, D.depthUnits, v);
      tileBody(v, it * kCellsPerTile + (tid >> 4));
      it++;
      if(!more)
        break;
      raw = rawN; xm = xmN; ym = ymN;
    }
  }
  else
  {
    auto tileInOrder = [&](const F3 (&v)[kPts])
    {
      tileBody(v, it * kCellsPerTile + (tid >> 4));
      it++;
    };
#ifndef SSD_K1_TWO_BODIES
    /* One copy of the body.  The next tile's loads go out before the current tile is processed and are first looked at behind it
     * (first_use: until round 5's end the compiler waited for them right where they were issued - see load_points_full()), then
     * moved into place: twelve moves per tile.  Tiles wholly inside the chunk take the branch-free loads; a frame that ends in a
     * part of a tile gets that part by load_points().  Measured against two copies taken in turn (SSD_STREAM_LOOP, which keeps only
     * one of its two prefetches in flight: the register allocator copies the other out of the loaded tuples at once), 1024 XGA
     * frames: single pass 2.06 against 2.18 ms, two passes 1.59 against 1.60. */
    {
      const int fullEnd = begin + (end - begin) / kTile * kTile;
      F3 va[kPts], vb[kPts];
      if(begin < fullEnd)
        load_points_full<SRC>(base, begin + kPts * tid, va);
      for(int i0 = begin; i0 < fullEnd; i0 += kTile)
      {
        const bool more = i0 + kTile < fullEnd;
        if(more)
          load_points_full<SRC>(base, i0 + kTile + kPts * tid, vb);
        tileInOrder(va);
        if(more)
        {
          first_use(vb);
#pragma unroll
          for(int j = 0; j < kPts; j++)
            va[j] = vb[j];
        }
      }
      if(fullEnd < end)
      {
        load_points<SRC>(base, fullEnd + kPts * tid, end, va, D);
        tileInOrder(va);
      }
    }
#else
    SSD_STREAM_LOOP(tileInOrder)
#endif
  }

  if constexpr(SPEC)
  {
    specwin_flush(SL.wins[tid >> 6], win, frameImg, imgWords, X.W64, SL.boxes, lane);
    specmiss_flush(missed, SL.boxes, lane);
    if(curT >= 0 && SSD_CHK(18, curT, kMaxPlanes))
      atomicAdd(&SL.ltot[curT][lane & 7], accT);
  }
  if(lane == 0 && nz)
    atomicAdd(&lNonZero, nz);
  __syncthreads();

  FrameState &fs = st[frame];
  if constexpr(SPEC)
  {
    if(tid < kMaxPlanes)
    {
      unsigned long long t = 0;
#pragma unroll
      for(int k = 0; k < 8; k++)
        t += SL.ltot[tid][k];
      if(t)
        atomicAdd(reinterpret_cast<unsigned long long *>(&fs.planeTotZ[tid]), t);
      if(SL.boxes[tid].yMax >= 0)
      {
        atomicMin(&fs.planeYMin[tid], SL.boxes[tid].yMin); atomicMax(&fs.planeYMax[tid], SL.boxes[tid].yMax);
        atomicMin(&fs.planeXMin[tid], SL.boxes[tid].xMin); atomicMax(&fs.planeXMax[tid], SL.boxes[tid].xMax);
      }
      if(SL.oob[tid])
        atomicAdd(&fs.planeOob[tid], SL.oob[tid]);
    }
  }
  {
    /* the bins' copies summed by all four waves: two threads per bin, half of the copies each in 16-byte reads (taken in an order
     * that changes from bin to bin: the bins' rows lie a whole number of bank cycles apart), the halves joined by one DPP step.
     * (Round 5: one thread per bin and 32 four-byte reads - the work of two waves of the four.) */
    const int b = tid >> 1, half = tid & 1;
    unsigned int s = 0;
    const uint4 *row = reinterpret_cast<const uint4 *>(lh + b * kHistCopies + half * (kHistCopies / 2));
#pragma unroll
    for(int k = 0; k < kHistCopies / 8; k++)
    {
      const uint4 q = row[(k + b) & (kHistCopies / 8 - 1)];
      s += (q.x + q.y) + (q.z + q.w);
    }
    s += static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(s), 0xb1, 0xf, 0xf, false));     /* quad_perm [1, 0, 3, 2]: the neighbour's half */
    if(half == 0 && b < P.nBins && s)
      atomicAdd(&fs.histAcc[b], s);
  }
  if(tid == 0 && lNonZero)
    atomicAdd(&fs.nNonZeroAcc, lNonZero);
  /* the block's cell records, in one burst */
  {
    uint2 *dst = tileMasks + static_cast<size_t>(frame) * tileMaskStride + static_cast<size_t>(begin / kCell);
    const int n = nStore >= 0 ? nStore : it * kCellsPerTile;
    for(int i = tid; i < n; i += kThreads)
      if(SSD_CHK(16, static_cast<size_t>(begin / kCell) + i, tileMaskStride) && SSD_CHK(17, i, kMaxCellsPerBlock))
#ifdef SSD_NT_STORES          /* tools: the records written past the caches */
        {
          __builtin_nontemporal_store(lInfo[i].x, &dst[i].x);
          __builtin_nontemporal_store(lInfo[i].y, &dst[i].y);
        }
#else
        dst[i] = lInfo[i];
#endif
  }
}

template<int SRC, bool CHECKS>
__global__ __launch_bounds__(kThreads, 6) void k_hist(const float *__restrict__ xyz, size_t strideFloats, PointParams P, PreXY Q,
                                                   FrameState *__restrict__ st, uint2 *__restrict__ tileMasks,
                                                   size_t tileMaskStride, int chunkPoints, DepthSrc D)
{
  __shared__ HistLds L;
  NoSpecLds none;
  hist_block<SRC, false, false, CHECKS>(L, none, xyz, strideFloats, P, Q, PixelParams{}, st, tileMasks, nullptr, tileMaskStride, chunkPoints, D, blockIdx.x, blockIdx.y);   /* frame on the fast grid axis: see launch note on XCD balance */
}

/* K1 of a single-pass batch: histogram, cell records AND the planes of the candidate bins.  31 KiB of LDS: five blocks per CU
 * (4 .. 6 measure the same). */
#ifndef SSD_K1S_WAVES
#define SSD_K1S_WAVES 5
#endif
template<int SRC, bool STRIPS, bool CHECKS>
__global__ __launch_bounds__(kThreads, SSD_K1S_WAVES) void k_hist_planes(const float *__restrict__ xyz, size_t strideFloats, PointParams P, PreXY Q, PixelParams X,
                                                   FrameState *__restrict__ st, uint2 *__restrict__ tileMasks,
                                                   unsigned long long *__restrict__ planeImg,
                                                   size_t tileMaskStride, int chunkPoints, DepthSrc D)
{
  __shared__ HistLds L;
  __shared__ SpecLds SL;
#ifdef SSD_K1_ROTATE      /* tools: frames walk their chunks in different orders, so that a CU holds tread rows and ground rows at once */
  const int chunkIdx = static_cast<int>((blockIdx.y + blockIdx.x * SSD_K1_ROTATE) % gridDim.y);
  const int frameIdx = static_cast<int>(blockIdx.x);
#elif defined(SSD_K1_ORDER)   /* tools: other walks of the (frame, chunk) grid than "frame fast" - 1: a frame's chunks in a row (over all XCDs); 2: each XCD a frame at a time */
  const int nF = static_cast<int>(gridDim.x), nC = static_cast<int>(gridDim.y);
  const int lin = static_cast<int>(blockIdx.y) * nF + static_cast<int>(blockIdx.x);
  int frameIdx, chunkIdx;
  if(SSD_K1_ORDER == 2 && (nF & 7) == 0)
  {
    const int xcd = lin & 7, i = lin >> 3;
    frameIdx = xcd + 8 * (i / nC);
    chunkIdx = i % nC;
  }
  else
  {
    frameIdx = lin / nC;
    chunkIdx = lin % nC;
  }
#else
  const int chunkIdx = static_cast<int>(blockIdx.y);
  const int frameIdx = static_cast<int>(blockIdx.x);
#endif
  hist_block<SRC, true, STRIPS, CHECKS>(L, SL, xyz, strideFloats, P, Q, X, st, tileMasks, planeImg, tileMaskStride, chunkPoints, D, frameIdx, chunkIdx);
}

/* K0 of a single-pass batch: which height bins may belong to a step plateau?  A histogram of one cell in every kSpecSample (a
 * different column of the camera image from row to row), the reference's peak filter (pointcloud.cpp:243-256) on it with
 * slack, a plane for each candidate peak's bin and its two neighbours (the plateau takes the peak bin and the fuller
 * neighbour: only the complete histogram decides which).  Nothing here has to be right: k_peaks checks the planes against the
 * plateaus it finds in the complete histogram, and a frame whose plateaus are not covered is rastered by k_raster.
 * Grid (frame, part): the blocks of a frame add their counts into FrameState::predHist; the last one to finish makes the table
 * and leaves the accumulators zero.  sabotage (tests): 1 = planes three bins above the right ones, 2 = no planes. */
template<int SRC>
__global__ __launch_bounds__(kThreads) void k_predict(const float *__restrict__ xyz, size_t strideFloats, PointParams P,
                                                      FrameState *__restrict__ st, DepthSrc D, int minHeight, int sabotage, int *__restrict__ fallback, int poolPlanes)
{
  __shared__ unsigned int sh[kMaxBins + 2];
  __shared__ unsigned int shc[kMaxBins * kHistCopies];        /* [bin][copy], as K1's: the lanes of a wave mostly vote for ONE bin */
  __shared__ int sLast;
  const int frame = blockIdx.x, part = blockIdx.y, nParts = gridDim.y, tid = threadIdx.x;
  const float *base = SRC == kSrcDepth16
    ? reinterpret_cast<const float *>(reinterpret_cast<const unsigned short *>(xyz) + static_cast<size_t>(frame) * strideFloats)
    : xyz + static_cast<size_t>(frame) * strideFloats;
  if(tid < kMaxBins + 2)
    sh[tid] = 0u;
  for(int i = tid; i < kMaxBins * kHistCopies; i += kThreads)
    shc[i] = 0u;
  __syncthreads();
  unsigned int *mine = shc + (tid & (kHistCopies - 1));
  /* the sample: runs of 16 consecutive points (four lanes' loads, 192 B), one of every kSpecSample runs, at a place in its group
   * of runs that changes from group to group.  (Whole cells of 64 points were tried first: neighbouring pixels share a height
   * bin, so a bin's count is really a count of runs - with 64-point runs a background bin's 150 samples were five runs and
   * one bin in six passed the peak filter by chance.) */
#ifndef SSD_K0_RUN
#define SSD_K0_RUN 16
#endif
  constexpr int kRun = SSD_K0_RUN, kLanesPerRun = kRun / kPts;
  const int nRuns = (P.nPoints + kRun - 1) / kRun;
  const int nGroups = (nRuns + kSpecSample - 1) / kSpecSample;
  for(int g = part * (kThreads / kLanesPerRun) + tid / kLanesPerRun; g < nGroups; g += nParts * (kThreads / kLanesPerRun))
  {
    const int run = g * kSpecSample + ((g * 5) & (kSpecSample - 1));
    F3 v[kPts];
    load_points<SRC>(base, run * kRun + kPts * (tid % kLanesPerRun), P.nPoints, v, D);
#pragma unroll
    for(int j = 0; j < kPts; j++)
    {
      double wx, wy, wz;
      if(world_point_flat(P, v[j], wx, wy, wz))
        atomicAdd(mine + height_bin(P, wz) * kHistCopies, 1u);
    }
  }
  __syncthreads();
  FrameState &fs = st[frame];
  if(tid < P.nBins)
  {
    unsigned int c = 0;
#pragma unroll
    for(int k = 0; k < kHistCopies; k++)
      c += shc[tid * kHistCopies + ((k + tid) & (kHistCopies - 1))];      /* rotated: conflict-free */
    /* The adds are taken WITH their results: a thread that holds the old value knows its add has been performed (device-scope
     * atomics act on the memory side, as does the exchange that reads the sums below), so after the barrier the block's
     * share is in place and its ticket may be drawn - without a __threadfence(), which on this part writes the whole L2
     * back: with two of them per block this kernel took 1.0 ms instead of 0.1. */
    if(c)
    {
      const unsigned int before = atomicAdd(&fs.predHist[tid], c);
      asm volatile("" :: "v"(before));
    }
  }
  __syncthreads();
  /* (Relaxed atomics throughout: the hand-over rests on returning device-scope atomics being performed at the memory side of
   * THIS part - gfx950, one L2 per XCD in front of the fabric's atomics -, not on the HIP memory model.  The build is pinned
   * to that part below, and the kernel's table is held against ssd_predict.h on every frame of four batches in the default
   * GPU tier: tests/test_gpu_single_pass.py::test_the_kernels_table_is_the_host_statements.  A table that differed would
   * cost the frame its planes - k_peaks checks them and k_raster steps in -, never a result.) */
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_predict's hand-over between blocks is written for gfx950 (see the comment above)"
#endif
  if(tid == 0)
    sLast = atomicAdd(&fs.predDone, 1u) == static_cast<unsigned int>(nParts - 1) ? 1 : 0;
  __syncthreads();
  if(!sLast)
    return;
  /* the frame's sample is complete: sh[1 + bin] (a zero either side) */
  const unsigned int mineCount = tid < P.nBins ? atomicExch(&fs.predHist[tid], 0u) : 0u;
  __syncthreads();
  if(tid < kMaxBins)
  {
    sh[1 + tid] = mineCount;
    fs.predSample[tid] = mineCount;
  }
  if(tid == 0)
    sh[0] = sh[kMaxBins + 1] = 0u;
  __syncthreads();
  /* (the table as a plain function of the sample: ssd_predict.h - the tests hold this kernel's table against it) */
  /* candidate peak: filterPeaks' two conditions (>= 2000 points; twice the count exceeds the neighbours' sum by more than
   * half the count) on the sample: from 1200 points scaled up, the sharpness as it is plus a few samples.  (A tread stands
   * far above both thresholds.  Looser - the neighbours' sum below 1.75 counts - and a flat background of a few thousand
   * points per bin, a wall behind the stairs, passes in every sixth of its bins on sampling noise alone: more candidates
   * than planes.) */
  auto candidate = [&](int b) -> bool
  {
    if(b < max(minHeight, 1) || b >= P.nBins - 1)
      return false;
    const unsigned int c = sh[1 + b], l = sh[b], r = sh[2 + b];
    return c > l && c >= r && c * kSpecSample >= 1200u && (l + r) * 2u < 3u * c + 16u;     /* a local maximum: of a tread's two almost equally full bins, one */
  };
  /* at most kMaxPlanes / 3 candidates: the fullest ones (a sampled background still throws up a false peak here and there;
   * they cost planes and stray raster work, never results) */
  __shared__ unsigned int sCand[kMaxBins];
  const bool cand = tid < kMaxBins && sabotage != 2 && candidate(tid);
  /* The plateau is the peak's bin and the fuller of its neighbours (extractPlateauPoints, pointcloud.cpp:300-335; the upper
   * one on a tie).  Where the sample leaves no doubt which - one neighbour more than twice the other - the other gets no
   * plane: it holds no tread, only what lies at that height elsewhere in the image, points that would miss the windows. */
  unsigned int code = 0u;
  if(cand)
  {
    const unsigned int l = sh[tid], r = sh[2 + tid];
    code = 1u | (r > 2u * l + 8u ? 0u : 2u) | (l > 2u * r + 8u ? 0u : 4u);        /* chosen | lower neighbour | upper neighbour */
  }
  if(tid < kMaxBins)
    sCand[tid] = cand ? sh[1 + tid] : 0u;
  __syncthreads();
  if(tid < kMaxBins)
  {
    int fuller = 0;
    if(cand)
      for(int b = 0; b < kMaxBins; b++)
        fuller += (sCand[b] > sCand[tid] || (sCand[b] == sCand[tid] && b < tid)) ? 1 : 0;
    sh[1 + tid] = cand && fuller < kMaxPlanes / 3 ? code : 0u;        /* the table of chosen peaks, zeros either side as before */
  }
  __syncthreads();
  /* Bins to planes.  A peak whose fuller neighbour is beyond doubt shares ONE plane with it - the plateau's image as k_raster
   * would make it, the tread's pixels in one window however the range noise deals its points to the two bins; a peak with
   * neighbours too alike to call gets three planes (its own holds the tread; k_outline reads it together with the
   * neighbour k_peaks picks).  code(k): the chosen-peak code of bin k (0: none), shifted by the sabotage. */
  const int shift = sabotage == 1 ? 3 : 0;
  auto code_of = [&](int k) -> unsigned int
  {
    k -= shift;
    return k >= 0 && k < kMaxBins ? sh[1 + k] : 0u;
  };
  bool want = false, start = false;
  if(tid < kMaxBins)
  {
    auto wanted = [&](int k) { return k >= 0 && k < P.nBins && ((code_of(k - 1) & 4u) | (code_of(k) & 1u) | (code_of(k + 1) & 2u)) != 0u; };
    want = wanted(tid);
    /* shares the plane of the bin below: the upper neighbour of a peak sure of it, or a peak sure of its lower neighbour */
    const bool withBelow = want && wanted(tid - 1) && (code_of(tid - 1) == 5u || code_of(tid) == 3u);
    start = want && !withBelow;
  }
  __shared__ unsigned long long sStart[2];
  if(tid < 2 * 64)
  {
    const unsigned long long m = __ballot(start);
    if((tid & 63) == 0)
      sStart[tid >> 6] = m;
  }
  __syncthreads();
  const int total = __popcll(sStart[0]) + __popcll(sStart[1]);
  /* the frame's planes out of the workspace's pool: one returning add on the batch's counter (fallback[2]; k_peaks of this batch
   * leaves it zero again).  A frame the pool cannot serve - more candidates per frame than kPoolPlanesPerFrame over a whole
   * batch - gets no planes: k_peaks will find its plateaus uncovered and k_raster does it. */
  __shared__ int sBase;
  if(tid == 0)
  {
    int base = -1;
    if(total > 0 && total <= kMaxPlanes)
    {
      base = atomicAdd(&fallback[2], total);
      if(base + total > poolPlanes)
        base = -1;
    }
    sBase = base;
  }
  __syncthreads();
  const bool fits = sBase >= 0;
  if(tid < kMaxBins)
  {
    /* plane = starts at or below this bin, minus one */
    const unsigned long long upTo = tid < 64 ? (2ull << tid) - 1ull : ~0ull;
    const int n = tid < 64 ? __popcll(sStart[0] & upTo) : __popcll(sStart[0]) + __popcll(sStart[1] & ((2ull << (tid - 64)) - 1ull));
    fs.specPlane[tid] = (want && fits) ? static_cast<unsigned char>(n - 1) : static_cast<unsigned char>(0xff);
  }
  if(tid < kMaxPlanes)
  {
    fs.planeYMin[tid] = 0x7fffffff; fs.planeYMax[tid] = -1;
    fs.planeXMin[tid] = 0x7fffffff; fs.planeXMax[tid] = -1;
    fs.planeTotZ[tid] = 0;
    fs.planeOob[tid] = 0u;
    fs.planeUsed[tid] = 0;
  }
  if(tid == 0)
  {
    fs.nPlanes = fits ? total : 0;
    fs.planeBase = fits ? sBase : 0;
    fs.predDone = 0u;
    if(frame == 0)
    {
      fallback[0] = 0;                   /* k_raster's work list of this batch starts empty (k_peaks appends) */
      fallback[1] = 0;                   /* frames without step plateaus (k_peaks counts) */
    }
  }
}

/* ========================================================================= */
/* K1b: peaks, plateaus, LUT — one thread per frame (121 bins: trivial)        */

/* one wave per frame: the histogram is staged in LDS, lane 0 walks it (the peak / plateau logic is a
 * sequential scan with carried state), all lanes write the tables out */
__global__ __launch_bounds__(64) void k_peaks(Params P, FrameState *__restrict__ st, int nframes, DebugFrame *__restrict__ dbg, int spec, int *__restrict__ fallback)
{
  __shared__ unsigned char sImgPlane[kMaxStepImages][2];
  __shared__ unsigned int sCovered;
  __shared__ unsigned int hist[kMaxBins + 1];
  __shared__ unsigned char lut[kMaxBins];
  __shared__ int plPeak[kMaxPlateaus], plLo[kMaxPlateaus], plHi[kMaxPlateaus], plEffLo[kMaxPlateaus], plEffHi[kMaxPlateaus], plN[kMaxPlateaus];
  __shared__ int dbgPeaks[kMaxPlateaus];
  __shared__ int sNPl, sNPeaks, sOverflow, sGround, sFirstStep, sNImg;

  const int frame = blockIdx.x, lane = threadIdx.x;
  if(frame >= nframes)
    return;
  FrameState &fs = st[frame];
  const int nb = P.nBins;

  unsigned int total = 0;
  for(int b = lane; b <= kMaxBins; b += 64)
  {
    /* K1's accumulator is taken over and left zero for the next call (FrameState::histAcc) */
    const unsigned int v = b < nb ? fs.histAcc[b] : 0u;
    hist[b] = v;
    total += v;
    if(b < kMaxBins)
    {
      lut[b] = 0xff;
      fs.hist[b] = v;
      if(b < nb)
        fs.histAcc[b] = 0u;
    }
  }
#pragma unroll
  for(int o = 32; o > 0; o >>= 1)
    total += __shfl_xor(total, o);
  __syncthreads();

  /* findPeaks (pointcloud.cpp:214-241) + filterPeaks (:243-256), all lanes: lane l looks at bins l and l + 64.  The
   * reference walks the bins with one carried flag: a bin is a peak when the count falls after it and the last CHANGE
   * before it was a rise (so a flat top reports its last index, quirk Q1).  Here: rise / fall bits of all bins by two
   * ballots each, then "the highest changed bin below me is a rise" by bit scans. */
  unsigned long long peakBits[2];
  {
    unsigned long long up[2], down[2];
#pragma unroll
    for(int w = 0; w < 2; w++)
    {
      const int i = lane + 64 * w;
      const bool inRange = i < nb - 1;
      const unsigned int c = hist[min(i, kMaxBins - 1)], succ = hist[min(i + 1, kMaxBins)];
      up[w] = __ballot(inRange && c < succ);
      down[w] = __ballot(inRange && c > succ);
    }
    const unsigned long long changed[2] = { up[0] | down[0], up[1] | down[1] };
#pragma unroll
    for(int w = 0; w < 2; w++)
    {
      const int i = lane + 64 * w;
      bool peak = false;
      if((down[w] >> lane) & 1ull)
      {
        const unsigned long long below = changed[w] & ((1ull << lane) - 1ull);
        bool lastWasRise = false;
        if(below)
          lastWasRise = (up[w] >> (63 - __clzll(static_cast<long long>(below)))) & 1ull;
        else if(w == 1 && changed[0])
          lastWasRise = (up[0] >> (63 - __clzll(static_cast<long long>(changed[0])))) & 1ull;
        if(lastWasRise)
        {
          const unsigned int np = hist[i];
          peak = np >= 2000u && (np * 2u - hist[i - 1] - hist[i + 1]) * 2u > np;     /* filterPeaks; i >= 1 after a rise */
        }
      }
      peakBits[w] = __ballot(peak);
    }
  }

  if(lane == 0)
  {
    int nPl = 0;
    int consumedUpTo = -1;           /* every bin <= this has already left pointsHt */
    bool overflow = false;
    int nPeaksDbg = 0;
    for(int w = 0; w < 2; w++)
      for(unsigned long long bits = peakBits[w]; bits; bits &= bits - 1ull)
      {
        const int i = 64 * w + __ffsll(static_cast<long long>(bits)) - 1;
        const unsigned int succ = hist[i + 1];
        if(nPeaksDbg < kMaxPlateaus)
          dbgPeaks[nPeaksDbg] = i;
        nPeaksDbg++;
        if(nPl >= kMaxPlateaus)
          overflow = true;
        else
        {
          /* extractPlateauPoints (:300-335): choose the pair, then take what is left of it */
          int hMin, hMax;
          if(hist[i - 1] > succ) { hMin = i - 1; hMax = i; }
          else { hMin = i; hMax = i + 1; }
          int lo, hi;
          if(hMin == 0)
          {
            /* quirk Q4: Height_t(heightMin - 1) wraps to 65535: everything goes to the remainder */
            lo = 1; hi = 0;
            consumedUpTo = nb;
          }
          else
          {
            lo = max(hMin, consumedUpTo + 1);
            hi = hMax;
            consumedUpTo = max(consumedUpTo, hMax);
          }
          unsigned int cnt = 0;
          for(int b = lo; b <= hi; b++)
          {
            cnt += hist[b];
            lut[b] = static_cast<unsigned char>(nPl);
          }
          plPeak[nPl] = i; plLo[nPl] = hMin; plHi[nPl] = hMax; plEffLo[nPl] = lo; plEffHi[nPl] = hi;
          plN[nPl] = static_cast<int>(cnt);
          nPl++;
        }
      }

    /* ground = most populous plateau below minHeight (pointcloud.cpp:402-418) */
    int groundInd = -1, i = 0;
    unsigned int maxGround = 0;
    for( ; i < nPl; i++)
    {
      if(plPeak[i] >= P.minHeight)
        break;
      if(maxGround < static_cast<unsigned int>(plN[i]))
      {
        maxGround = static_cast<unsigned int>(plN[i]);
        groundInd = i;
      }
    }
    int nImg = nPl - i;
    if(nImg > P.maxStepImages)
    {
      nImg = P.maxStepImages;
      overflow = true;
    }
    sNPl = nPl; sNPeaks = nPeaksDbg; sOverflow = overflow ? 1 : 0; sGround = groundInd; sFirstStep = i; sNImg = nImg;

    /* single pass: did k_predict give every bin of a step plateau a plane?  (A bin without points needs none.)  Plateau by
     * plateau: the covered ones are read from their planes, k_raster does the others of the frame (typically a weak last
     * peak at the top of the range that the sample did not show). */
    unsigned int covered = 0u;
    for(int slot = 0; spec && slot < nImg; slot++)
    {
      bool ok = spec != 0;
      unsigned char a = 0xff, b2 = 0xff;
      const int lo = plEffLo[i + slot], hi = plEffHi[i + slot];
      for(int b = lo; b <= hi; b++)                                        /* two bins at most */
      {
        const unsigned char p = spec ? fs.specPlane[b] : static_cast<unsigned char>(0xff);
        if(p == 0xff)
          ok = ok && hist[b] == 0u;
        else if(a == 0xff)
          a = p;
        else if(p != a)
          b2 = p;
      }
      /* a plane may hold two bins (k_predict): it must not bring points of a bin outside the plateau */
      if(spec && lo <= hi)
      {
        if(lo > 0 && fs.specPlane[lo - 1] != 0xff && (fs.specPlane[lo - 1] == a || fs.specPlane[lo - 1] == b2))
          ok = ok && hist[lo - 1] == 0u;
        if(hi + 1 < nb && fs.specPlane[hi + 1] != 0xff && (fs.specPlane[hi + 1] == a || fs.specPlane[hi + 1] == b2))
          ok = ok && hist[hi + 1] == 0u;
      }
      sImgPlane[slot][0] = a; sImgPlane[slot][1] = b2;
      covered |= ok ? 1u << slot : 0u;
    }
    sCovered = covered;
  }
  __syncthreads();

  const int nPl = sNPl, firstStep = sFirstStep, nImg = sNImg;
  unsigned int wanted = 0u;
  for(int b = lane; b < kMaxBins; b += 64)
  {
    const unsigned char l = lut[b];
    fs.lut[b] = l;
    const int slot = static_cast<int>(l) - firstStep;
    if(b < nb && l != 0xff && slot >= 0 && slot < nImg && !((sCovered >> slot) & 1u))      /* k_raster's gate: the plateaus it has to do */
      wanted |= 1u << (b / kBinsPerGroup);
  }
#pragma unroll
  for(int o = 32; o > 0; o >>= 1)
    wanted |= __shfl_xor(wanted, o);
  if(lane < nPl)
  {
    PlateauState &pl = fs.pl[lane];
    pl.peakBin = plPeak[lane];
    pl.binLo = plLo[lane];
    pl.binHi = plHi[lane];
    pl.effLo = plEffLo[lane];
    pl.effHi = plEffHi[lane];
    pl.nPoints = plN[lane];
    pl.isStep = plPeak[lane] >= P.minHeight ? 1 : 0;
    pl.outlineFound = 0;
    pl.valid = 0;
    for(int k = 0; k < 8; k++) { pl.quadImg[k] = 0.0; pl.quadWorld[k] = 0.0; }
  }
  /* The step images' boxes and z sums: empty / zero for k_raster to fill, or - single pass, every plateau covered - what
   * k_hist found in the planes of the plateau's bins (k_outline merges those planes into the image). */
  const unsigned int covered = sCovered;
  const bool specOk = nImg > 0 && covered == (nImg >= 32 ? 0xffffffffu : (1u << nImg) - 1u);
  unsigned int oobSum = 0u;
  if(lane <= kMaxStepImages)                          /* [kMaxStepImages] = the ground image */
  {
    int y0 = 0x7fffffff, y1 = -1, x0 = 0x7fffffff, x1 = -1;
    long long tz = 0;
    const bool mine = lane < nImg && ((covered >> lane) & 1u) != 0u;
    if(mine)
    {
#pragma unroll
      for(int k = 0; k < 2; k++)
      {
        const int p = sImgPlane[lane][k];
        if(p == 0xff)
          continue;
        if(fs.planeYMax[p] >= fs.planeYMin[p])
        {
          y0 = min(y0, fs.planeYMin[p]); y1 = max(y1, fs.planeYMax[p]);
          x0 = min(x0, fs.planeXMin[p]); x1 = max(x1, fs.planeXMax[p]);
        }
        tz += fs.planeTotZ[p];
        oobSum += fs.planeOob[p];
      }
      /* the planes hold sums of the bits of (z + 6144) over exactly the plateau's points (covered: every bin of it with
       * points has a plane, and no plane of it a point of another bin): minus the constant's bits per point, modulo 2^64,
       * this is the sum of round(z * 2^40) k_raster would have left (z_plus_magic_bits) */
      tz = static_cast<long long>(static_cast<unsigned long long>(tz) - static_cast<unsigned long long>(kMagicBits) * static_cast<unsigned long long>(plN[firstStep + lane]));
    }
    fs.imgYMin[lane] = y0; fs.imgYMax[lane] = y1;
    fs.imgXMin[lane] = x0; fs.imgXMax[lane] = x1;
    if(lane < kMaxStepImages)
    {
      fs.totZ[lane] = tz;                              /* k_raster's sums start from zero */
      fs.imgPlane[lane][0] = mine ? sImgPlane[lane][0] : static_cast<unsigned char>(0xff);
      fs.imgPlane[lane][1] = mine ? sImgPlane[lane][1] : static_cast<unsigned char>(0xff);
    }
  }
#pragma unroll
  for(int o = 32; o > 0; o >>= 1)
    oobSum += __shfl_xor(oobSum, o);
  if(spec && lane < kMaxPlanes)
  {
    bool used = false;
    for(int sl = 0; sl < nImg; sl++)
      used = used || (((covered >> sl) & 1u) != 0u && (sImgPlane[sl][0] == lane || sImgPlane[sl][1] == lane));
    fs.planeUsed[lane] = used ? 1 : 0;
  }
  if(lane == 0)
  {
    if(spec && !specOk && nImg > 0)
    {
      const int at = atomicAdd(&fallback[0], 1);
      if(SSD_CHK(25, at, nframes))
        fallback[kFallbackList + at] = frame;          /* k_raster's work list (k_predict zeroed the count) */
    }
    if(spec && frame == 0)
      fallback[2] = 0;                                                       /* the plane pool's counter: every k_predict block of this batch is long done */
    if(spec && nImg == 0)
      atomicAdd(&fallback[1], 1);                                            /* frames WITHOUT a step plateau (the rare kind where stairs are in sight: a thousand adds to one address cost k_peaks 10 us): what the single pass cannot gain on, ssd_fetch_back */
    fs.specOk = specOk ? 1 : 0;
    fs.slotCovered = covered;
    fs.nNonZero = fs.nNonZeroAcc;
    fs.nNonZeroAcc = 0u;
    fs.nOob = oobSum;
    fs.status = (sOverflow ? static_cast<unsigned int>(SSD_ST_OVERFLOW) : 0u) | (oobSum ? static_cast<unsigned int>(SSD_ST_OOB_PIXEL) : 0u);     /* the frame's status starts here */
    fs.nInRange = total;
    fs.nPlateaus = nPl;
    fs.groundInd = sGround;
    fs.firstStep = firstStep;
    fs.nStepImages = nImg;
    fs.firstValidInd = -1;
    fs.wantedSteps = wanted;
    fs.wantedQuads = 0u;
    fs.anyActive = 0u;
  }

  if(dbg)
  {
    ssd_debug_frame &d = dbg[frame].d;
    for(int b = lane; b < kMaxBins; b += 64)
      d.hist[b] = hist[b];
    if(lane < min(sNPeaks, kMaxPlateaus))
      d.peaks[lane] = dbgPeaks[lane];
    if(lane < nPl)
    {
      ssd_debug_plateau &p = d.plateaus[lane];
      p.peak_bin = plPeak[lane]; p.bin_lo = plLo[lane]; p.bin_hi = plHi[lane];
      p.eff_lo = plEffLo[lane]; p.eff_hi = plEffHi[lane];
      p.n_points = plN[lane];
      p.is_step = plPeak[lane] >= P.minHeight ? 1 : 0;
    }
    if(lane == 0)
    {
      d.n_nonzero = static_cast<int>(fs.nNonZero);
      d.n_inrange = static_cast<int>(total);
      d.n_bins = nb;
      d.min_height = P.minHeight;
      d.min_img_y_extent = P.minImgYExtent;
      d.n_peaks = sNPeaks;
      d.n_plateaus = nPl;
      d.first_step = firstStep;
      d.ground_ind = sGround;
    }
  }
}

/* ========================================================================= */
/* K2: raster the step plateaus into bit images                               */

struct RasterLds
{
  unsigned long long wins[kThreads / 64][kWinWords];
  unsigned int wmiss[kThreads / 64][kWaveMissWords];
  ImageBox boxes[kMaxStepImages];
  unsigned char lut[kMaxBins];
  unsigned int lOob;
  unsigned short cellList[kMaxCellsPerBlockRaster];
  unsigned int listScratch[2 * kWavesPerBlock];
  unsigned long long ltot[kMaxStepImages][8];       /* sum of round(z * 2^40) over ALL points of each step plateau (this block's share) */
};

template<int SRC>
__device__ __forceinline__ void raster_block(RasterLds &L, const float *__restrict__ xyz, size_t strideFloats, const PointParams &P,
                                             const PixelParams &X, FrameState *__restrict__ st,
                                             unsigned long long *__restrict__ stepImg,
                                             const uint2 *__restrict__ tileMasks, size_t tileMaskStride, int chunkPoints, const DepthSrc &D,
                                             const int frame, const int chunkIdx)
{
  unsigned long long (&wins)[kThreads / 64][kWinWords] = L.wins;
  unsigned int (&wmiss)[kThreads / 64][kWaveMissWords] = L.wmiss;
  ImageBox (&boxes)[kMaxStepImages] = L.boxes;
  unsigned char (&lut)[kMaxBins] = L.lut;
  unsigned int &lOob = L.lOob;
  unsigned short (&cellList)[kMaxCellsPerBlockRaster] = L.cellList;
  unsigned int (&listScratch)[2 * kWavesPerBlock] = L.listScratch;
  unsigned long long (&ltot)[kMaxStepImages][8] = L.ltot;

  const int tid = threadIdx.x, lane = tid & 63;
  FrameState &fs = st[frame];
  /* all the block needs of the frame's state in one memory round trip (the image count decides whether there is work) */
  const int nImg = fs.nStepImages;
  const int firstStep = fs.firstStep;
  const unsigned char lutMine = tid < kMaxBins ? fs.lut[tid] : static_cast<unsigned char>(0xff);
  const unsigned int wantedSteps = fs.wantedSteps;
  const unsigned int slotCovered = fs.slotCovered;        /* single pass: plateaus k_hist has rastered already (k_peaks; 0 otherwise) */
  if(nImg == 0 || wantedSteps == 0u)
    return;
  BlockPhase ph(0);
  if(tid < kMaxBins)
  {
    /* bin -> image slot of a step plateau, 0xff = no image for this bin */
    const int sl = static_cast<int>(lutMine) - firstStep;
    lut[tid] = (lutMine != 0xff && sl >= 0 && sl < nImg && !((slotCovered >> sl) & 1u)) ? static_cast<unsigned char>(sl) : static_cast<unsigned char>(0xff);
  }
  if(tid < kMaxStepImages)
    boxes[tid] = ImageBox{ 0x7fffffff, -1, 0x7fffffff, -1 };
  if(tid < kMaxStepImages * 8)
    (&ltot[0][0])[tid] = 0ull;
  if(tid == 0)
    lOob = 0;
  for(int i = tid; i < (kThreads / 64) * kWinWords; i += kThreads)
    (&wins[0][0])[i] = 0ull;
  if(tid < kThreads / 64)
    wavemiss_init(wmiss[tid]);
  __syncthreads();
  ph.mark(0);

  /* strideFloats counts floats, or 16-bit depth values for kSrcDepth16 */
  const float *base = SRC == kSrcDepth16
    ? reinterpret_cast<const float *>(reinterpret_cast<const unsigned short *>(xyz) + static_cast<size_t>(frame) * strideFloats)
    : xyz + static_cast<size_t>(frame) * strideFloats;
  const int begin = chunkIdx * chunkPoints;
  const int end = min(begin + chunkPoints, P.nPoints);
  const unsigned int imgWords = static_cast<unsigned int>(X.H) * X.W64;
  unsigned long long *frameImg = stepImg + static_cast<size_t>(frame) * X.maxStepImages * imgWords;
  unsigned long long *ww = wins[tid >> 6];
  unsigned int *wm = wmiss[tid >> 6];
  WaveWindow win;
  win.limitWords = static_cast<unsigned int>(X.maxStepImages) * imgWords;
  unsigned int oob = 0;
  /* The z sum of EVERY point of each step plateau, in and out of its outline (the count is the histogram's): k_inquad
   * then only has to visit the cells that can hold points OUTSIDE the quadrilateral and take those off again.  A lane
   * walks down a camera column, so its hits nearly always stay on one plateau: running sum in registers. */
  int curT = -1;
  unsigned long long accT = 0;
  auto flushT = [&]()
  {
    if(curT >= 0)
      atomicAdd(&ltot[curT][lane & 7], accT);
  };

  /* Only the cells that hold a bin of a step plateau are walked (cell_list_build), four per wave iteration, with the
   * loads of the NEXT group issued before the current one is processed: a wave that waits for its own loads each
   * iteration leaves the SIMD to seven others, which no longer cover the memory latency once the body is this short. */
  const int cell0 = begin / kCell;
  const int nCells = (end - begin + kCell - 1) / kCell;
  const int count = cell_list_build(tileMasks + static_cast<size_t>(frame) * tileMaskStride + cell0, nCells, X.cellCols,
                                    [&](const uint2 info) { return (info.x & wantedSteps) != 0u; }, cellList, listScratch);
  ph.mark(1);
  const int nGroups = (count + 3) >> 2;
  int g = (tid >> 6) * nGroups / kWavesPerBlock;
  const int gEnd = ((tid >> 6) + 1) * nGroups / kWavesPerBlock;
  if(g < gEnd)
  {
    F3 v[kPts], vn[kPts];
    load_cell<SRC>(base, cell0, cellList, 4 * g + (lane >> 4), count, lane, P.nPoints, v, D);
    while(true)
    {
    const bool more = g + 1 < gEnd;
    if(more)
      load_cell<SRC>(base, cell0, cellList, 4 * (g + 1) + (lane >> 4), count, lane, P.nPoints, vn, D);
    unsigned int key[kPts];
    #pragma unroll
    for(int j = 0; j < kPts; j++)
    {
      /* few, flat decisions per point (see k_inquad) */
      key[j] = kNoPixel;
      double wx, wy, wz;
      const bool ok = world_point_flat(P, v[j], wx, wy, wz);
      const int bin = ok ? height_bin(P, wz) : 0;
      const int slot = SSD_CHK(30, bin, kMaxBins) ? lut[bin] : 0xff;
      if(!(ok & (slot != 0xff)))
        continue;
      int ix, iy;
      const bool inside = image_pixel(P, X, wx, wy, ix, iy);
      if(slot != curT)
      {
        flushT();
        curT = SSD_CHK(31, slot, kMaxStepImages) ? slot : -1;
        accT = 0;
      }
      accT += static_cast<unsigned long long>(z_to_fixed(wz));
      oob += inside ? 0u : 1u;                                /* quirk Q5 */
      key[j] = inside ? pixel_key(slot, iy, ix) : kNoPixel;
    }
    wavewin_emit(ww, wm, win, frameImg, imgWords, X.W64, X.winShift, boxes, key, lane);
    if(!more)
      break;
    g++;
#pragma unroll
    for(int j = 0; j < kPts; j++)
      v[j] = vn[j];
    }
  }
  ph.mark(2);
  wavewin_flush(ww, win, frameImg, imgWords, X.W64, X.winShift, boxes, lane);
  wavemiss_flush(wm, boxes, lane);
  flushT();
  if(oob)
    atomicAdd(&lOob, oob);
  ph.mark(5);
  __syncthreads();
  ph.mark(3);
  if(tid < nImg)
  {
    unsigned long long t = 0;
#pragma unroll
    for(int k = 0; k < 8; k++)
      t += ltot[tid][k];
    if(t)
      atomicAdd(reinterpret_cast<unsigned long long *>(&fs.totZ[tid]), t);
  }
  if(tid < nImg && boxes[tid].yMax >= 0)
  {
    atomicMin(&fs.imgYMin[tid], boxes[tid].yMin); atomicMax(&fs.imgYMax[tid], boxes[tid].yMax);
    atomicMin(&fs.imgXMin[tid], boxes[tid].xMin); atomicMax(&fs.imgXMax[tid], boxes[tid].xMax);
  }
  if(tid == 0 && lOob)
  {
    atomicAdd(&fs.nOob, lOob);
    atomicOr(&fs.status, static_cast<unsigned int>(SSD_ST_OOB_PIXEL));
  }
  ph.mark(4);
  ph.finish();
}

/* LIST (single pass): the grid's x blocks share the frames k_peaks listed in `fallback` (count, then frame indices) - the
 * frames whose step plateaus the planes did not cover; none, as a rule, and then every block leaves after one cached load.
 * A template flag, not a run-time test: with both forms in one kernel the two-pass instantiation - the one every depth-16,
 * small, backed-off or partial call takes - paid for the list's loop with 13 - 17 vector registers in scratch (round 4).  The
 * list's own instantiation runs a handful of frames per batch: built for five waves per SIMD, which leaves it the registers
 * its loop wants (no scratch either). */
template<int SRC, bool LIST>
__global__ __launch_bounds__(kThreads, LIST ? 5 : SSD_K2_WAVES) void k_raster(const float *__restrict__ xyz, size_t strideFloats, PointParams P,
                                                        PixelParams X, FrameState *__restrict__ st,
                                                        unsigned long long *__restrict__ stepImg,
                                                        const uint2 *__restrict__ tileMasks, size_t tileMaskStride, int chunkPoints, DepthSrc D,
                                                        const int *__restrict__ fallback)
{
  __shared__ RasterLds L;
  if constexpr(LIST)
  {
    const int n = fallback[0];
    for(int e = blockIdx.x; e < n; e += gridDim.x)
    {
      raster_block<SRC>(L, xyz, strideFloats, P, X, st, stepImg, tileMasks, tileMaskStride, chunkPoints, D, fallback[kFallbackList + e], blockIdx.y);
      __syncthreads();
    }
  }
  else
    raster_block<SRC>(L, xyz, strideFloats, P, X, st, stepImg, tileMasks, tileMaskStride, chunkPoints, D, blockIdx.x, blockIdx.y);
}

/* ========================================================================= */
/* BestLine (segmentation.cpp:409-487): a pair of points per lane, the waves dealt out over the lists */

/* (the residual of a pair's line, three forms: ssd_bestline.h — host + device code, the CPU suite runs its host build) */

/* line_residual_onepass for lists of at most 64 points held one per lane (myX, myY of lane i = point i): the walk over the
 * points reads them with v_readlane (a few cycles) instead of LDS (a round trip per point in a lone wave) */
__device__ __forceinline__ double line_residual_lanes(int myX, int myY, int m, int p, int q, LineI &line)
{
  line = line_through_i(__shfl(myX, p), __shfl(myY, p), __shfl(myX, q), __shfl(myY, q));
  return line_residual_onepass([&](int i, int &x, int &y) { x = __builtin_amdgcn_readlane(myX, i); y = __builtin_amdgcn_readlane(myY, i); }, m, line);
}

/* all 64 lanes of the calling wave take part, on the pairs tStart + lane, tStart + tStride + lane, ..: the best of them
 * (residual, pair index, line; pair index 0x7fffffff = none) is returned in every lane */
__device__ void wave_best_line_part(const int *px, const int *py, int m, int lane, bool smallImage, int tStart, int tStride,
                                    double &bestRes, int &bestT, LineI &bestLine)
{
  const int nPairs = m * (m - 1) / 2;
  bestRes = 1.0e300;
  bestT = 0x7fffffff;
  bestLine = LineI{ 0, 0, 0 };
  const bool inLanes = smallImage && m <= 64;
  const int myX = inLanes && lane < m ? px[lane] : 0, myY = inLanes && lane < m ? py[lane] : 0;
  /* every lane runs the same number of rounds (the lanes' points are exchanged by shuffles); a lane without a pair
   * works on pair 0 and discards the result */
  for(int t0 = tStart; t0 < nPairs; t0 += tStride)
  {
    const int t = t0 + lane;
    const bool real = t < nPairs;
    /* decode pair t in the order of the reference's double loop (p ascending, q > p ascending) */
    int p = 0, rem = real ? t : 0;
    while(rem >= m - 1 - p)
    {
      rem -= m - 1 - p;
      p++;
    }
    const int q = p + 1 + rem;
    LineI l;
    double r;
    if(inLanes)
      r = line_residual_lanes(myX, myY, m, p, q, l);
    else
      r = smallImage && m <= 128 ? line_residual_keys(px, py, m, p, q, l) : line_residual_generic(px, py, m, p, q, l);
    if(real && (bestT == 0x7fffffff || r < bestRes))     /* min_element: first of equal minima (t ascends per lane) */
    {
      bestRes = r;
      bestT = t;
      bestLine = l;
    }
  }
  /* butterfly: smallest residual, ties to the smaller pair index */
#pragma unroll
  for(int o = 32; o > 0; o >>= 1)
  {
    const double oRes = __shfl_xor(bestRes, o);
    const int oT = __shfl_xor(bestT, o);
    const int oa = __shfl_xor(bestLine.a, o), ob = __shfl_xor(bestLine.b, o), oc = __shfl_xor(bestLine.c, o);
    const bool mineValid = bestT != 0x7fffffff, otherValid = oT != 0x7fffffff;
    const bool take = otherValid && (!mineValid || oRes < bestRes || (oRes == bestRes && oT < bestT));
    if(take)
    {
      bestRes = oRes; bestT = oT; bestLine.a = oa; bestLine.b = ob; bestLine.c = oc;
    }
  }
}

/* one wave, all pairs */
__device__ LineI wave_best_line(const int *px, const int *py, int m, int lane, bool smallImage)
{
  double res;
  int t;
  LineI l;
  wave_best_line_part(px, py, m, lane, smallImage, 0, 64, res, t, l);
  return l;
}

/* ========================================================================= */
/* K3: outline of one step plateau image — one workgroup per (slot, frame)      */

/* threads of the image kernels (K3, K5): one block per image, a chain of short phases.  More waves = fewer rows per
 * thread in the closing (latency), but the arithmetic phases (BestLine) run at the CU's rate whatever the count, and every
 * barrier and every per-wave preamble is paid per wave: measured on 1024 XGA frames, K3 takes 0.106 / 0.146 / 0.307 ms with
 * 256 / 512 / 1024 threads, on a single frame 30 / 26.5 / 28 us.  Hence two instantiations: 256 for batches, 512 for a few
 * frames. */
/* (round 4, FHD stress batch of 256 frames, same box, alternating: k_outline with 256 / 512 / 1024 threads per image 0.310 / 0.363 /
 * 0.438 ms — the batch is bound by the images' total work, which more threads do not shrink; tools/mkvariant.sh with
 * -DSSD_IMG_THREADS_BATCH=..) */
#ifndef SSD_IMG_THREADS_BATCH
#define SSD_IMG_THREADS_BATCH 256
#endif
constexpr int kImgThreadsBatch = SSD_IMG_THREADS_BATCH, kImgThreadsFew = 512;   /* chosen per launch (launch_outline / launch_final) */
constexpr int kMaxImgWaves = (kImgThreadsBatch > kImgThreadsFew ? kImgThreadsBatch : kImgThreadsFew) / 64;
constexpr int kImgFewFrames = 64;
constexpr int kMaxCols = SSD_MAX_SCANS;      /* scan columns per image (W/25 + 1 <= 128) */
constexpr int kMaxProbe = SSD_MAX_EDGE_PTS;  /* probe rows per vertical edge (H/10 + 1 <= 256) */

struct OutlineShared
{
  int yFirst[kMaxCols], ySecond[kMaxCols];
  int ex[4][kMaxCols], ey[4][kMaxCols];      /* the four point lists: FL, FR, BL, BR */
  int en[4];
  LineI line[4];
  int nRight, nLeft;
  int found;
  /* vertical edges */
  int vActive[2], vLeft[2], vRight[2], vYStart[2], vYEnd[2], vProbe[2];
  int vx[2][kMaxProbe];
  int vpx[2][kMaxProbe], vpy[2][kMaxProbe], vn[2];
  double vdist[2][kMaxProbe];
  int vBest[2];
  int sortStack[2][3][kSortStack];      /* gnu_sort_on's explicit recursion stack, one per side (lane 0 of the side's wave) */
  LineD baseLine;
  LineD nline[4];                            /* the edges' lines normalised, left edges reversed (calcBaseLine) */
  double partRes[kMaxImgWaves];                 /* BestLine: every wave's best over its share of its edge's pairs */
  int partT[kMaxImgWaves];
  LineI partLine[kMaxImgWaves];
  double bounds[4][2][2];
  unsigned int status;
};

template<int T>
__global__ __launch_bounds__(T) void k_outline(Params P, FrameState *__restrict__ st,
                                                      unsigned long long *__restrict__ stepImg,
                                                      unsigned long long *__restrict__ planeImg,
                                                      DebugFrame *__restrict__ dbg,
                                                      unsigned long long *__restrict__ dbgImg)
{
  __shared__ OutlineShared S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slot = blockIdx.y, frame = blockIdx.x;
  FrameState &fs = st[frame];
  const size_t imgWords = static_cast<size_t>(P.H) * P.W64;
  if(planeImg)
  {
    /* single pass: the planes no step image is made of (bins beside a plateau; every plane of a frame that k_raster had to
     * do) are cleared where k_hist set bits, and their boxes emptied - by the frame's blocks that have no image to work on */
    const int nBusy = min(fs.nStepImages, static_cast<int>(gridDim.y));
    const bool everyone = nBusy >= static_cast<int>(gridDim.y);              /* no block without an image: all share the duty */
    const int pFirst = everyone ? slot : slot - nBusy, pStep = everyone ? static_cast<int>(gridDim.y) : static_cast<int>(gridDim.y) - nBusy;
    for(int p = pFirst < 0 ? fs.nPlanes : pFirst; p < fs.nPlanes; p += pStep)
    {
      if(fs.planeUsed[p])
        continue;
      const int y0 = fs.planeYMin[p], y1 = fs.planeYMax[p], c0 = fs.planeXMin[p], c1 = fs.planeXMax[p];
      if(y1 < y0)
        continue;
      unsigned long long *pi = planeImg + (static_cast<size_t>(fs.planeBase) + p) * imgWords;
      const int bw = c1 - c0 + 1, n = bw * (y1 - y0 + 1);
      for(int t = tid; t < n; t += T)
      {
        const int y = y0 + t / bw, c = c0 + t % bw;
        if(SSD_CHK(20, static_cast<size_t>(y) * P.W64 + c, imgWords) && SSD_CHK(26, c, P.W64) && SSD_CHK(27, p, kMaxPlanes))
          pi[static_cast<size_t>(y) * P.W64 + c] = 0ull;
      }
      __syncthreads();                               /* everybody has read the box */
      if(tid == 0)
      {
        fs.planeYMin[p] = 0x7fffffff; fs.planeYMax[p] = -1;
        fs.planeXMin[p] = 0x7fffffff; fs.planeXMax[p] = -1;
      }
    }
  }
  if(slot >= fs.nStepImages)
    return;
  const int plIdx = fs.firstStep + slot;
  PlateauState &pl = fs.pl[plIdx];
  ssd_debug_plateau *dp = dbg ? &dbg[frame].d.plateaus[plIdx] : nullptr;

  /* The raw image: the step image k_raster filled, or - single pass, the frame's plateaus covered - the planes of the
   * plateau's (two) bins, read together (BitImg::w2).  Its box (k_peaks) is the union of the planes' boxes. */
  unsigned long long *img = stepImg + (static_cast<size_t>(frame) * P.maxStepImages + slot) * imgWords;
  unsigned long long *img2 = nullptr;
  const bool fromPlanes = planeImg && ((fs.slotCovered >> slot) & 1u) != 0u;
  if(fromPlanes)
  {
    const int pa = fs.imgPlane[slot][0], pb = fs.imgPlane[slot][1];
    if(pa != 0xff)
      img = planeImg + (static_cast<size_t>(fs.planeBase) + pa) * imgWords;
    if(pb != 0xff)
      img2 = planeImg + (static_cast<size_t>(fs.planeBase) + pb) * imgWords;
  }
  const BitImg im{ img, P.W, P.H, P.W64, img2 };
  SSD_PHASE(0, 0);

  /* scan columns: x_j = xr0 + 25 j; the centre column xc = W/2 is j = jc */
  const int xStep = 25;
  const int xc = P.W / 2;
  const int xr0 = xc % xStep;
  const int jc = xc / xStep;
  const int nCols = min(kMaxCols, (P.W - 1 - xr0) / xStep + 1);

  for(int j = tid; j < kMaxCols; j += T)
  {
    S.yFirst[j] = 0x7fffffff;
    S.ySecond[j] = -1;
  }
  if(tid == 0)
  {
    S.found = 0;
    S.status = 0;
    S.nRight = S.nLeft = 0;
    S.vActive[0] = S.vActive[1] = 0;
    S.vn[0] = S.vn[1] = 0;
    S.vProbe[0] = S.vProbe[1] = 0;
  }
  __syncthreads();
  SSD_PHASE(0, 1);

  /* ---- phase B: closed image, column extents (Scanner::probeVertical, segmentation.cpp:88-111) ---- */
  unsigned long long *dbgRaw = nullptr, *dbgClosed = nullptr;
  if(dbgImg)
  {
    dbgRaw = dbgImg + ((static_cast<size_t>(frame) * (P.maxStepImages + 1) + slot) * 2) * imgWords;
    dbgClosed = dbgRaw + imgWords;
  }
  /* Only the bounding box of the raw bits is visited (with debug capture on, the whole image, so that the
   * captured images are complete).  The closing of a set stays inside the set's bounding box — except along
   * the image border: the erosion ignores out-of-image pixels (segmentation.cpp:888,928, default border), so a
   * lit pixel one row / column away from the border closes the border pixel next to it as well. */
  int by0 = fs.imgYMin[slot], by1 = fs.imgYMax[slot], bc0 = fs.imgXMin[slot], bc1 = fs.imgXMax[slot];
  const bool emptyImg = by1 < by0;
  grow_box_to_border(P.W, P.H, P.W64, by0, by1, bc1);
  if(dbgImg)
  {
    by0 = 0; by1 = P.H - 1; bc0 = 0; bc1 = P.W64 - 1;
  }
  const int bw = emptyImg && !dbgImg ? 0 : bc1 - bc0 + 1, bh = emptyImg && !dbgImg ? 0 : by1 - by0 + 1;
  if(dbgImg)
  {
    /* debug capture: the raw and the closed image whole, word by word (thread -> word column, band of rows) */
    const int nBands = bw > 0 ? max(1, T / bw) : 0;
    const int bandRows = nBands > 0 ? (bh + nBands - 1) / nBands : 0;
    for(int t = tid; t < bw * nBands; t += T)
    {
      const int band = t / bw;
      const int c = bc0 + (t - band * bw);
      const int yA = by0 + band * bandRows, yB = min(yA + bandRows, by1 + 1);
      closed_column(im, c, yA, yB, true, [&](int y, unsigned long long cw)
      {
        dbgRaw[y * P.W64 + c] = raw_word(im, y, c);
        dbgClosed[y * P.W64 + c] = cw;
      });
    }
  }
  /* the scan columns inside the box, each cut into bands of rows: thread -> (column, band); first and last closed row of
   * the band in registers, one LDS atomic pair per thread */
  if(bw > 0)
  {
    const int xLo = 64 * bc0, xHi = min(64 * bc1 + 63, P.W - 1);
    const int jLo = xLo <= xr0 ? 0 : (xLo - xr0 + xStep - 1) / xStep;
    const int jHi = xHi < xr0 ? -1 : min(nCols - 1, (xHi - xr0) / xStep);
    const int nJ = jHi - jLo + 1;
    const int nBands = nJ > 0 ? max(1, T / nJ) : 0;
    const int bandRows = nBands > 0 ? (bh + nBands - 1) / nBands : 0;
    for(int t = tid; t < nJ * nBands; t += T)
    {
      const int band = t / nJ;
      const int j = jLo + (t - band * nJ);
      const int yA = by0 + band * bandRows, yB = min(yA + bandRows, by1 + 1);
      int yLo = 0x7fffffff, yHi = -1;
      closed_scan_column(im, xr0 + xStep * j, yA, yB, [&](int y)
      {
        yLo = min(yLo, y);
        yHi = y;
      });
      if(yHi >= 0)
      {
        atomicMin(&S.yFirst[j], yLo);
        atomicMax(&S.ySecond[j], yHi);
      }
    }
  }
  __syncthreads();
  SSD_PHASE(0, 2);

  /* ---- phase C: scans (Scanner::scan :59-85, HorizontalEdgesDetector::detect :607-620),
   *      point lists (Scanner::obtainLinePoints :129-156) ---- */
  {
    /* every wave counts for itself (ballots over the columns' "long enough" bits, no serial walk through LDS) */
    bool ok0 = false, ok1 = false;
    if(lane < nCols)
      ok0 = S.ySecond[lane] >= 0 && S.ySecond[lane] - S.yFirst[lane] >= P.minImgYExtent;
    if(lane + 64 < nCols)
      ok1 = S.ySecond[lane + 64] >= 0 && S.ySecond[lane + 64] - S.yFirst[lane + 64] >= P.minImgYExtent;
    const unsigned long long okLo = __ballot(ok0), okHi = __ballot(ok1);
    auto okAt = [&](int j) { return j >= 0 && j < nCols && (((j < 64 ? okLo : okHi) >> (j & 63)) & 1ull) != 0ull; };
    int nR = 0, nL = 0;
    while(okAt(jc + nR))
      nR++;
    if(nR > 0)
      while(okAt(jc - 1 - nL))
        nL++;
    const bool found = nR > 0 && nR + nL >= 3;
    /* The four point lists (Scanner::obtainLinePoints :129-156).  Its pushes come out as two runs of consecutive columns:
     * the right lists take the columns cR0, cR0 + 1, .. (nRl of them), the left lists cL0, cL0 - 1, .. (nLl), with
     * cR0 = cL0 = the column where the reference starts: as many columns of the longer side as put total/2 + 1 on it. */
    const int total = nR + nL;
    const int half = total / 2 + 1;
    int c0, nRl, nLl;
    if(nL >= half)
    {
      const int a = nL - half;
      c0 = jc - 1 - a;
      nRl = a + 1 + nR;
      nLl = half;
    }
    else
    {
      const int b = nR > half ? nR - half : 0;
      c0 = jc + b;
      nRl = nR - b;
      nLl = b + 1 + nL;
    }
    if(found)
    {
      for(int k = tid; k < nRl; k += T)
      {
        const int j = c0 + k, x = xr0 + xStep * j;
        S.ex[kFR][k] = x; S.ey[kFR][k] = S.ySecond[j];
        S.ex[kBR][k] = x; S.ey[kBR][k] = S.yFirst[j];
      }
      for(int k = tid; k < nLl; k += T)
      {
        const int j = c0 - k, x = xr0 + xStep * j;
        S.ex[kFL][k] = x; S.ey[kFL][k] = S.ySecond[j];
        S.ex[kBL][k] = x; S.ey[kBL][k] = S.yFirst[j];
      }
    }
    if(tid == 0)
    {
      S.nRight = nR;
      S.nLeft = nL;
      if(found)
      {
        S.found = 1;
        S.en[kFR] = S.en[kBR] = nRl;
        S.en[kFL] = S.en[kBL] = nLl;
      }
    }
  }
  __syncthreads();

  SSD_PHASE(0, 3);
  if(dp && tid == 0)
  {
    dp->n_scans_right = S.nRight;
    dp->n_scans_left = S.nLeft;
    for(int s = 0; s < S.nRight; s++)
    { dp->scans_right[s][0] = xr0 + xStep * (jc + s); dp->scans_right[s][1] = S.yFirst[jc + s]; dp->scans_right[s][2] = S.ySecond[jc + s]; }
    for(int s = 0; s < S.nLeft; s++)
    { dp->scans_left[s][0] = xr0 + xStep * (jc - 1 - s); dp->scans_left[s][1] = S.yFirst[jc - 1 - s]; dp->scans_left[s][2] = S.ySecond[jc - 1 - s]; }
  }

  if(S.found)
  {
    /* ---- BestLine per horizontal edge (HorizontalEdges::Edge :570-583): wave w works on edge w % 4; then the edge's
     *      BoundaryPoints (:521-552): the first and the last list point within 10 pixels of the line, by ballots ---- */
    static_assert((T / 64) % 4 == 0, "the waves are dealt out over the four edges");
    {
      /* wave w works on edge w % 4, on every ((T / 64) / 4)-th round of 64 pairs */
      constexpr int kParts = (T / 64) / 4;
      double res;
      int t;
      LineI l;
      wave_best_line_part(S.ex[wave & 3], S.ey[wave & 3], S.en[wave & 3], lane, 3ll * P.W * P.H < (1ll << 25), 64 * (wave >> 2), 64 * kParts, res, t, l);
      if(lane == 0)
      {
        S.partRes[wave] = res;
        S.partT[wave] = t;
        S.partLine[wave] = l;
      }
    }
    __syncthreads();
    if(wave < 4)
    {
      const int e = wave, n = S.en[e];
      /* the parts' bests combined: smallest residual, ties to the smaller pair index (min_element's first of equals) */
      int best = -1;
      for(int w = e; w < (T / 64); w += 4)
        if(S.partT[w] != 0x7fffffff && (best < 0 || S.partRes[w] < S.partRes[best] || (S.partRes[w] == S.partRes[best] && S.partT[w] < S.partT[best])))
          best = w;
      const LineI l = best >= 0 ? S.partLine[best] : LineI{ 0, 0, 0 };
      const double fm = static_cast<double>(-l.a) / l.b;       /* FlatLine :508-511 */
      const double fn = static_cast<double>(-l.c) / l.b;
      int first = -1, last = -1;
      for(int i0 = 0; i0 < n; i0 += 64)
      {
        const int i = i0 + lane;
        const bool near = i < n && fabs(S.ex[e][i < n ? i : 0] * fm + fn - S.ey[e][i < n ? i : 0]) < 10;
        const unsigned long long m = __ballot(near);
        if(m != 0ull)
        {
          if(first < 0)
            first = i0 + __ffsll(static_cast<long long>(m)) - 1;
          last = i0 + 63 - __clzll(static_cast<long long>(m));
        }
      }
      if(lane == 0)
      {
        S.line[e] = l;
        double in[2] = { -1.0, -1.0 }, out[2] = { -1.0, -1.0 };
        if(first >= 0)
        {
          in[0] = S.ex[e][first]; in[1] = S.ex[e][first] * fm + fn;
          out[0] = S.ex[e][last]; out[1] = S.ex[e][last] * fm + fn;
        }
        if((in[0] == -1.0 && in[1] == -1.0) || (out[0] == -1.0 && out[1] == -1.0))
          atomicOr(&S.status, static_cast<unsigned int>(SSD_ST_ASSERT));
        S.bounds[e][0][0] = in[0]; S.bounds[e][0][1] = in[1];
        S.bounds[e][1][0] = out[0]; S.bounds[e][1][1] = out[1];
        /* calcBaseLine's first step, one edge per wave: left edges reversed */
        const bool leftEdge = e == kFL || e == kBL;
        S.nline[e] = leftEdge ? normalized_line(-l.a, -l.b, -l.c) : normalized_line(l.a, l.b, l.c);
      }
    }
    __syncthreads();
    SSD_PHASE(0, 4);

    /* ---- base line (:672-679), detectEdge windows (:681-697) ---- */
    if(tid == 0)
    {
      /* calcBaseLine: bisectors of (reversed left, right) front and back lines, then of those two;
       * slope-corrected by xyRatio^2, perpendicular through the first front-left point */
      const LineD nfl = S.nline[kFL], nfr = S.nline[kFR], nbl = S.nline[kBL], nbr = S.nline[kBR];
      const LineD front{ nfl.a + nfr.a, nfl.b + nfr.b, nfl.c + nfr.c };
      const LineD back{ nbl.a + nbr.a, nbl.b + nbr.b, nbl.c + nbr.c };
      const LineD nf = normalized_line(front.a, front.b, front.c), nb = normalized_line(back.a, back.b, back.c);
      const LineD center{ nf.a + nb.a, nf.b + nb.b, nf.c + nb.c };
      const double corr = P.xyRatio * P.xyRatio;
      const double sa = center.a * corr, sb = center.b;
      const int p0x = S.ex[kFL][0], p0y = S.ey[kFL][0];
      S.baseLine = LineD{ -sb, sa, sb * p0x - sa * p0y };

      for(int side = 0; side < 2; side++)
      {
        const int fe = side == 0 ? kFL : kFR, be = side == 0 ? kBL : kBR;
        const double fx = S.bounds[fe][1][0], fy = S.bounds[fe][1][1];
        const double bx = S.bounds[be][1][0], by = S.bounds[be][1][1];
        int lft = static_cast<int>(fmin(fx, bx) - xStep);
        int rgt = static_cast<int>(fmax(fx, bx) + xStep);
        int yStart = static_cast<int>(fy - 10);
        int yEnd = static_cast<int>(by + 10);
        if(lft < 0) lft = 0;
        if(rgt >= P.W) rgt = P.W - 1;
        if(yStart >= P.H) yStart = P.H - 1;
        if(yEnd < 0) yEnd = 0;
        if(yStart < yEnd)
          continue;
        if(rgt - lft <= 0)
        {
          S.status |= SSD_ST_ASSERT;
          continue;
        }
        int np = (yStart - yEnd) / 10 + 1;
        if(np > kMaxProbe)
        {
          np = kMaxProbe;
          S.status |= SSD_ST_OVERFLOW;
        }
        S.vActive[side] = 1;
        S.vLeft[side] = lft; S.vRight[side] = rgt; S.vYStart[side] = yStart; S.vYEnd[side] = yEnd;
        S.vProbe[side] = np;
      }
    }
    __syncthreads();
    SSD_PHASE(0, 5);

    /* ---- phase D: row probes (VerticalEdgePointsDetector :243-312) on freshly closed rows ---- */
    {
      const int np0 = S.vProbe[0], np1 = S.vProbe[1];
      for(int t = tid; t < np0 + np1; t += T)
      {
        const int side = t < np0 ? 0 : 1;
        const int k = side == 0 ? t : t - np0;
        const int y = S.vYStart[side] - 10 * k;
        const int lft = S.vLeft[side], rgt = S.vRight[side];
        /* left edge: first lit pixel of [lft, rgt-1]; right edge: last lit pixel of [lft+1, rgt] */
        const int xa = side == 0 ? lft : lft + 1;
        const int xb = side == 0 ? rgt - 1 : rgt;
        int found = -1;
        if(side == 0)
        {
          for(int c = xa >> 6; c <= (xb >> 6) && found < 0; c++)
          {
            unsigned long long cw = closed_word(im, y, c);
            if(c == (xa >> 6)) cw &= ~0ull << (xa & 63);
            if(c == (xb >> 6)) cw &= ~0ull >> (63 - (xb & 63));
            if(cw)
              found = 64 * c + __ffsll(static_cast<long long>(cw)) - 1;
          }
        }
        else
        {
          for(int c = xb >> 6; c >= (xa >> 6) && found < 0; c--)
          {
            unsigned long long cw = closed_word(im, y, c);
            if(c == (xa >> 6)) cw &= ~0ull << (xa & 63);
            if(c == (xb >> 6)) cw &= ~0ull >> (63 - (xb & 63));
            if(cw)
              found = 64 * c + 63 - __clzll(static_cast<long long>(cw));
          }
        }
        S.vx[side][k] = found;
      }
    }
    __syncthreads();
    SSD_PHASE(0, 6);

    /* compact the probe hits in scan order (y descending), distances to the base line (:708-721): wave 0 the left
     * edge's, wave 1 the right edge's; places by ballot and prefix count */
    if(wave < 2)
    {
      const int side = wave, np = S.vProbe[side];
      int n = 0;
      for(int k0 = 0; k0 < np; k0 += 64)
      {
        const int k = k0 + lane;
        const int x = k < np ? S.vx[side][k] : -1;
        const unsigned long long m = __ballot(x >= 0);
        if(x >= 0)
        {
          const int at = n + __popcll(m & ((1ull << lane) - 1ull));
          const int y = S.vYStart[side] - 10 * k;
          S.vpx[side][at] = x;
          S.vpy[side][at] = y;
          S.vdist[side][at] = fabs(x * S.baseLine.a + y * S.baseLine.b + S.baseLine.c);
        }
        n += __popcll(m);
      }
      if(lane == 0)
      {
        S.vn[side] = n;
        S.vBest[side] = -1;
      }
    }
    __syncthreads();
    SSD_PHASE(0, 7);

    /* findBestPoint (segmentation.cpp:708-728): the element of rank 2n/3 by distance, by counting; ties see below */
    for(int t = tid; t < S.vn[0] + S.vn[1]; t += T)
    {
      const int side = t < S.vn[0] ? 0 : 1;
      const int i = side == 0 ? t : t - S.vn[0];
      const int n = S.vn[side];
      const double di = S.vdist[side][i];
      int rank = 0;
      for(int k = 0; k < n; k++)
      {
        const double dk = S.vdist[side][k];
        rank += (dk < di || (dk == di && k < i)) ? 1 : 0;
      }
      if(rank == 2 * n / 3)
        S.vBest[side] = i;
    }
    __syncthreads();
    SSD_PHASE(0, 8);

    /* The reference sorts with std::ranges::sort (segmentation.cpp:724), which is not stable: among points at exactly
     * the same distance, the one on rank 2n/3 is whatever libstdc++'s introsort leaves there.  Only when that can
     * matter — the selected distance is duplicated AND a duplicate would give another line — the keys are sorted the
     * way the library does it (ssd_sort.h; vdist in place, vx as the payload: neither is needed afterwards). */
    if(wave < 2 && S.vn[wave] > 0 && S.vBest[wave] >= 0)
    {
      const int side = wave, n = S.vn[side], best = S.vBest[side];
      const double D = S.vdist[side][best];
      const double cBest = -S.baseLine.a * S.vpx[side][best] - S.baseLine.b * S.vpy[side][best];
      bool mine = false;
      for(int k = lane; k < n; k += 64)
        if(k != best && S.vdist[side][k] == D && (-S.baseLine.a * S.vpx[side][k] - S.baseLine.b * S.vpy[side][k]) != cBest)
          mine = true;
      const bool ambiguous = __ballot(mine) != 0ull;
      if(ambiguous && lane == 0)
      {
        for(int k = 0; k < n; k++)
          S.vx[side][k] = k;
        gnu_sort_on(SortKeys{ S.vdist[side], S.vx[side] }, n, S.sortStack[side][0], S.sortStack[side][1], S.sortStack[side][2]);
        S.vBest[side] = S.vx[side][2 * n / 3];
      }
    }
    __syncthreads();
  }
  SSD_PHASE(0, 9);

  /* ---- Corners (:731-751), quadrilateral, isConvex (:758-772), imgPointsToWorld (pointcloud.cpp:476-487) ---- */
  if(tid == 0)
  {
    double quad[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    int valid = 0;
    int cornerFound[4] = { 0, 0, 0, 0 };
    LineD vl[2] = { { 0, 0, 0 }, { 0, 0, 0 } };
    int vFound[2] = { 0, 0 };
    if(S.found)
    {
      for(int side = 0; side < 2; side++)
        if(S.vActive[side] && S.vn[side] > 0 && S.vBest[side] >= 0)
        {
          const int bx = S.vpx[side][S.vBest[side]], by = S.vpy[side][S.vBest[side]];
          vl[side] = LineD{ S.baseLine.a, S.baseLine.b, -S.baseLine.a * bx - S.baseLine.b * by };   /* parallel(), :368-371 */
          vFound[side] = 1;
        }
      for(int e = 0; e < 4; e++)
      {
        const int side = (e == kFL || e == kBL) ? 0 : 1;
        double x = S.bounds[e][1][0], y = S.bounds[e][1][1];         /* value_or(bounds.outer) :947-953 */
        if(vFound[side])
        {
          const LineD hl{ static_cast<double>(S.line[e].a), static_cast<double>(S.line[e].b), static_cast<double>(S.line[e].c) };
          double ix, iy;
          if(intersect60(vl[side], hl, ix, iy))
          {
            x = ix; y = iy;
            cornerFound[e] = 1;
          }
        }
        quad[2 * e] = x;
        quad[2 * e + 1] = y;
      }
      /* isConvex: edge vectors q0->q1, q1->q3, q3->q2, q2->q0 must all turn the same way */
      const double vx[4] = { quad[2] - quad[0], quad[6] - quad[2], quad[4] - quad[6], quad[0] - quad[4] };
      const double vy[4] = { quad[3] - quad[1], quad[7] - quad[3], quad[5] - quad[7], quad[1] - quad[5] };
      auto pos = [&](int i, int k) { return vx[i] * vy[k] - vx[k] * vy[i] > 0; };
      const bool positive = pos(0, 1);
      valid = (positive == pos(1, 2) && positive == pos(2, 3) && positive == pos(3, 0)) ? 1 : 0;
    }
    pl.outlineFound = S.found;
    pl.valid = valid;
    for(int k = 0; k < 4; k++)
    {
      pl.quadImg[2 * k] = quad[2 * k];
      pl.quadImg[2 * k + 1] = quad[2 * k + 1];
      pl.quadWorld[2 * k] = P.xMin + quad[2 * k] * P.xToWorld;            /* Projection2D::imageToWorld :84-88 */
      pl.quadWorld[2 * k + 1] = P.yMax - quad[2 * k + 1] * P.yToWorld;
    }
    if(S.status)
      atomicOr(&fs.status, S.status);

    if(dp)
    {
      dp->outline_found = S.found;
      dp->valid = valid;
      for(int k = 0; k < 8; k++) { dp->quad_img[k] = pl.quadImg[k]; dp->quad_world[k] = pl.quadWorld[k]; }
      if(S.found)
      {
        for(int e = 0; e < 4; e++)
        {
          dp->n_edge_pts[e] = S.en[e];
          dp->line[e][0] = S.line[e].a; dp->line[e][1] = S.line[e].b; dp->line[e][2] = S.line[e].c;
          for(int io = 0; io < 2; io++) { dp->bounds[e][io][0] = S.bounds[e][io][0]; dp->bounds[e][io][1] = S.bounds[e][io][1]; }
          dp->corner_found[e] = cornerFound[e];
        }
        dp->base_line[0] = S.baseLine.a; dp->base_line[1] = S.baseLine.b; dp->base_line[2] = S.baseLine.c;
        for(int side = 0; side < 2; side++)
        {
          dp->vedge_found[side] = vFound[side];
          dp->n_vpts[side] = S.vn[side];
          for(int k = 0; k < S.vn[side]; k++) { dp->vpts[side][k][0] = S.vpx[side][k]; dp->vpts[side][k][1] = S.vpy[side][k]; }
          if(vFound[side])
          {
            dp->best_pt[side][0] = S.vpx[side][S.vBest[side]]; dp->best_pt[side][1] = S.vpy[side][S.vBest[side]];
            dp->vline[side][0] = vl[side].a; dp->vline[side][1] = vl[side].b; dp->vline[side][2] = vl[side].c;
          }
        }
      }
    }
  }
  __syncthreads();
  SSD_PHASE(0, 10);

  /* leave the raw image zeroed for the next batch (all its bits lie inside the bounding box) */
  if(!emptyImg)
  {
    const int cy0 = fs.imgYMin[slot], cc0 = fs.imgXMin[slot];
    const int cw = fs.imgXMax[slot] - cc0 + 1, ch = fs.imgYMax[slot] - cy0 + 1;
    for(int idx = tid; idx < cw * ch; idx += T)
    {
      const int ry = idx / cw;
      const size_t o = static_cast<size_t>(cy0 + ry) * P.W64 + cc0 + (idx - ry * cw);
      if(!SSD_CHK(21, o, imgWords))
        continue;
      img[o] = 0ull;                                 /* unconditionally: a load first would make every word a round trip */
      if(img2)
        img2[o] = 0ull;
    }
    if(fromPlanes && tid < 2)
    {
      /* the planes' boxes go with their bits (k_predict resets them as well; this keeps "box empty = plane zero" at all times) */
      const int p = fs.imgPlane[slot][tid];
      if(p != 0xff)
      {
        fs.planeYMin[p] = 0x7fffffff; fs.planeYMax[p] = -1;
        fs.planeXMin[p] = 0x7fffffff; fs.planeXMax[p] = -1;
      }
    }
  }
  SSD_PHASE(0, 11);
}

/* ========================================================================= */
/* K3b: ground quadrilateral and the prepared quadrilateral tests              */

/* Thresholds of k_inquad's cell classification for one live quadrilateral, on the grid of K1's cell boxes.  A cell's box
 * [x0, x1] x [y0, y1] holds points with xMin + x0 / boxX <= x < xMin + (x1 + 1) / boxX (K1 truncates).  Ground: the box lies
 * wholly outside the quadrilateral's strict bounding box iff x1 <= b.x, x0 >= b.y, y1 <= b.z or y0 >= b.w; tread: wholly inside
 * the constant cell iff x0 >= b.x, x1 <= b.y, y0 >= b.z, y1 <= b.w — with a 1e-9 m margin against the truncation's rounding. */
__device__ __forceinline__ int4 live_box_thresholds(const QuadTest &t, bool ground, const PointParams &P)
{
  const double m = 1.0e-9;
  int4 b;
  if(ground)
  {
    b.x = static_cast<int>(floor((t.bxLo - P.xMin - m) * P.boxX)) - 1;
    b.y = static_cast<int>(ceil((t.bxUp - P.xMin + m) * P.boxX));
    b.z = static_cast<int>(floor((t.byLo - P.yMin - m) * P.boxY)) - 1;
    b.w = static_cast<int>(ceil((t.byUp - P.yMin + m) * P.boxY));
  }
  else
  {
    b.x = static_cast<int>(ceil((t.fx0 - P.xMin + m) * P.boxX));
    b.y = static_cast<int>(floor((t.fx1 - P.xMin - m) * P.boxX)) - 1;
    b.z = static_cast<int>(ceil((t.fy0 - P.yMin + m) * P.boxY));
    b.w = static_cast<int>(floor((t.fy1 - P.yMin - m) * P.boxY)) - 1;
  }
  return b;
}

/* one wave per frame: lane k < kMaxPlateaus builds the test of plateau k, lane kGroundAcc the ground's */
__global__ __launch_bounds__(64) void k_quads(Params P, FrameState *__restrict__ st, int nframes, DebugFrame *__restrict__ dbg)
{
  static_assert(kMaxPlateaus + 1 <= 64, "one lane per accumulator");
  const int frame = blockIdx.x, lane = threadIdx.x;
  if(frame >= nframes)
    return;
  FrameState &fs = st[frame];
  SSD_PHASE(1, 0);
  if(lane <= kMaxPlateaus)
  {
    fs.sumZ[lane] = 0;
    fs.cnt[lane] = 0;
  }
  for(int i = lane; i < kMaxGroundStrips; i += 64)
    fs.groundStripMax[i] = -1;
  const int first = fs.firstStep, last = fs.firstStep + fs.nStepImages;
  const int groundInd = fs.groundInd;
  const bool mine = lane >= first && lane < last && lane < kMaxPlateaus;
  /* every lane with an image fetches its quadrilateral before anyone knows whether it is valid: one round trip for both */
  double quad[8];
#pragma unroll
  for(int k = 0; k < 8; k++)
    quad[k] = mine ? fs.pl[lane].quadWorld[k] : 0.0;
  const bool valid = mine && fs.pl[lane].valid;
  const unsigned long long validMask = __ballot(valid);
  const int firstValid = validMask ? __ffsll(static_cast<long long>(validMask)) - 1 : -1;
  const bool groundLane = lane == kGroundAcc && firstValid >= 0 && groundInd >= 0;
  int err = 0;
  /* the live tests once more in LDS, for the check of their maps (quad_cell_unsound) that all lanes share below */
  __shared__ QuadTest sT[kMaxLive];
  __shared__ QuadEdgesD sW[kMaxLive];
  __shared__ unsigned int sBad[kMaxLive];
  QuadEdgesF ef;
  int mySlot = -1;
  SSD_PHASE_IF(lane == firstValid, 1, 1);
  {
    /* calcGroundQuadrilateral (pointcloud.cpp:489-512) from the front edge of the first valid step */
    const int srcLane = firstValid >= 0 ? firstValid : 0;
    double q[4];
#pragma unroll
    for(int k = 0; k < 4; k++)
      q[k] = __shfl(quad[k], srcLane);
    if(groundLane)
    {
      const double yMin = P.yMin;
      if(q[1] < q[3])
      {
        quad[0] = q[0]; quad[1] = yMin;
        quad[2] = q[2] + (q[3] - yMin) * (q[3] - q[1]) / (q[2] - q[0]); quad[3] = yMin;
      }
      else
      {
        quad[0] = q[0] + (q[1] - yMin) * (q[1] - q[3]) / (q[0] - q[2]); quad[1] = yMin;
        quad[2] = q[2]; quad[3] = yMin;
      }
      quad[4] = q[0]; quad[5] = q[1];
      quad[6] = q[2]; quad[7] = q[3];
#pragma unroll
      for(int k = 0; k < 8; k++)
        fs.groundQuadWorld[k] = quad[k];
    }
  }
  SSD_PHASE_IF(lane == firstValid, 1, 2);
  if(groundLane || valid)
  {
    /* built in registers (ssd_quadtest.h), stored to the frame's table once */
    QuadTest t;
    SSD_PHASE_IF(lane == firstValid, 1, 3);
    build_quad_test(quad, t);
    err = t.err;
    SSD_PHASE_IF(lane == firstValid, 1, 4);
    /* one quadrilateral the reference would throw on ends the frame (quadrilateralTest.cpp:283-372) */
    const bool threwHere = __ballot(err != 0) != 0ull;
    /* The live quadrilaterals go into a compact table (at most one per image slot + the ground): k_inquad copies it
     * whole, without asking which entries matter.  Slot order = accumulator order (treads ascending, ground last). */
    const unsigned long long liveMask = threwHere ? 0ull : __ballot(true);
    if(!threwHere)
    {
      const int slot = __popcll(liveMask & ((1ull << lane) - 1ull));
      if(slot < kMaxLive)
      {
        fs.qtLive[slot] = t;
        QuadGridSegs sg;
        build_grid_segs(t, P.pt.xMin, P.pt.yMin, P.pt.boxX, P.pt.boxY, sg);
        fs.segLive[slot] = sg;
        QuadEdgesD w;
        quad_edges_coeffs(t, sg.ok, P.pt.xMin, P.pt.xMax, P.pt.yMin, P.pt.yMax, w, ef);
        sT[slot] = t;
        sW[slot] = w;
        sBad[slot] = 0u;
        mySlot = slot;
        fs.liveBox[slot] = live_box_thresholds(t, groundLane, P.pt);
        fs.liveAcc[slot] = static_cast<unsigned char>(lane);
        /* the groups of 4 height bins this accumulator's plateau occupies (matched against the cells' masks) */
        const PlateauState &pl = fs.pl[groundLane ? groundInd : lane];
        unsigned int groups = 0u;
        const int effLo = pl.effLo, effHi = pl.effHi;
        for(int b = effLo; b <= effHi; b++)
          groups |= 1u << (b / kBinsPerGroup);
        fs.liveGroups[slot] = groups;
      }
    }
  }
  {
    /* k_inquad's single-precision edge tests (ssd_quadtest.h, build_quad_edges): the reference's map of each live quadrilateral is
     * checked cell by cell - nine cells each, dealt out to the wave's lanes (one lane doing its own quadrilateral's nine: 0.086 ms
     * per launch of 1024 frames against 0.031) - and a quadrilateral whose map accepts points beyond an edge gets no margin */
    const int nL = __popcll(__ballot(mySlot >= 0));
    __syncthreads();
    for(int item = lane; item < nL * 9; item += 64)
    {
      const int sl = item / 9, cell = item - 9 * sl;
      if(sW[sl].fine != 0 && quad_cell_unsound(sT[sl], sW[sl], cell / 3, cell % 3))
        atomicOr(&sBad[sl], 1u);
    }
    __syncthreads();
    if(mySlot >= 0)
    {
      ef.m = quad_edges_margin(sW[mySlot], sBad[mySlot] != 0u);
      fs.edgeLive[mySlot] = ef;
    }
  }
  SSD_PHASE_IF(lane == firstValid, 1, 5);
  const bool threw = __ballot(err != 0) != 0ull;
  const bool active = (groundLane || valid) && !threw;
  if(lane <= kMaxPlateaus)
    fs.accActive[lane] = active ? 1 : 0;
  const unsigned long long activeMask = __ballot(active);
  /* bin -> slot of the live table (0xff = no live quadrilateral takes this bin) */
  for(int b = lane; b < kMaxBins; b += 64)
  {
    const int p = b < P.nBins ? fs.lut[b] : 0xff;
    const int acc = p == groundInd ? kGroundAcc : p;
    const bool live = p != 0xff && ((activeMask >> acc) & 1ull);
    fs.lutLive[b] = live ? static_cast<unsigned char>(__popcll(activeMask & ((1ull << acc) - 1ull))) : static_cast<unsigned char>(0xff);
  }

  SSD_PHASE_IF(lane == firstValid, 1, 6);
  unsigned int wanted = 0u;
  for(int b = lane; b < P.nBins; b += 64)
  {
    const int p = fs.lut[b];
    if(p != 0xff && ((activeMask >> (p == groundInd ? kGroundAcc : p)) & 1ull))
      wanted |= 1u << (b / kBinsPerGroup);
  }
#pragma unroll
  for(int o = 32; o > 0; o >>= 1)
    wanted |= __shfl_xor(wanted, o);
  if(lane == 0)
  {
    fs.firstValidInd = firstValid;
    if(threw)
      fs.status |= SSD_ST_THROW;
    fs.wantedQuads = wanted;
    fs.anyActive = activeMask != 0ull ? 1u : 0u;
    fs.nLive = min(__popcll(activeMask), kMaxLive);
  }
  SSD_PHASE_IF(lane == firstValid, 1, 7);
  if(dbg)
  {
    ssd_debug_frame &d = dbg[frame].d;
    if(lane == 0)
      d.first_valid_ind = firstValid;
    if(lane == kGroundAcc)
    {
      for(int k = 0; k < 8; k++)
        d.ground_quad_world[k] = groundLane ? fs.groundQuadWorld[k] : 0.0;
      d.ground_quad_err = groundLane ? err : 0;
    }
    if(lane >= first && lane < last && lane < kMaxPlateaus)
      d.plateaus[lane].quad_err = valid ? err : 0;
  }
}

/* ========================================================================= */
/* K4: in-quadrilateral filter, z sums, ground image                           */

/* What of the ground image is ever looked at: detectFrontEdge (segmentation.cpp:879-917) closes it and BottomScanner
 * (:159-241) probes the pixel columns x_j = W/2 + 50 k bottom-up, stopping at the image centre.  A closed pixel is a function
 * of the raw 5 x 5 neighbourhood (ssd_closing.h, closed_scan_column), so k_final reads raw pixels only in the columns
 * x_j - 2 .. x_j + 2 and the rows >= H/2 - 1.  Outside debug capture k_inquad therefore sets just those: a twentieth of the
 * ground pixels, few enough to go to memory one atomic each — no LDS windows, no flushes, no keys in the production kernel.
 * (kGroundScanStep = BottomScanner's column distance; the same constants are k_final's.) */
constexpr int kGroundScanStep = 50;
__device__ __forceinline__ int ground_scan_x0(int W) { return (W / 2) % kGroundScanStep; }
__device__ __forceinline__ int ground_strip_row0(int H) { return H / 2 - 1; }
/* is pixel column ix within two of a scan column x_j = x0 + 50 j, j any integer (also j = -1 and beyond the last scanned
 * column: k_final clears those strips with the others)?  u / 50 by multiply-high with a constant that IS exact over the whole
 * range a pixel column can take (ix < 8192: u < 8300; proven at compile time below — round 3's 1311 >> 16 was off by one from
 * u = 4699 on and only the predicate happened to survive it). */
constexpr unsigned int strip_div50(unsigned int u) { return (u * 5243u) >> 18; }
constexpr bool strip_div50_exact()
{
  for(unsigned int u = 0; u < 16384u; u++)
    if(strip_div50(u) != u / 50u)
      return false;
  return true;
}
static_assert(strip_div50_exact(), "strip_div50 must equal u / 50 for every pixel column of an image up to 8192 wide");
__device__ __forceinline__ bool ground_strip_column(int ix, int x0)
{
  const unsigned int u = static_cast<unsigned int>(ix + 2 - x0 + kGroundScanStep);       /* >= 3 */
  const unsigned int r = u - kGroundScanStep * strip_div50(u);
  return r <= 4u;
}
/* the same with the strip's number (0 = the strip left of the first scan column; below kMaxGroundStrips for any width up to
 * 8192) and whether ix is the strip's CENTRE column, i.e. a scan column itself */
__device__ __forceinline__ bool ground_strip_of(int ix, int x0, int &strip, bool &centre)
{
  const unsigned int u = static_cast<unsigned int>(ix + 2 - x0 + kGroundScanStep);
  const unsigned int q = strip_div50(u), r = u - kGroundScanStep * q;
  strip = static_cast<int>(q);
  centre = r == 2u;
  return r <= 4u;
}
static_assert((8191 + 2 + kGroundScanStep) / kGroundScanStep < kMaxGroundStrips, "a strip number for every pixel column");

template<bool FULL>
struct InquadLds
{
  unsigned long long wins[FULL ? kThreads / 64 : 1][FULL ? kWinWords : 1];
  unsigned int wmiss[FULL ? kThreads / 64 : 1][kWaveMissWords];
  ImageBox box[1];
  QuadTest qts[kMaxLive];                       /* FrameState::qtLive: slot k = accumulator liveAcc[k] */
  QuadGridSegs segs[kMaxLive];                  /* FrameState::segLive */
  QuadEdgesF edges[kMaxLive + 1];               /* FrameState::edgeLive with PreXY::dE0 folded into m; row kMaxLive ("no quadrilateral"): m = infinity */
  K1Consts kc;                                  /* what only the rare double-precision paths need (as K1) */
  unsigned char lut[kMaxBins];                  /* bin -> live slot, kMaxLive = none */
  unsigned long long lsum[kMaxLive][8];
  unsigned int lcnt[kMaxLive][8];
  unsigned int lOob;
  unsigned short cellList[kMaxCellsPerBlockInquad];
  unsigned int listScratch[2 * kWavesPerBlock];
  unsigned int liveGroups[kMaxLive];
  int4 liveBox[kMaxLive];                       /* per live quadrilateral: thresholds of the cell classification on K1's grid */
  unsigned char liveAcc[kMaxLive];
  int nLive, groundSlot;
  int nextGroup;                                /* the walk's groups of four cells are dealt out to the waves as they come free */
  int stripMax[kMaxGroundStrips];               /* FrameState::groundStripMax as the block found it, raised by its own centre-column pixels */
};

template<int SRC, bool FULL, bool CHECKS>
__device__ __forceinline__ void inquad_block(InquadLds<FULL> &L, const float *__restrict__ xyz, size_t strideFloats, const PointParams &P,
                                             const PreXY &Q, const PixelParams &X, FrameState *__restrict__ st,
                                             unsigned long long *__restrict__ groundImg,
                                             const uint2 *__restrict__ tileMasks, size_t tileMaskStride, int chunkPoints, const DepthSrc &D,
                                             const int frame, const int chunkIdx)
{
  auto &wins = L.wins;
  auto &wmiss = L.wmiss;
  ImageBox (&box)[1] = L.box;
  QuadTest (&qts)[kMaxLive] = L.qts;
  QuadGridSegs (&segs)[kMaxLive] = L.segs;
  unsigned char (&lut)[kMaxBins] = L.lut;
  unsigned long long (&lsum)[kMaxLive][8] = L.lsum;
  unsigned int (&lcnt)[kMaxLive][8] = L.lcnt;
  unsigned int &lOob = L.lOob;
  unsigned short (&cellList)[kMaxCellsPerBlockInquad] = L.cellList;
  unsigned int (&listScratch)[2 * kWavesPerBlock] = L.listScratch;
  unsigned int (&liveGroups)[kMaxLive] = L.liveGroups;
  int4 (&liveBox)[kMaxLive] = L.liveBox;
  unsigned char (&liveAcc)[kMaxLive] = L.liveAcc;
  int &nLive = L.nLive;
  int &groundSlot = L.groundSlot;

  const int tid = threadIdx.x, lane = tid & 63;
  FrameState &fs = st[frame];
  /* Everything the block needs of the frame's state is requested at once, the "anything to do?" flag first: one memory round
   * trip for the lot (as code in sequence — flag, early exit, tables — it was two, a tenth of a block's life). */
  const unsigned int anyActive = fs.anyActive;              /* block-uniform: set by k_quads */
  const unsigned char lutMine = tid < kMaxBins ? fs.lutLive[tid] : static_cast<unsigned char>(0xff);
  const unsigned char accMine = tid < kMaxLive ? fs.liveAcc[tid] : static_cast<unsigned char>(0);
  const unsigned int groupsMine = tid < kMaxLive ? fs.liveGroups[tid] : 0u;
  const int stripMaxMine = (!FULL && tid < kMaxGroundStrips) ? fs.groundStripMax[tid] : -1;
  const int4 boxMine = tid < kMaxLive ? fs.liveBox[tid] : make_int4(0, 0, 0, 0);
  const int nLiveG = fs.nLive;
  const unsigned char groundActive = fs.accActive[kGroundAcc];
  constexpr int qtWords = kMaxLive * static_cast<int>(sizeof(QuadTest) / 4), sgWords = kMaxLive * static_cast<int>(sizeof(QuadGridSegs) / 4);
  constexpr int qtPer = (qtWords + kThreads - 1) / kThreads, sgPer = (sgWords + kThreads - 1) / kThreads;
  unsigned int qtw[qtPer], sgw[sgPer];
  constexpr int egWords = kMaxLive * static_cast<int>(sizeof(QuadEdgesF) / 4);
  static_assert(sizeof(QuadEdgesF) == 64 && egWords + 16 <= 2 * kThreads, "two words of the edge table per thread");
  float edgeMine[2];
#pragma unroll
  for(int k = 0; k < 2; k++)
    edgeMine[k] = tid + k * kThreads < egWords ? reinterpret_cast<const float *>(fs.edgeLive)[tid + k * kThreads] : 0.0f;
  {
    /* the live quadrilateral tests as 32-bit words: the whole table, unconditionally — asking "how many are live?" first
     * would make every element two dependent round trips */
    const unsigned int *src = reinterpret_cast<const unsigned int *>(fs.qtLive);
    const unsigned int *srcS = reinterpret_cast<const unsigned int *>(fs.segLive);
#pragma unroll
    for(int k = 0; k < qtPer; k++)
      qtw[k] = tid + k * kThreads < qtWords ? src[tid + k * kThreads] : 0u;
#pragma unroll
    for(int k = 0; k < sgPer; k++)
      sgw[k] = tid + k * kThreads < sgWords ? srcS[tid + k * kThreads] : 0u;
  }
  if(!anyActive)
    return;
  BlockPhase ph(1);
  if(tid == 0)
  {
    box[0] = ImageBox{ 0x7fffffff, -1, 0x7fffffff, -1 };
    lOob = 0;
    nLive = nLiveG;
    groundSlot = (nLiveG > 0 && groundActive) ? nLiveG - 1 : -1;     /* the ground is the last accumulator */
    L.nextGroup = 0;
    K1Consts &c = L.kc;
#pragma unroll
    for(int i = 0; i < 9; i++)
      c.a[i] = P.a[i];
    c.b[0] = P.b[0]; c.b[1] = P.b[1]; c.b[2] = P.b[2];
    c.xMin = P.xMin; c.xMax = P.xMax; c.yMin = P.yMin; c.yMax = P.yMax; c.zMin = P.zMin; c.zMax = P.zMax;
    c.boxX = P.boxX; c.boxY = P.boxY;
    c.recip = P.recip;
    c.xToImage = X.xToImage; c.yToImage = X.yToImage;
  }
  if(tid < kMaxBins)
    lut[tid] = lutMine == 0xff ? static_cast<unsigned char>(kMaxLive) : lutMine;      /* bin -> slot of the live table, kMaxLive = nothing to do for this bin */
  /* the edges for the single-precision test: a row per live quadrilateral, m (word 12) with the bound of d that does not depend on
   * the point; the row behind them answers "not for sure" to everything - the row of a bin without quadrilateral */
#pragma unroll
  for(int k = 0; k < 2; k++)
  {
    const int w = tid + k * kThreads;
    if(w < egWords + 16)
      reinterpret_cast<float *>(L.edges)[w] = w < egWords ? ((w & 15) == 12 ? edgeMine[k] + Q.dE0 : edgeMine[k]) : ((w & 15) == 12 ? INFINITY : 0.0f);
  }
  if(!FULL && tid < kMaxGroundStrips)
    L.stripMax[tid] = stripMaxMine;
  if(tid < kMaxLive)
  {
    liveAcc[tid] = accMine;
    liveGroups[tid] = groupsMine;
    liveBox[tid] = boxMine;                     /* thresholds of the cell classification (k_quads, live_box_thresholds) */
  }
  for(int i = tid; i < kMaxLive * 8; i += kThreads)
  {
    (&lsum[0][0])[i] = 0ull;
    (&lcnt[0][0])[i] = 0u;
  }
  if(FULL)
  {
    for(int i = tid; i < (kThreads / 64) * kWinWords; i += kThreads)
      (&wins[0][0])[i] = 0ull;
    if(tid < kThreads / 64)
      wavemiss_init(wmiss[tid]);
  }
  {
    unsigned int *dst = reinterpret_cast<unsigned int *>(qts);
    unsigned int *dstS = reinterpret_cast<unsigned int *>(segs);
#pragma unroll
    for(int k = 0; k < qtPer; k++)
      if(tid + k * kThreads < qtWords)
        dst[tid + k * kThreads] = qtw[k];
#pragma unroll
    for(int k = 0; k < sgPer; k++)
      if(tid + k * kThreads < sgWords)
        dstS[tid + k * kThreads] = sgw[k];
  }
  __syncthreads();
  ph.mark(0);                                               /* tables in LDS */

  /* strideFloats counts floats, or 16-bit depth values for kSrcDepth16 */
  const float *base = SRC == kSrcDepth16
    ? reinterpret_cast<const float *>(reinterpret_cast<const unsigned short *>(xyz) + static_cast<size_t>(frame) * strideFloats)
    : xyz + static_cast<size_t>(frame) * strideFloats;
  const int begin = chunkIdx * chunkPoints;
  const int end = min(begin + chunkPoints, P.nPoints);
  unsigned long long *gimg = groundImg + static_cast<size_t>(frame) * X.H * X.W64;
  const int copy = lane & 7;
  const unsigned int imgWords = static_cast<unsigned int>(X.H) * X.W64;
  unsigned long long *ww = wins[FULL ? tid >> 6 : 0];
  unsigned int *wm = wmiss[FULL ? tid >> 6 : 0];
  WaveWindow win;
  win.limitWords = imgWords;                   /* the ground image: one per frame */
  /* strips only (!FULL): the scan columns' offset, the first row of interest, the lane's rows written so far */
  const int stripX0 = ground_scan_x0(X.W), stripRow0 = ground_strip_row0(X.H);
  int gy0 = 0x7fffffff, gy1 = -1;
  unsigned int *gimg32 = reinterpret_cast<unsigned int *>(gimg);

  /* calcAverageZ (pointcloud.cpp:574-581) as an order-independent fixed-point sum: a thread walks down a
   * camera column, so consecutive hits nearly always belong to the same quadrilateral — the running sum
   * stays in registers and goes to LDS only when the quadrilateral changes */
  int curQ = -1;
  unsigned long long accZ = 0;                               /* modulo 2^64: see z_plus_magic_bits */
  unsigned int accN = 0, oob = 0;
  auto flushAcc = [&]()
  {
    if(curQ >= 0 && accN)
    {
      atomicAdd(&lsum[curQ][copy], accZ - static_cast<unsigned long long>(accN) * static_cast<unsigned long long>(kMagicBits));
      atomicAdd(&lcnt[curQ][copy], accN);
    }
  };

  /* only the cells that hold a bin of a live quadrilateral are walked, four per wave iteration (cell_list_build) */
  const int cell0 = begin / kCell;
  const int nCells = (end - begin + kCell - 1) / kCell;
  /* A cell is walked only if it can change a sum.  Ground: its points INSIDE the ground quadrilateral are added (and
   * rastered) — a cell whose bounding box lies outside the quadrilateral's (strict) bounding box has none.  Treads: k_raster
   * already summed every point of the plateau, so the points OUTSIDE the quadrilateral are taken off here — a cell whose
   * bounding box lies inside the quadrilateral's large constant cell (QuadTest::fx0 ..) has none.  The box covers all
   * in-range points of the cell, whatever their plateau: conservative for each of them. */
  const unsigned int wantedQuads = fs.wantedQuads;
  const int gSlot = groundSlot;
  const int count = cell_list_build(tileMasks + static_cast<size_t>(frame) * tileMaskStride + cell0, nCells, X.cellCols,
                                    [&](const uint2 info)
                                    {
                                      if((info.x & wantedQuads) == 0u)
                                        return false;
                                      /* the box on K1's 256 x 256 grid against each live quadrilateral's thresholds on the
                                       * same grid (liveBox, computed once per block from the doubles,
                                       * margin included) */
                                      const int x0 = info.y & 0xffu, x1 = (info.y >> 8) & 0xffu, y0 = (info.y >> 16) & 0xffu, y1 = info.y >> 24;
                                      bool need = false;
                                      for(int q = 0; q < nLive; q++)
                                      {
                                        if((info.x & liveGroups[q]) == 0u)
                                          continue;
                                        const int4 b = liveBox[q];
                                        if(q == gSlot)
                                          need = need || !(x1 <= b.x || x0 >= b.y || y1 <= b.z || y0 >= b.w);      /* not wholly outside */
                                        else
                                        {
                                          /* wholly inside the constant cell, or inside all four edges (the tread is turned
                                           * against the axes) */
                                          const bool in = (x0 >= b.x && x1 <= b.y && y0 >= b.z && y1 <= b.w) || grid_box_inside(segs[q], x0, x1, y0, y1);
                                          need = need || !in;
                                        }
                                      }
                                      return need;
                                    }, cellList, listScratch);
  ph.mark(1);                                               /* cell list */
  const int nGroups = (count + 3) >> 2;
  /* The groups of four listed cells go to the waves as they come free (a counter in LDS, one returning atomic per group,
   * asked for one group ahead so that its latency and the group's loads hide behind the current group's arithmetic): a
   * quarter of the list each left the block waiting 8 % of its life for its slowest wave — treads' edges cost several
   * times the ground's interior. */
  auto grab = [&]()
  {
    int v = 0;
    if(lane == 0)
      v = atomicAdd(&L.nextGroup, 1);
    return __builtin_amdgcn_readfirstlane(v);
  };
  int g = grab();
  F3 v[kPts], vn[kPts];
  /* the lane's copies of constants that are the SECOND scalar operand of an instruction (one is allowed), as in K1 */
  float zc3 = Q.zc[3], zh0 = Q.zH0;
  f32x2 c3xy = f32x2{ Q.c[3][0], Q.c[3][1] };
  asm volatile("" : "+v"(zc3), "+v"(zh0), "+v"(c3xy));
  /* the coarse test "can the bottom scan see this ground pixel, can it lie outside the image?" on d (see the loop): within two
   * pixels of the border; a row from the strips' first less one on; a column within 3.5 of a strip's centre (strip: 2.5 either
   * side), as the distance of (px + 2 - x0 + 50 - 2.5) / 50 from the nearest integer */
  const float pixBorder = 0.5f - 2.0f / static_cast<float>(min(X.W, X.H));
  const float pixRowThr = 0.5f - static_cast<float>(stripRow0 - 1) / static_cast<float>(X.H);
  const float pixColA = static_cast<float>(X.W) * 0.02f;
  const float pixColB = (0.5f * static_cast<float>(X.W) + static_cast<float>(2 - stripX0 + kGroundScanStep) - 2.5f) * 0.02f + 0.5f;
  if(g < nGroups)
    load_cell<SRC>(base, cell0, cellList, 4 * g + (lane >> 4), count, lane, P.nPoints, v, D);
  while(g < nGroups)
  {
    const int gNext = grab();
    if(gNext < nGroups)
      load_cell<SRC>(base, cell0, cellList, 4 * gNext + (lane >> 4), count, lane, P.nPoints, vn, D);
    unsigned int key[kPts];
    #pragma unroll
    for(int j = 0; j < kPts; j++)
    {
      key[j] = kNoPixel;
      const F3 p{ v[j].x, v[j].y, v[j].z };
      /* Round 6: every decision in single precision first, as K1 takes them (ssd_prexy.h) - the x / y range, the z range and the
       * height bin, then the quadrilateral as four half-planes on the same d (ssd_quadtest.h, build_quad_edges) - and kept as the
       * wave's lane masks; the reference's doubles - all three rows, the compares, the bin, QuadrilateralTest with its box, map and
       * segments - only for the points within a bound of a limit, a bin edge or a quadrilateral's edge: one wave-uniform block, a wave
       * slot in twenty.  (Until here that test ran for every point of every cell an edge passes through, nested and divergent:
       * 169 vector instructions per point slot, two thirds of the kernel - profiles/r06_k4_ground_kernel.txt.) */
      const unsigned long long mValid = __ballot(p.z > 0.0f);
      f32x2 d = pre_xy(Q, c3xy, p.x, p.y, p.z);
      const float M = absmax2(d.x, d.y);
      unsigned long long mInxy = __ballot(M < Q.lo);
      unsigned long long mMaybexy = ~__ballot(M > Q.hi);
      const float M3 = absmax3(p.x, p.y, p.z);
      if constexpr(CHECKS)
      {
        /* (the rare configurations' tests, as K1's: launch_inquad picks the instantiation) */
        const unsigned long long mFar = Q.checkInput ? ~__ballot(M3 <= Q.maxInput) : 0ull;
        mInxy &= ~mFar;
        mMaybexy |= mFar;
      }
      const float t = __builtin_fmaf(Q.zc[0], p.x, __builtin_fmaf(Q.zc[1], p.y, __builtin_fmaf(Q.zc[2], p.z, zc3)));
      const float g = __builtin_amdgcn_fractf(t) - 0.5f;
      const float h = __builtin_fmaf(M3, Q.zNegK, zh0);
      unsigned long long mSurez = __ballot(__builtin_fabsf(g) < h);
      const unsigned long long mInz = __ballot(__float_as_uint(t) < Q.zTopBits);
      if constexpr(CHECKS)
      {
        if(Q.zCheckTop)
          mSurez &= __ballot(__builtin_fabsf(t - Q.zTop) > 0.5f - h);
      }
#if defined(SSD_SABOTAGE_PRE) && (SSD_SABOTAGE_PRE & 1)
      mSurez = ~0ull;
#endif
      const unsigned int b = cvt_u32_f32(t);
      const unsigned long long mInSure = mValid & mInz & mInxy & mSurez;
      unsigned long long mSlow = mValid & mMaybexy & (~mSurez | (mInz & ~mInxy));
      /* the bin's quadrilateral (row kMaxLive: none) and its four edges on d */
      unsigned int q = kMaxLive;
      if(__builtin_amdgcn_inverse_ballot_w64(mInSure))
        q = SSD_CHK(28, b, kMaxBins) ? lut[b] : kMaxLive;
      const unsigned long long mLive = __ballot(q != static_cast<unsigned int>(kMaxLive));
      const QuadEdgesF &E = L.edges[q];
      const float4 gx = *reinterpret_cast<const float4 *>(E.gx), gy = *reinterpret_cast<const float4 *>(E.gy), g2 = *reinterpret_cast<const float4 *>(E.g2);
      f32x2 e01 = __builtin_elementwise_fma(f32x2{ gx.x, gx.y }, f32x2{ d.x, d.x }, f32x2{ g2.x, g2.y });
      f32x2 e23 = __builtin_elementwise_fma(f32x2{ gx.z, gx.w }, f32x2{ d.x, d.x }, f32x2{ g2.z, g2.w });
      e01 = __builtin_elementwise_fma(f32x2{ gy.x, gy.y }, f32x2{ d.y, d.y }, e01);
      e23 = __builtin_elementwise_fma(f32x2{ gy.z, gy.w }, f32x2{ d.y, d.y }, e23);
      const float emin = min3_f32(e01.x, e01.y, min_f32(e23.x, e23.y));
      const float hq = __builtin_fmaf(M3, Q.dK, E.m);              /* infinity for the row "none" and for a quadrilateral single precision does not serve */
      /* (Measured and not kept: a flag on the list for the cells that lie wholly inside their quadrilaterals - the ground's interior,
       * 58 % of the cells walked - and a wave-uniform branch around these twelve instructions: 0.62 -> 0.70 ms, the branch costs more
       * than it skips.) */
#if defined(SSD_SABOTAGE_PRE) && (SSD_SABOTAGE_PRE & 4)   /* tools: the band along the edges NOT handed to the doubles - the tests built for it must fail */
      const unsigned long long mSureQ = mLive;
#else
      const unsigned long long mSureQ = __ballot(__builtin_fabsf(emin) > hq);
#endif
      const unsigned long long mOutQ = __ballot(emin < 0.0f);
      unsigned long long mGround = __ballot(q == static_cast<unsigned int>(gSlot));
      /* ground: the points inside its quadrilateral count; treads: the points outside theirs (k_raster summed the plateau whole) */
      unsigned long long mCount = mSureQ & (mGround ^ mOutQ);
      mSlow |= mLive & ~mSureQ;
      if(__builtin_expect(mSlow != 0ull, 0))
      {
        /* rare, wave-uniform so that the masks stay scalars: the reference's arithmetic, all of it, for the lanes it is for */
        const K1ConstsLds c = k1_consts(L.kc);
        const double x = p.x, y = p.y, z = p.z;
        double wx = (c->a[0] * x + c->a[1] * y) + c->a[2] * z;
        double wy = (c->a[3] * x + c->a[4] * y) + c->a[5] * z;
        double wz = (c->a[6] * x + c->a[7] * y) + c->a[8] * z;
        wx = wx + c->b[0];
        wy = wy + c->b[1];
        wz = wz + c->b[2];
        const bool mine = __builtin_amdgcn_inverse_ballot_w64(mSlow);
        const bool inRange = (wx > c->xMin) & (wx < c->xMax) & (wy > c->yMin) & (wy < c->yMax) & (wz > c->zMin) & (wz < c->zMax);
        unsigned int qD = kMaxLive;
        bool counts = false;
        if(mine && inRange)
        {
          const int bD = static_cast<int>((wz - c->zMin) * c->recip);                       /* height_bin */
          qD = SSD_CHK(29, bD, kMaxBins) ? lut[bD] : kMaxLive;
          if(qD != static_cast<unsigned int>(kMaxLive))
          {
            const QuadTest &tq = qts[qD];
            const bool fast = wx >= tq.fx0 && wx < tq.fx1 && wy >= tq.fy0 && wy < tq.fy1;
            counts = (fast || quad_test(tq, wx, wy)) == (qD == static_cast<unsigned int>(gSlot));
          }
        }
        mCount = (mCount & ~mSlow) | __ballot(counts);
        mGround = (mGround & ~mSlow) | __ballot(qD == static_cast<unsigned int>(gSlot));
        q = mine ? qD : q;
        d.x = mine ? static_cast<float>((wx - c->xMin) * c->boxX * 0.00390625 - 0.5) : d.x;      /* D rounded once: inside PreXY::dE0 */
        d.y = mine ? static_cast<float>((wy - c->yMin) * c->boxY * 0.00390625 - 0.5) : d.y;
      }
      if(__builtin_amdgcn_inverse_ballot_w64(mCount))
      {
        /* calcAverageZ's summand in the reference's doubles (world_z_flat's row) */
        double wz = (P.a[6] * static_cast<double>(p.x) + P.a[7] * static_cast<double>(p.y)) + P.a[8] * static_cast<double>(p.z);
        wz = wz + P.b[2];
        if(static_cast<int>(q) != curQ)
        {
          flushAcc();
          curQ = static_cast<int>(q); accZ = 0; accN = 0;
        }
        accZ += static_cast<unsigned long long>(z_plus_magic_bits(wz));                        /* the constant's bits come off at the flush */
        accN++;
      }
      /* projectToBinaryImage(pointsInQuadri) (pointcloud.cpp:531) for the counted ground points.  Outside debug capture only the
       * pixels the bottom scan can see are wanted (above), and a pixel outside the image (quirk Q5) is counted: on the same d, a
       * point whose row lies above the strips' first by more than a pixel, or whose column lies more than a pixel off every strip,
       * and that is not within two pixels of the image's border, needs no pixel at all - five instructions say so for nineteen
       * ground points in twenty */
      unsigned long long mPixel = mCount & mGround;
      if(!FULL)
      {
        /* (no branch around these six for the waves without a counted ground point: a branch costs this loop more) */
        const unsigned long long mBorder = __ballot(M > pixBorder);
        const unsigned long long mRow = __ballot(d.y <= pixRowThr);
        const float gc = __builtin_amdgcn_fractf(__builtin_fmaf(d.x, pixColA, pixColB)) - 0.5f;
        const unsigned long long mCol = __ballot(__builtin_fabsf(gc) < 0.07f);
        mPixel &= mSlow | mBorder | (mRow & mCol);
      }
      if(FULL ? mPixel != 0ull : __builtin_expect(mPixel != 0ull, 0))
      {
        if(__builtin_amdgcn_inverse_ballot_w64(mPixel))
        {
          /* the pixel from d where single precision is certain of it (make_pre_pixel(), as K1's candidates: farther from every
           * pixel edge than the bound for this magnitude - such a pixel lies inside the image), else Projection2D::worldToImage
           * in doubles with the image's bounds */
          const float px = __builtin_fmaf(d.x, X.fW, X.fHalfW), py = __builtin_fmaf(d.y, X.fNegH, X.fHalfH);
          const f32x2 gg = f32x2{ __builtin_amdgcn_fractf(px), __builtin_amdgcn_fractf(py) } + f32x2{ -0.5f, -0.5f };
          const float hp = __builtin_fmaf(M3, X.pxNegK, X.pxH0);
          int ix = static_cast<int>(cvt_u32_f32(px)), iy = static_cast<int>(cvt_u32_f32(py));
          bool inside = true;
#if defined(SSD_SABOTAGE_PRE) && (SSD_SABOTAGE_PRE & 2)
          if(false)
#else
          if(!(absmax2(gg.x, gg.y) < hp))
#endif
          {
            const K1ConstsLds c = k1_consts(L.kc);
            const double x = p.x, y = p.y, z = p.z;
            double wx = (c->a[0] * x + c->a[1] * y) + c->a[2] * z;
            double wy = (c->a[3] * x + c->a[4] * y) + c->a[5] * z;
            wx = wx + c->b[0];
            wy = wy + c->b[1];
            ix = static_cast<int>((wx - c->xMin) * c->xToImage);
            iy = static_cast<int>((c->yMax - wy) * c->yToImage);
            inside = (static_cast<unsigned int>(ix) < static_cast<unsigned int>(X.W)) & (static_cast<unsigned int>(iy) < static_cast<unsigned int>(X.H));
          }
          oob += inside ? 0u : 1u;                              /* quirk Q5 */
          if(FULL)
            key[j] = inside ? pixel_key(0, iy, ix) : kNoPixel;
          else if(inside && iy >= stripRow0)
          {
            int strip;
            bool centre;
            if(ground_strip_of(ix, stripX0, strip, centre))
            {
              /* One of the few pixels the bottom scan can see.  It looks, per scan column, for the BOTTOM-most lit pixel of the
               * closed image, which lies at or below the bottom-most raw pixel of the column itself (closing only adds) and is a
               * function of the raw rows within two of it: a pixel more than two rows above a centre-column pixel already seen
               * in its strip cannot matter and stays unwritten (round 4: the blocks run from the bottom of the camera image up,
               * so after a frame's first blocks nearly nothing is written: 4.5 k -> a few hundred global atomics per frame). */
              const int seen = SSD_CHK(24, strip, kMaxGroundStrips) ? L.stripMax[strip] : 0x7fffffff;
              if(iy >= seen - 2)
              {
                if(SSD_CHK(23, static_cast<unsigned int>(iy) * (2u * X.W64) + (static_cast<unsigned int>(ix) >> 5), 2u * static_cast<unsigned int>(X.H) * X.W64))
                  atomicOr(gimg32 + (static_cast<unsigned int>(iy) * (2u * X.W64) + (static_cast<unsigned int>(ix) >> 5)), 1u << (ix & 31));
                gy0 = min(gy0, iy);
                gy1 = max(gy1, iy);
                if(centre && iy > seen)
                  atomicMax(&L.stripMax[strip], iy);
              }
            }
          }
        }
      }
    }
    if(FULL)
      wavewin_emit(ww, wm, win, gimg, imgWords, X.W64, X.winShiftGround, box, key, lane);
    g = gNext;
#pragma unroll
    for(int j = 0; j < kPts; j++)
      v[j] = vn[j];
  }
  if(FULL)
  {
    wavewin_flush(ww, win, gimg, imgWords, X.W64, X.winShiftGround, box, lane);
    wavemiss_flush(wm, box, lane);
  }
  else
  {
    /* the rows this wave wrote; all strips of those rows are k_final's to read and to clear (word columns: the whole row) */
    gy1 = wave_max_i(gy1);
    if(gy1 >= 0)
    {
      gy0 = wave_min_i(gy0);
      if(lane == 0)
      {
        atomicMin(&box[0].yMin, gy0); atomicMax(&box[0].yMax, gy1);
        atomicMin(&box[0].xMin, 0); atomicMax(&box[0].xMax, X.W64 - 1);
      }
    }
  }
  ph.mark(2);                                               /* the walk */
  flushAcc();
  if(oob)
    atomicAdd(&lOob, oob);
  __syncthreads();
  if(!FULL && tid < kMaxGroundStrips && L.stripMax[tid] > stripMaxMine)
    atomicMax(&fs.groundStripMax[tid], L.stripMax[tid]);       /* for the frame's blocks still to come */
  ph.mark(3);                                               /* waiting for the block's other waves */
  if(tid < nLive)
  {
    unsigned long long s = 0;
    unsigned int c = 0;
    for(int k = 0; k < 8; k++)
    {
      s += lsum[tid][k];
      c += lcnt[tid][k];
    }
    if(c)
    {
      const int acc = liveAcc[tid];
      atomicAdd(reinterpret_cast<unsigned long long *>(&fs.sumZ[acc]), s);
      atomicAdd(&fs.cnt[acc], c);
    }
  }
  if(tid == 0)
  {
    if(box[0].yMax >= 0)
    {
      atomicMin(&fs.imgYMin[kMaxStepImages], box[0].yMin); atomicMax(&fs.imgYMax[kMaxStepImages], box[0].yMax);
      atomicMin(&fs.imgXMin[kMaxStepImages], box[0].xMin); atomicMax(&fs.imgXMax[kMaxStepImages], box[0].xMax);
    }
    if(lOob)
    {
      atomicAdd(&fs.nOob, lOob);
      atomicOr(&fs.status, static_cast<unsigned int>(SSD_ST_OOB_PIXEL));
    }
  }
  ph.mark(4);                                               /* sums out */
  ph.finish();
}

template<int SRC, bool FULL, bool CHECKS>
__global__ __launch_bounds__(kThreads, FULL ? 4 : SSD_K4_WAVES) void k_inquad(const float *__restrict__ xyz, size_t strideFloats, PointParams P, PreXY Q,
                                                        PixelParams X, FrameState *__restrict__ st,
                                                        unsigned long long *__restrict__ groundImg,
                                                        const uint2 *__restrict__ tileMasks, size_t tileMaskStride, int chunkPoints, DepthSrc D)
{
  __shared__ InquadLds<FULL> L;
  /* The chunks at the bottom of the camera image first (an eighth of them, bottom-most first): the ground nearest to the camera,
   * whose pixels decide what the strip raster may leave out (see there); then the others top-down as ever.  (All chunks
   * bottom-up: XGA 0.79 -> 0.73 ms like this order, FHD stress 0.55 -> 0.59 — its eight treads, the heavy blocks, came last.) */
  const int nChunks = static_cast<int>(gridDim.y), first = max(1, nChunks / 8), by = static_cast<int>(blockIdx.y);
  const int chunkIdx = by < first ? nChunks - 1 - by : by - first;
  inquad_block<SRC, FULL, CHECKS>(L, xyz, strideFloats, P, Q, X, st, groundImg, tileMasks, tileMaskStride, chunkPoints, D, blockIdx.x, chunkIdx);
}

/* ========================================================================= */
/* K5: ground front edge and the per-frame result — one workgroup per frame     */

/* calcAverageZ (pointcloud.cpp:574-581): sum / points.size().  A quadrilateral that accepted no point makes that 0.0 / 0.0,
 * which on the reference's x86 is the DEFAULT NaN with the sign bit SET (0xfff8...): the line then reads "-nan"
 * (stairs.cpp:43 through operator<<).  This GPU's 0.0 / 0.0 is the positive quiet NaN — so the empty case is stated, not divided. */
__device__ __forceinline__ double mean_of_fixed(long long sumFixed, unsigned int n)
{
  if(n == 0u)
    return __longlong_as_double(static_cast<long long>(0xfff8000000000000ull));
  return (static_cast<double>(sumFixed) / static_cast<double>(1ll << kZFixShift)) / n;
}

struct FinalShared
{
  int yEdge[kMaxCols];
  int px[kMaxCols], py[kMaxCols];
  int n;
  LineI line;
  double partRes[kMaxImgWaves];             /* BestLine: every wave's best over its share of the pairs */
  int partT[kMaxImgWaves];
  LineI partLine[kMaxImgWaves];
  double stepsWorld[SSD_MAX_STEPS][9];       /* z, 4 x (x,y) in camera-dependent world coordinates; thread 0 only:
                                                in LDS because a private array would live in scratch memory */
};

template<int T>
__global__ __launch_bounds__(T) void k_final(Params P, FrameState *__restrict__ st,
                                                    unsigned long long *__restrict__ groundImg,
                                                    ssd_frame_result *__restrict__ results,
                                                    DebugFrame *__restrict__ dbg,
                                                    unsigned long long *__restrict__ dbgImg)
{
  __shared__ FinalShared S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frame = blockIdx.x;
  FrameState &fs = st[frame];
  ssd_frame_result &res = results[frame];
  const bool threw = (fs.status & SSD_ST_THROW) != 0;
  const bool haveGround = !threw && fs.firstValidInd >= 0 && fs.groundInd >= 0;

  const size_t imgWords = static_cast<size_t>(P.H) * P.W64;
  unsigned long long *img = groundImg + static_cast<size_t>(frame) * imgWords;
  const BitImg im{ img, P.W, P.H, P.W64 };

  /* bottom-scan columns: x_j = xr0 + 50 j, centre column j = jc */
  const int xStep = kGroundScanStep;
  const int xc = P.W / 2;
  const int xr0 = ground_scan_x0(P.W);
  const int jc = xc / xStep;
  const int nCols = min(kMaxCols, (P.W - 1 - xr0) / xStep + 1);
  SSD_PHASE(2, 0);

  for(int j = tid; j < kMaxCols; j += T)
    S.yEdge[j] = -1;
  if(tid == 0)
    S.n = 0;
  /* k_inquad's record of the strips' bottom-most centre pixels has served (this frame's strip raster is complete): left at
   * "none" for whichever k_inquad comes next, also one of a partial run that k_quads — its other resetter — does not precede */
  for(int j = tid; j < kMaxGroundStrips; j += T)
    fs.groundStripMax[j] = -1;
  __syncthreads();

  if(haveGround)
  {
    /* BottomScanner::probeBottomUp (segmentation.cpp:225-241): lowest lit pixel with y > H/2 */
    const int yStop = P.H / 2;
    unsigned long long *dbgRaw = nullptr, *dbgClosed = nullptr;
    if(dbgImg)
    {
      dbgRaw = dbgImg + ((static_cast<size_t>(frame) * (P.maxStepImages + 1) + P.maxStepImages) * 2) * imgWords;
      dbgClosed = dbgRaw + imgWords;
    }
    int by0 = fs.imgYMin[kMaxStepImages], by1 = fs.imgYMax[kMaxStepImages];
    int bc0 = fs.imgXMin[kMaxStepImages], bc1 = fs.imgXMax[kMaxStepImages];
    const bool emptyImg = by1 < by0;
    grow_box_to_border(P.W, P.H, P.W64, by0, by1, bc1);
    if(dbgImg)
    {
      by0 = 0; by1 = P.H - 1; bc0 = 0; bc1 = P.W64 - 1;
    }
    else if(by0 <= yStop)
      by0 = yStop + 1;                               /* only rows below the image centre are probed */
    const int bw = (emptyImg && !dbgImg) || by1 < by0 ? 0 : bc1 - bc0 + 1, bh = by1 - by0 + 1;
    if(dbgImg)
    {
      /* debug capture: the raw and the closed image whole, word by word */
      const int nBands = bw > 0 ? max(1, T / bw) : 0;
      const int bandRows = nBands > 0 ? (bh + nBands - 1) / nBands : 0;
      for(int t = tid; t < bw * nBands; t += T)
      {
        const int band = t / bw;
        const int c = bc0 + (t - band * bw);
        const int yA = by0 + band * bandRows, yB = min(yA + bandRows, by1 + 1);
        closed_column(im, c, yA, yB, true, [&](int y, unsigned long long cw)
        {
          dbgRaw[y * P.W64 + c] = img[y * P.W64 + c];
          dbgClosed[y * P.W64 + c] = cw;
        });
      }
    }
    /* the scan columns inside the box in bands of rows, as in k_outline: the last closed row below the image centre */
    if(bw > 0)
    {
      const int yTop = max(by0, yStop + 1);
      const int nRowsBelow = by1 - yTop + 1;
      const int xLo = 64 * bc0, xHi = min(64 * bc1 + 63, P.W - 1);
      const int jLo = xLo <= xr0 ? 0 : (xLo - xr0 + xStep - 1) / xStep;
      const int jHi = xHi < xr0 ? -1 : min(nCols - 1, (xHi - xr0) / xStep);
      const int nJ = nRowsBelow > 0 ? jHi - jLo + 1 : 0;
      const int nBands = nJ > 0 ? max(1, T / nJ) : 0;
      const int bandRows = nBands > 0 ? (nRowsBelow + nBands - 1) / nBands : 0;
      for(int t = tid; t < nJ * nBands; t += T)
      {
        const int band = t / nJ;
        const int j = jLo + (t - band * nJ);
        const int yA = yTop + band * bandRows, yB = min(yA + bandRows, by1 + 1);
        int yHi = -1;
        closed_scan_column(im, xr0 + xStep * j, yA, yB, [&](int y) { yHi = y; });
        if(yHi >= 0)
          atomicMax(&S.yEdge[j], yHi);
      }
    }
    __syncthreads();
    SSD_PHASE(2, 1);

    /* BottomScanner::scan (:170-222): first hit rightwards from the centre (else leftwards), then
     * contiguous hits to the right of it, then to the left of it */
    {
      /* every wave for itself: the columns' hit bits by ballot, the run around the first hit by bit scans */
      const bool h0 = lane < nCols && S.yEdge[lane] >= 0, h1 = lane + 64 < nCols && S.yEdge[lane + 64] >= 0;
      const unsigned long long hLo = __ballot(h0), hHi = __ballot(h1);
      auto hitAt = [&](int j) { return j >= 0 && j < nCols && (((j < 64 ? hLo : hHi) >> (j & 63)) & 1ull) != 0ull; };
      int jStart = jc;
      while(jStart < nCols && !hitAt(jStart))
        jStart++;
      if(jStart >= nCols)
      {
        jStart = jc - 1;
        while(jStart >= 0 && !hitAt(jStart))
          jStart--;
      }
      int nRight = 0, nLeft = 0;                 /* hits to the right of the start column, to the left of it */
      if(jStart >= 0)
      {
        while(hitAt(jStart + 1 + nRight))
          nRight++;
        while(hitAt(jStart - 1 - nLeft))
          nLeft++;
      }
      const int n = jStart >= 0 ? 1 + nRight + nLeft : 0;
      /* point k: the start column, then rightwards, then leftwards */
      for(int k = tid; k < n; k += T)
      {
        const int j = k <= nRight ? jStart + k : jStart - (k - nRight);
        S.px[k] = xr0 + xStep * j;
        S.py[k] = S.yEdge[j];
      }
      if(tid == 0)
        S.n = n;
    }
    __syncthreads();
    SSD_PHASE(2, 2);
    if(S.n >= 2)
    {
      /* BestLine: the pairs dealt out over the block's waves, the waves' bests combined in pair order */
      double res;
      int t;
      LineI l;
      wave_best_line_part(S.px, S.py, S.n, lane, 3ll * P.W * P.H < (1ll << 25), 64 * wave, T, res, t, l);
      if(lane == 0)
      {
        S.partRes[wave] = res;
        S.partT[wave] = t;
        S.partLine[wave] = l;
      }
    }
    __syncthreads();
    if(S.n >= 2 && tid == 0)
    {
      int best = -1;
      for(int w = 0; w < T / 64; w++)
        if(S.partT[w] != 0x7fffffff && (best < 0 || S.partRes[w] < S.partRes[best] || (S.partRes[w] == S.partRes[best] && S.partT[w] < S.partT[best])))
          best = w;
      S.line = best >= 0 ? S.partLine[best] : LineI{ 0, 0, 0 };
    }
    __syncthreads();
  }
  SSD_PHASE(2, 3);

  /* ---- the emitted surfaces, one lane each (wave 0): lane 0 the ground (calcGround, pointcloud.cpp:528-547), lane 1 + i the
   *      plateau firstValidInd + i (calcStairStep :549-558); every lane fetches its own sums and corners (one round trip
   *      for all), its place in the result is the count of emitted surfaces before it; then the detectStairs tail
   *      (:370-383): ToExternalWorld (transformation.cpp:190-194), straight from the lane's registers ---- */
  static_assert(kMaxPlateaus + 1 <= 64 && SSD_MAX_STEPS <= 64, "one lane per surface");
  if(wave == 0)
  {
    const bool any = !threw && fs.firstValidInd >= 0;
    const int firstValidInd = fs.firstValidInd, firstStep = fs.firstStep, last = fs.firstStep + fs.nStepImages;
    const bool groundLane = any && lane == 0 && fs.groundInd >= 0;
    const int k = firstValidInd + lane - 1;
    const bool stepLane = any && lane >= 1 && k < last && fs.pl[k < last ? max(k, 0) : 0].valid;
    double sW[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };         /* z, 4 x (x, y): camera-dependent world coordinates */
    if(groundLane)
    {
      /* "return {}" when no front edge is found (quirk Q6): the ground stays all zero */
      const bool valid = S.n >= 2;
      fs.groundFrontValid = valid ? 1 : 0;
      const double meanZ = mean_of_fixed(fs.sumZ[kGroundAcc], fs.cnt[kGroundAcc]);
      double fimg[4] = { 0, 0, 0, 0 };
      if(valid)
      {
        /* detectFrontEdge tail (segmentation.cpp:896-906) */
        const LineI l = S.line;
        const double fm = static_cast<double>(-l.a) / l.b, fn = static_cast<double>(-l.c) / l.b;
        int xl = S.px[0], xr = S.px[0];
        for(int i = 1; i < S.n; i++) { xl = min(xl, S.px[i]); xr = max(xr, S.px[i]); }
        fimg[0] = xl; fimg[1] = xl * fm + fn;
        fimg[2] = xr; fimg[3] = xr * fm + fn;
        const double flx = P.xMin + fimg[0] * P.xToWorld, fly = P.yMax - fimg[1] * P.yToWorld;
        const double frx = P.xMin + fimg[2] * P.xToWorld, fry = P.yMax - fimg[3] * P.yToWorld;
        const LineD frontLine = line_through_d(flx, fly, frx, fry);
        const double *g = fs.groundQuadWorld;
        const LineD leftSide = line_through_d(g[0], g[1], g[4], g[5]);
        const LineD rightSide = line_through_d(g[2], g[3], g[6], g[7]);
        /* StairsDetector::Line::intersection (pointcloud.cpp:520-525) */
        const double dl = frontLine.a * leftSide.b - leftSide.a * frontLine.b;
        const double dr = frontLine.a * rightSide.b - rightSide.a * frontLine.b;
        sW[0] = meanZ;
        sW[1] = (frontLine.b * leftSide.c - leftSide.b * frontLine.c) / dl;
        sW[2] = (leftSide.a * frontLine.c - frontLine.a * leftSide.c) / dl;
        sW[3] = (frontLine.b * rightSide.c - rightSide.b * frontLine.c) / dr;
        sW[4] = (rightSide.a * frontLine.c - frontLine.a * rightSide.c) / dr;
        sW[5] = g[4]; sW[6] = g[5];
        sW[7] = g[6]; sW[8] = g[7];
      }
      if(dbg)
      {
        ssd_debug_frame &d = dbg[frame].d;
        d.ground_front_valid = valid ? 1 : 0;
        d.ground_n_in_quad = static_cast<int>(fs.cnt[kGroundAcc]);
        d.ground_mean_z = meanZ;
        d.ground_n_pts = S.n;
        for(int i = 0; i < S.n; i++) { d.ground_pts[i][0] = S.px[i]; d.ground_pts[i][1] = S.py[i]; }
        if(valid)
        {
          d.ground_line[0] = S.line.a; d.ground_line[1] = S.line.b; d.ground_line[2] = S.line.c;
          for(int i = 0; i < 4; i++) d.ground_front_img[i] = fimg[i];
        }
      }
    }
    if(stepLane)
    {
      /* calcAverageZ over the plateau's points inside its quadrilateral = all of them (k_raster's sum, the histogram's
       * count) minus the ones outside (k_inquad's sum and count) */
      const long long inZ = fs.totZ[k - firstStep] - fs.sumZ[k];
      const unsigned int inN = static_cast<unsigned int>(fs.pl[k].nPoints) - fs.cnt[k];
      const double meanZ = mean_of_fixed(inZ, inN);
      sW[0] = meanZ;
#pragma unroll
      for(int c = 0; c < 8; c++)
        sW[1 + c] = fs.pl[k].quadWorld[c];
      if(dbg)
      {
        ssd_debug_plateau &p = dbg[frame].d.plateaus[k];
        p.n_in_quad = static_cast<int>(inN);
        p.sum_z_fix = inZ;
        p.mean_z = meanZ;
      }
    }
    const unsigned long long emitted = __ballot(groundLane || stepLane);
    const int place = __popcll(emitted & ((1ull << lane) - 1ull));
    const int n = min(__popcll(emitted), SSD_MAX_STEPS);
    if((groundLane || stepLane) && place < SSD_MAX_STEPS)
    {
      /* transformation.cpp:209-211; a NaN mean keeps its sign through the sum on x86 (the operand NaN is propagated) */
      res.steps[place].height = sW[0] != sW[0] ? sW[0] : P.worldZ + sW[0];
#pragma unroll
      for(int c = 0; c < 4; c++)
      {
        const double x = sW[1 + 2 * c], y = sW[2 + 2 * c];
        double ex = P.r2[0] * x + P.r2[1] * y;
        double ey = P.r2[2] * x + P.r2[3] * y;
        ex = ex + P.t2[0];
        ey = ey + P.t2[1];
        res.steps[place].quad[2 * c] = ex;
        res.steps[place].quad[2 * c + 1] = ey;
      }
#pragma unroll
      for(int c = 0; c < 9; c++)
        S.stepsWorld[place][c] = sW[c];              /* for the risers below */
    }
    if(lane >= n && lane < SSD_MAX_STEPS)            /* unused slots are zero, so results compare bytewise */
    {
      res.steps[lane].height = 0.0;
#pragma unroll
      for(int c = 0; c < 8; c++)
        res.steps[lane].quad[c] = 0.0;
    }
    if(lane == 0)
    {
      S.n = n;                                       /* from here on: the number of emitted surfaces */
      res.n_steps = n;
      res.status = static_cast<int>(fs.status);
      if(dbg)
      {
        dbg[frame].d.status = static_cast<int>(fs.status);
        dbg[frame].d.n_oob = static_cast<int>(fs.nOob);
      }
    }
  }
  SSD_PHASE(2, 4);
  __syncthreads();
  if(P.risers && tid == 0)
  {
    const int n = S.n;
    double (&stepsWorld)[SSD_MAX_STEPS][9] = S.stepsWorld;
    {
      /* vertical faces (extension, include/ssd_hip.h): one riser under the front edge of every emitted surface but
       * the lowest; its evidence is gathered by k_risers from the bins of no plateau between the two heights */
      for(int b = 0; b < kMaxBins; b++)
        fs.riserOfBin[b] = -1;
      unsigned int wanted = 0u;
      const int nR = n > 1 ? n - 1 : 0;
      for(int i = 0; i < nR; i++)
      {
        const double *lower = stepsWorld[i], *upper = stepsWorld[i + 1];
        RiserState &R = fs.riser[i];
        R.leftX = upper[1]; R.leftY = upper[2]; R.rightX = upper[3]; R.rightY = upper[4];
        R.zBottom = lower[0]; R.zTop = upper[0];
        const double dx = R.rightX - R.leftX, dy = R.rightY - R.leftY;
        const double len = sqrt(dx * dx + dy * dy);
        R.ox = R.leftX; R.oy = R.leftY;
        R.zLo = lower[0] + P.heightInterval;
        R.zHi = upper[0] - P.heightInterval;
        const bool usable = len > 0.0 && R.zLo < R.zHi;
        R.ux = usable ? dx / len : 0.0;
        R.uy = usable ? dy / len : 0.0;
        R.len = usable ? len : -1.0;                        /* no point has 0 <= t <= -1 */
        if(usable)
        {
          int bLo = static_cast<int>((R.zLo - P.zMin) * P.pt.recip), bHi = static_cast<int>((R.zHi - P.zMin) * P.pt.recip);
          bLo = max(0, min(bLo, P.nBins - 1));
          bHi = max(0, min(bHi, P.nBins - 1));
          for(int b = bLo; b <= bHi; b++)
            if(fs.lut[b] == 0xff && fs.riserOfBin[b] < 0)
            {
              fs.riserOfBin[b] = static_cast<signed char>(i);
              wanted |= 1u << (b / kBinsPerGroup);
            }
        }
      }
      for(int i = 0; i < kMaxRisers; i++)
      {
        fs.rSum[i] = 0;
        fs.rCnt[i] = 0u;
      }
      fs.nRisers = nR;
      fs.wantedRisers = wanted;
    }
  }
  __syncthreads();
  SSD_PHASE(2, 5);

  /* leave the ground image zeroed for the next batch (all its bits lie inside the bounding box) */
  if(fs.imgYMax[kMaxStepImages] >= fs.imgYMin[kMaxStepImages])
  {
    const int cy0 = fs.imgYMin[kMaxStepImages], cc0 = fs.imgXMin[kMaxStepImages];
    const int cw = fs.imgXMax[kMaxStepImages] - cc0 + 1, ch = fs.imgYMax[kMaxStepImages] - cy0 + 1;
    if(P.px.groundFull)
    {
      for(int idx = tid; idx < cw * ch; idx += T)
      {
        const int ry = idx / cw;
        const size_t o = static_cast<size_t>(cy0 + ry) * P.W64 + cc0 + (idx - ry * cw);
        if(SSD_CHK(22, o, static_cast<size_t>(P.H) * P.W64))
          img[o] = 0ull;                                 /* unconditionally: a load first would make every word a round trip */
      }
    }
    else
    {
      /* k_inquad set strip pixels only (ground_strip_column: within two of x_j for ANY integer j, also j = -1 and one past the
       * last scanned column): the one or two 32-bit words of every such strip, rows cy0 .. */
      unsigned int *img32 = reinterpret_cast<unsigned int *>(img);
      const int nStrips = (P.W - 1 + 2 - xr0) / xStep + 3;           /* j = -1 .. (W + 1 - xr0) / 50 */
      const int words32 = 2 * P.W64;
      for(int idx = tid; idx < ch * nStrips * 2; idx += T)
      {
        const int ry = idx / (nStrips * 2), rest = idx - ry * (nStrips * 2);
        const int x = xr0 + xStep * ((rest >> 1) - 1) + ((rest & 1) ? 2 : -2);
        const int w = x >> 5;                                          /* arithmetic shift: negative stays negative */
        if(x >= 0 && w < words32)
          img32[static_cast<size_t>(cy0 + ry) * words32 + w] = 0u;
      }
    }
  }
  SSD_PHASE(2, 6);
}

/* ========================================================================= */
/* K6 (extension): evidence of the vertical faces                              */

template<int SRC>
__global__ __launch_bounds__(kThreads, 8) void k_risers(const float *__restrict__ xyz, size_t strideFloats, PointParams P, double tol,
                                                        FrameState *__restrict__ st, const uint2 *__restrict__ tileMasks,
                                                        size_t tileMaskStride, int chunkPoints, int cellCols, DepthSrc D)
{
  __shared__ unsigned short cellList[kMaxCellsPerBlock];
  __shared__ unsigned int listScratch[2 * kWavesPerBlock];
  __shared__ RiserState rs[kMaxRisers];
  __shared__ signed char riserOfBin[kMaxBins];
  __shared__ unsigned long long lsum[kMaxRisers][8];
  __shared__ unsigned int lcnt[kMaxRisers][8];

  const int tid = threadIdx.x, lane = tid & 63;
  const int frame = blockIdx.x;
  FrameState &fs = st[frame];
  const unsigned int wanted = fs.wantedRisers;
  if(wanted == 0u)                                          /* block-uniform: set by k_final */
    return;
  const int nR = fs.nRisers;
  if(tid < kMaxBins)
    riserOfBin[tid] = fs.riserOfBin[tid];
  if(tid < nR)
    rs[tid] = fs.riser[tid];
  for(int i = tid; i < kMaxRisers * 8; i += kThreads)
  {
    (&lsum[0][0])[i] = 0ull;
    (&lcnt[0][0])[i] = 0u;
  }
  __syncthreads();

  const float *base = SRC == kSrcDepth16
    ? reinterpret_cast<const float *>(reinterpret_cast<const unsigned short *>(xyz) + static_cast<size_t>(frame) * strideFloats)
    : xyz + static_cast<size_t>(frame) * strideFloats;
  const int begin = blockIdx.y * chunkPoints;
  const int end = min(begin + chunkPoints, P.nPoints);
  const int copy = lane & 7;

  int curR = -1;
  long long accS = 0;
  unsigned int accN = 0;
  auto flushAcc = [&]()
  {
    if(curR >= 0 && accN)
    {
      atomicAdd(&lsum[curR][copy], static_cast<unsigned long long>(accS));
      atomicAdd(&lcnt[curR][copy], accN);
    }
  };

  /* only the cells that hold a bin of a riser are walked (as k_raster / k_inquad) */
  const int cell0 = begin / kCell;
  const int nCells = (end - begin + kCell - 1) / kCell;
  const int count = cell_list_build(tileMasks + static_cast<size_t>(frame) * tileMaskStride + cell0, nCells, cellCols,
                                    [&](const uint2 info) { return (info.x & wanted) != 0u; }, cellList, listScratch);
  const int nGroups = (count + 3) >> 2;
  const int gEnd = ((tid >> 6) + 1) * nGroups / kWavesPerBlock;
  for(int g = (tid >> 6) * nGroups / kWavesPerBlock; g < gEnd; g++)
  {
    F3 v[kPts];
    load_cell<SRC>(base, cell0, cellList, 4 * g + (lane >> 4), count, lane, P.nPoints, v, D);
    #pragma unroll
    for(int j = 0; j < kPts; j++)
    {
      double wx, wy, wz;
      if(!world_z(P, v[j], wz))
        continue;
      const int r = riserOfBin[height_bin(P, wz)];
      if(r < 0)
        continue;
      const RiserState &R = rs[r];
      if(!(wz > R.zLo && wz < R.zHi))
        continue;
      if(!world_xy(P, v[j], wx, wy))
        continue;
      const double a = wx - R.ox, b = wy - R.oy;
      const double sd = b * R.ux - a * R.uy;                 /* signed distance from the edge line */
      const double t = a * R.ux + b * R.uy;                  /* position along the edge */
      if(!(fabs(sd) <= tol && t >= 0.0 && t <= R.len))
        continue;
      if(r != curR)
      {
        flushAcc();
        curR = r; accS = 0; accN = 0;
      }
      accS += z_to_fixed(sd);
      accN++;
    }
  }
  flushAcc();
  __syncthreads();
  if(tid < nR)
  {
    unsigned long long sum = 0;
    unsigned int c = 0;
    for(int k = 0; k < 8; k++)
    {
      sum += lsum[tid][k];
      c += lcnt[tid][k];
    }
    if(c)
    {
      atomicAdd(reinterpret_cast<unsigned long long *>(&fs.rSum[tid]), sum);
      atomicAdd(&fs.rCnt[tid], c);
    }
  }
}

/* one thread per frame: the riser records in external world coordinates (ToExternalWorld as in k_final) */
__global__ void k_riser_results(Params P, const FrameState *__restrict__ st, ssd_frame_risers *__restrict__ out, int nframes)
{
  const int frame = blockIdx.x * blockDim.x + threadIdx.x;
  if(frame >= nframes)
    return;
  const FrameState &fs = st[frame];
  ssd_frame_risers &o = out[frame];
  const int nR = fs.nRisers;
  o.n_risers = nR;
  o.reserved = 0;
  for(int i = 0; i < kMaxRisers; i++)
  {
    ssd_riser &q = o.risers[i];
    if(i >= nR)
    {
      q.n_points = 0; q.detected = 0; q.height_bottom = 0.0; q.height_top = 0.0;
      q.left[0] = q.left[1] = q.right[0] = q.right[1] = 0.0; q.mean_offset = 0.0;
      continue;
    }
    const RiserState &R = fs.riser[i];
    const unsigned int c = fs.rCnt[i];
    q.n_points = static_cast<int>(c);
    q.detected = c >= static_cast<unsigned int>(P.riserMinSupport) ? 1 : 0;
    q.height_bottom = P.worldZ + R.zBottom;
    q.height_top = P.worldZ + R.zTop;
    double ex = P.r2[0] * R.leftX + P.r2[1] * R.leftY, ey = P.r2[2] * R.leftX + P.r2[3] * R.leftY;
    q.left[0] = ex + P.t2[0]; q.left[1] = ey + P.t2[1];
    ex = P.r2[0] * R.rightX + P.r2[1] * R.rightY; ey = P.r2[2] * R.rightX + P.r2[3] * R.rightY;
    q.right[0] = ex + P.t2[0]; q.right[1] = ey + P.t2[1];
    q.mean_offset = c ? (static_cast<double>(fs.rSum[i]) / static_cast<double>(1ll << kZFixShift)) / c : 0.0;
  }
}

} // namespace ssd

/* ========================================================================= */
/* launchers (declared in ssd_launch.h)                                        */
#include "ssd_launch.h"

namespace ssd
{

/* Grid layout of every per-frame kernel: FRAME on the fast axis (blockIdx.x), chunk / image slot on the slow
 * one.  Workgroups are dealt round-robin over the 8 XCDs in linear-id order; with the chunk (or the image
 * slot) on the fast axis every XCD would always get the same chunk positions of every frame — the stairs
 * for some XCDs, the skipped ground for others (measured: +30 % on K2, and K3 running on half of the XCDs
 * because only slots 0..3 hold images).  Frame-fastest gives each XCD whole frames: balanced, and a frame's
 * state and images stay in one XCD's L2. */
static inline int chunks_for(int nPoints, int chunkPoints) { return (nPoints + chunkPoints - 1) / chunkPoints; }

static inline bool aligned16(const float *xyz, size_t strideFloats, int nPoints)
{
  return (reinterpret_cast<uintptr_t>(xyz) & 15u) == 0 && (strideFloats & 3u) == 0 && (nPoints & 3) == 0;
}

void launch_predict(const float *xyz, size_t strideFloats, const Params &P, FrameState *st, int nframes, const DepthSrc *depth,
                    int *fallback, int poolPlanes, int sabotage, hipStream_t s)
{
  dim3 pgrid(nframes, kPredictParts);
  if(depth)
    hipLaunchKernelGGL(k_predict<kSrcDepth16>, pgrid, dim3(kThreads), 0, s, xyz, strideFloats, P.pt, st, *depth, P.minHeight, sabotage, fallback, poolPlanes);
  else if(aligned16(xyz, strideFloats, P.nPoints))
    hipLaunchKernelGGL(k_predict<kSrcF3Aligned>, pgrid, dim3(kThreads), 0, s, xyz, strideFloats, P.pt, st, DepthSrc{}, P.minHeight, sabotage, fallback, poolPlanes);
  else
    hipLaunchKernelGGL(k_predict<kSrcF3>, pgrid, dim3(kThreads), 0, s, xyz, strideFloats, P.pt, st, DepthSrc{}, P.minHeight, sabotage, fallback, poolPlanes);
}
void launch_hist(const float *xyz, size_t strideFloats, const Params &P, FrameState *st, uint2 *tileMasks, size_t tileMaskStride,
                 int nframes, int chunkPoints, const DepthSrc *depth, unsigned long long *planeImg, hipStream_t s)
{
  dim3 grid(nframes, chunks_for(P.nPoints, chunkPoints));
  /* the rare configurations' per-point tests (inputs beyond PreXY::maxInput that could read "inside", a z range that does not end on
   * a bin edge) live in instantiations of their own: as run-time flags they cost the common one two instructions per point */
  const bool checks = P.pre.checkInput != 0 || P.pre.zCheckTop != 0;
  if(planeImg)
  {
    /* a tile a whole number of camera rows: the tile loop keeps every wave in its band of columns; else the sorted strips */
    const bool strips = kTile % P.W != 0;
#define SSD_LAUNCH_PLANES2(SRC, STRIPS, DEPTH)                                                                                          \
    if(checks)                                                                                                                             \
      hipLaunchKernelGGL((k_hist_planes<SRC, STRIPS, true>), grid, dim3(kThreads), 0, s, xyz, strideFloats, P.pt, P.pre, P.px, st, tileMasks, planeImg, tileMaskStride, chunkPoints, DEPTH); \
    else                                                                                                                                   \
      hipLaunchKernelGGL((k_hist_planes<SRC, STRIPS, false>), grid, dim3(kThreads), 0, s, xyz, strideFloats, P.pt, P.pre, P.px, st, tileMasks, planeImg, tileMaskStride, chunkPoints, DEPTH);
#define SSD_LAUNCH_PLANES(SRC, DEPTH)                                                                                                    \
    if(strips)                                                                                                                             \
    {                                                                                                                                      \
      SSD_LAUNCH_PLANES2(SRC, true, DEPTH)                                                                                                 \
    }                                                                                                                                      \
    else                                                                                                                                   \
    {                                                                                                                                      \
      SSD_LAUNCH_PLANES2(SRC, false, DEPTH)                                                                                                \
    }
    if(depth)
    {
      SSD_LAUNCH_PLANES(kSrcDepth16, *depth)
    }
    else if(aligned16(xyz, strideFloats, P.nPoints))
    {
      SSD_LAUNCH_PLANES(kSrcF3Aligned, DepthSrc{})
    }
    else
    {
      SSD_LAUNCH_PLANES(kSrcF3, DepthSrc{})
    }
#undef SSD_LAUNCH_PLANES2
#undef SSD_LAUNCH_PLANES
    return;
  }
#define SSD_LAUNCH_HIST(SRC, DEPTH)                                                                                                      \
  if(checks)                                                                                                                               \
    hipLaunchKernelGGL((k_hist<SRC, true>), grid, dim3(kThreads), 0, s, xyz, strideFloats, P.pt, P.pre, st, tileMasks, tileMaskStride, chunkPoints, DEPTH); \
  else                                                                                                                                     \
    hipLaunchKernelGGL((k_hist<SRC, false>), grid, dim3(kThreads), 0, s, xyz, strideFloats, P.pt, P.pre, st, tileMasks, tileMaskStride, chunkPoints, DEPTH);
  if(depth)
  {
    SSD_LAUNCH_HIST(kSrcDepth16, *depth)
  }
  else if(aligned16(xyz, strideFloats, P.nPoints))
  {
    SSD_LAUNCH_HIST(kSrcF3Aligned, DepthSrc{})
  }
  else
  {
    SSD_LAUNCH_HIST(kSrcF3, DepthSrc{})
  }
#undef SSD_LAUNCH_HIST
}
void launch_peaks(const Params &P, FrameState *st, int nframes, DebugFrame *dbg, int *fallback, hipStream_t s)
{
  hipLaunchKernelGGL(k_peaks, dim3(nframes), dim3(64), 0, s, P, st, nframes, dbg, fallback ? 1 : 0, fallback);
}
void launch_raster(const float *xyz, size_t strideFloats, const Params &P, FrameState *st, unsigned long long *stepImg,
                   const uint2 *tileMasks, size_t tileMaskStride, int nframes, int chunkPoints, const DepthSrc *depth, const int *fallback, hipStream_t s)
{
  dim3 grid(fallback ? (nframes + 3) / 4 : nframes, chunks_for(P.nPoints, chunkPoints));
  const DepthSrc D = depth ? *depth : DepthSrc{};
#define SSD_LAUNCH_RASTER(SRC)                                                                                                                                           \
  {                                                                                                                                                                      \
    if(fallback)                                                                                                                                                         \
      hipLaunchKernelGGL((k_raster<SRC, true>), grid, dim3(kThreads), 0, s, xyz, strideFloats, P.pt, P.px, st, stepImg, tileMasks, tileMaskStride, chunkPoints, D, fallback); \
    else                                                                                                                                                                 \
      hipLaunchKernelGGL((k_raster<SRC, false>), grid, dim3(kThreads), 0, s, xyz, strideFloats, P.pt, P.px, st, stepImg, tileMasks, tileMaskStride, chunkPoints, D, fallback); \
  }
  if(depth)
    SSD_LAUNCH_RASTER(kSrcDepth16)
  else if(aligned16(xyz, strideFloats, P.nPoints))
    SSD_LAUNCH_RASTER(kSrcF3Aligned)
  else
    SSD_LAUNCH_RASTER(kSrcF3)
#undef SSD_LAUNCH_RASTER
}
void launch_outline(const Params &P, FrameState *st, unsigned long long *stepImg, unsigned long long *planeImg, int nframes, DebugFrame *dbg, unsigned long long *dbgImg, hipStream_t s)
{
  dim3 grid(nframes, P.maxStepImages);
  /* while every image has a CU of its own, the block that gets through its phases soonest; beyond that the cheapest */
  if(nframes <= kImgFewFrames)
    hipLaunchKernelGGL(k_outline<kImgThreadsFew>, grid, dim3(kImgThreadsFew), 0, s, P, st, stepImg, planeImg, dbg, dbgImg);
  else
    hipLaunchKernelGGL(k_outline<kImgThreadsBatch>, grid, dim3(kImgThreadsBatch), 0, s, P, st, stepImg, planeImg, dbg, dbgImg);
}
void launch_quads(const Params &P, FrameState *st, int nframes, DebugFrame *dbg, hipStream_t s)
{
  hipLaunchKernelGGL(k_quads, dim3(nframes), dim3(64), 0, s, P, st, nframes, dbg);
}
void launch_inquad(const float *xyz, size_t strideFloats, const Params &P, FrameState *st, unsigned long long *groundImg,
                   const uint2 *tileMasks, size_t tileMaskStride, int nframes, int chunkPoints, const DepthSrc *depth, hipStream_t s)
{
  dim3 grid(nframes, chunks_for(P.nPoints, chunkPoints));
  const int src = depth ? kSrcDepth16 : aligned16(xyz, strideFloats, P.nPoints) ? kSrcF3Aligned : kSrcF3;
  const DepthSrc D = depth ? *depth : DepthSrc{};
#define SSD_LAUNCH_INQUAD(SRC, FULL, CHECKS) \
  hipLaunchKernelGGL((k_inquad<SRC, FULL, CHECKS>), grid, dim3(kThreads), 0, s, xyz, strideFloats, P.pt, P.pre, P.px, st, groundImg, tileMasks, tileMaskStride, chunkPoints, D)
  /* the instantiation with the rare configurations' per-point tests (a magnitude test of the input, a z range that does not end on a bin
   * edge: ssd_prexy.h) only where the configuration needs them; debug capture (the whole ground image) always takes it */
  const bool checks = P.pre.checkInput != 0 || P.pre.zCheckTop != 0;
  if(P.px.groundFull)
  {
    if(src == kSrcDepth16) SSD_LAUNCH_INQUAD(kSrcDepth16, true, true);
    else if(src == kSrcF3Aligned) SSD_LAUNCH_INQUAD(kSrcF3Aligned, true, true);
    else SSD_LAUNCH_INQUAD(kSrcF3, true, true);
  }
  else if(checks)
  {
    if(src == kSrcDepth16) SSD_LAUNCH_INQUAD(kSrcDepth16, false, true);
    else if(src == kSrcF3Aligned) SSD_LAUNCH_INQUAD(kSrcF3Aligned, false, true);
    else SSD_LAUNCH_INQUAD(kSrcF3, false, true);
  }
  else
  {
    if(src == kSrcDepth16) SSD_LAUNCH_INQUAD(kSrcDepth16, false, false);
    else if(src == kSrcF3Aligned) SSD_LAUNCH_INQUAD(kSrcF3Aligned, false, false);
    else SSD_LAUNCH_INQUAD(kSrcF3, false, false);
  }
#undef SSD_LAUNCH_INQUAD
}
void launch_final(const Params &P, FrameState *st, unsigned long long *groundImg, ssd_frame_result *results, int nframes, DebugFrame *dbg, unsigned long long *dbgImg, hipStream_t s)
{
  if(nframes <= kImgFewFrames)
    hipLaunchKernelGGL(k_final<kImgThreadsFew>, dim3(nframes), dim3(kImgThreadsFew), 0, s, P, st, groundImg, results, dbg, dbgImg);
  else
    hipLaunchKernelGGL(k_final<kImgThreadsBatch>, dim3(nframes), dim3(kImgThreadsBatch), 0, s, P, st, groundImg, results, dbg, dbgImg);
}
void launch_risers(const float *xyz, size_t strideFloats, const Params &P, FrameState *st, const uint2 *tileMasks, size_t tileMaskStride,
                   ssd_frame_risers *out, int nframes, int chunkPoints, const DepthSrc *depth, hipStream_t s)
{
  dim3 grid(nframes, chunks_for(P.nPoints, chunkPoints));
  if(depth)
    hipLaunchKernelGGL(k_risers<kSrcDepth16>, grid, dim3(kThreads), 0, s, xyz, strideFloats, P.pt, P.riserTol, st, tileMasks, tileMaskStride, chunkPoints, P.px.cellCols, *depth);
  else if(aligned16(xyz, strideFloats, P.nPoints))
    hipLaunchKernelGGL(k_risers<kSrcF3Aligned>, grid, dim3(kThreads), 0, s, xyz, strideFloats, P.pt, P.riserTol, st, tileMasks, tileMaskStride, chunkPoints, P.px.cellCols, DepthSrc{});
  else
    hipLaunchKernelGGL(k_risers<kSrcF3>, grid, dim3(kThreads), 0, s, xyz, strideFloats, P.pt, P.riserTol, st, tileMasks, tileMaskStride, chunkPoints, P.px.cellCols, DepthSrc{});
  hipLaunchKernelGGL(k_riser_results, dim3((nframes + 63) / 64), dim3(64), 0, s, P, st, out, nframes);
}

} // namespace ssd

#ifdef SSD_COUNT
/* tools only: reads and clears K1's counters (tiles, tiles with candidates, window moves, words flushed, pixels, pixels that missed, candidate points) */
extern "C" int ssd_tools_k1_counters(unsigned long long *out)
{
  unsigned long long zero[8] = { 0 };
  if(hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(out, HIP_SYMBOL(ssd::ssd_k1_counters), sizeof(zero)) != hipSuccess)
    return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(ssd::ssd_k1_counters), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

#include "ssd_phase_readers.h"
