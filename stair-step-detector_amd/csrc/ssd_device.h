/*
 * ssd_device.h — device-side data layout shared by the kernels and the C ABI.
 *
 * HBM layout per handle (sized for max_frames_per_batch = F frames, S = max_step_plateaus):
 *   FrameState  state[F]            per-frame histogram, plateau table, quad tests, accumulators, result
 *   uint32_t    step_img[F][S][H][W32]   raw top-down bit images of the step plateaus (1 bit / pixel)
 *   uint32_t    ground_img[F][H][W32]    raw bit image of the ground points inside the ground quadrilateral
 *   uint32_t    plane_img[plane_pool_size(F, W H)][H][W32]   single-pass batches only: per HEIGHT BIN images of the bins k_predict expects
 *                                   step plateaus in, rastered by k_hist itself; k_outline merges a plateau's (two) planes
 *                                   into its step image (section "single pass" of DESIGN.md)
 *   DebugFrame  debug[F]            only when debug capture is on
 * W32 = ceil(W/64)*2 32-bit words per image row (rows are padded to whole 64-bit words).
 * Invariant: all bit images are zero between batches (the kernel that consumes an image clears it).
 */
#ifndef SSD_DEVICE_H_
#define SSD_DEVICE_H_

#include "../../include/ssd_hip.h"
#include <hip/hip_runtime.h>

namespace ssd
{

constexpr int kMaxBins = SSD_MAX_BINS;
constexpr int kMaxPlateaus = SSD_MAX_PLATEAUS;
constexpr int kMaxStepImages = SSD_MAX_STEP_IMAGES;
constexpr int kGroundAcc = kMaxPlateaus;          /* accumulator slot of the ground quadrilateral */
constexpr int kMaxRisers = SSD_MAX_RISERS;
constexpr int kMaxLive = kMaxStepImages + 1;      /* quadrilaterals a frame can have points tested against: one per step image + the ground */
constexpr int kZFixShift = 40;                    /* mean z accumulates round(z * 2^40) in int64 */
/* single pass: planes (one bit image per predicted height bin) a frame can have; beyond, the frame is rastered by k_raster */
constexpr int kMaxPlanes = SSD_MAX_PLANES;
/* k_predict samples one cell (64 consecutive points) of every kSpecSample */
#ifndef SSD_SPEC_SAMPLE
#define SSD_SPEC_SAMPLE 16
#endif
constexpr int kSpecSample = SSD_SPEC_SAMPLE;
/* the batch's list for k_raster (single pass): [0] = frames listed, [1] = frames without step plateaus, [2] = planes drawn from the
 * batch's pool so far (k_predict adds, k_peaks of the same batch leaves it zero), the frame indices from [kFallbackList] */
constexpr int kFallbackList = 3;
/* Planes come from a pool per workspace (round 5): a frame needs 6 (XGA bench scenes) to 9 (FHD stress) of the kMaxPlanes it may
 * have - 15 at VGA, where the sample is a sixth of XGA's and more peaks have neighbours too alike to call (three planes each) -,
 * so the pool holds kPoolPlanesPerFrame planes per frame of the largest batch (kPoolPlanesPerSmallFrame below
 * kPoolSmallFramePoints points, where a plane is small), + kPoolPlanesExtra, and never fewer than what a few frames could ask
 * for in full; k_predict draws a frame's planes with one atomic add, and a frame the pool cannot serve gets none - k_raster
 * does it, as any frame the predictor does not cover. */
constexpr int kPoolPlanesPerFrame = SSD_POOL_PLANES_PER_FRAME, kPoolPlanesPerSmallFrame = 16, kPoolSmallFramePoints = 600000, kPoolPlanesExtra = 2 * kMaxPlanes;
__host__ __device__ constexpr long long plane_pool_size(long long frames, long long pointsPerFrame)
{
  const long long per = pointsPerFrame < kPoolSmallFramePoints ? kPoolPlanesPerSmallFrame : kPoolPlanesPerFrame;
  const long long byShare = frames * per + kPoolPlanesExtra, few = (frames < 8 ? frames : 8) * kMaxPlanes;
  return byShare > few ? byShare : few;
}
constexpr int kMaxGroundStrips = 168;              /* pixel strips of the ground image the bottom scan reads: one per 50 columns (+ 2), width <= 8192 */

/* the four horizontal edges of a plateau outline, in this order everywhere (segmentation.cpp:585-589) */
enum { kFL = 0, kFR = 1, kBL = 2, kBR = 3 };

/* Kernel arguments are passed by value and live in SGPRs; a 256-thread block is admitted 8x per CU only
 * while the kernel needs <= 80 SGPRs (MI355X_MICROARCH.md, residency), so the streaming kernels get just
 * the constants they use: PointParams (44 SGPRs) and PixelParams, not the whole Params. */
struct PointParams
{
  double a[9], b[3];                          /* CameraToWorld: w = a * x + b (transformation.h:59-64) */
  double xMin, xMax, yMin, yMax, zMin, zMax;  /* Configuration::MeasuringRange (configuration.h:40-46) */
  double recip;                               /* 1 / heightInterval */
  double boxX, boxY;                          /* 256 / (xMax - xMin), 256 / (yMax - yMin): grid of the cells' bounding boxes */
  int nPoints, nBins;
};

/* K1's single-precision pre-filter of the x / y range test (round 5).  The reference decides "x and y in range" on doubles
 * (pointcloud.cpp:150-165); K1 first evaluates d = ((x, y) - centre of the range) / (extent of the range) with three packed
 * single-precision FMAs, whose distance from the exact value is bounded by `e` for inputs of magnitude <= maxInput: a point
 * with max(|dx|, |dy|) < lo = 0.5 - e is inside whatever the doubles say, one with max(..) > hi = 0.5 + e outside; only the
 * band between them (and inputs beyond maxInput, NaNs) takes the double-precision rows.  d also gives the cells' bounding
 * boxes (grid cell = (d + 0.5 -+ e) * 256).  make_pre_xy() derives the constants and the bound (ssd_capi.hip). */
struct PreXY
{
  float c[4][2];                              /* (d.x, d.y) = c[0] * x + c[1] * y + c[2] * z + c[3], pairs (x row, y row): one packed FMA per input */
  float lo, hi;
  float maxInput;                             /* the bound behind lo / hi holds for inputs up to this magnitude ... */
  float boxLo, boxHi;                         /* the cell of a minimum is rn(d * 256 + boxLo), of a maximum rn(d * 256 + boxHi), saturated to 0 .. 255 */
  int checkInput;                             /* ... and K1 tests it per point (1) unless make_pre_xy() could show that a larger input cannot be called "inside" (0) */
  /* Round 6: the z row in single precision first as well (make_pre_z(), ssd_prexy.h).  t = zc[0] x + zc[1] y + zc[2] z + zc[3] is the
   * point's height above zMin in histogram bins; its distance from the exact value is bounded by e(M) = zK M + zE0 with
   * M = max(|x|, |y|, |z|) - the bound follows the input's magnitude -, so a point whose t lies farther than e(M) from every
   * integer has the reference's bin floor(t) and the reference's z-range decision (0 < t < zTop on the bits); the others, NaNs
   * and overflows take the reference's doubles.  sure  <=>  |fract(t) - 1/2| < zH0 + zNegK * M. */
  float zc[4];
  float zNegK, zH0;                           /* -zK rounded away from zero, 1/2 - zE0 rounded down */
  unsigned int zTopBits;                      /* bits of zTop: the sure point is in the z range iff bits(t) < zTopBits (t >= +0 and t < zTop) */
  float zTop;                                 /* (zMax - zMin) / heightInterval: the integer next to it (zCheckTop == 0) or itself rounded (1) */
  int zCheckTop;                              /* the z range does not end within 2^-20 of a bin edge: |t - zTop| <= e(M) is unsure too (two more instructions per point) */
  /* the bound of d itself, following the point's magnitude M: |d - D| <= dK * M + dE0 in either coordinate, also for a d that came
   * from the doubles (rounded once) - what k_inquad's single-precision edge tests add to their own margin (QuadEdgesF) */
  float dK, dE0;
};

struct PixelParams
{
  double xToImage, yToImage;                  /* Projection2D (pointcloud.cpp:73-74) */
  int W, H, W64;                              /* W64 = 64-bit words per image row */
  int maxStepImages;
  int winShift;                               /* the waves' LDS image windows are (256 >> winShift) rows x (1 << winShift) words */
  int winShiftGround;                         /* the same for the ground image of k_inquad */
  int cellCols;                               /* cells (64 consecutive points) per camera row, rounded: cells this far apart are
                                                 vertical neighbours in the camera image (order of the kernels' cell lists) */
  int groundFull;                             /* k_inquad rasters every ground pixel (debug capture: the whole image is compared);
                                                 0 = only the pixels k_final's bottom scan can see (ground_strip_pixel) */
  /* round 6, K1's candidates: the pixel from the single-precision d of the range test (make_pre_pixel(), ssd_prexy.h):
   * px = d.x * fW + fHalfW, py = d.y * fNegH + fHalfH; certain iff max(|fract(px) - 1/2|, |fract(py) - 1/2|) < pxH0 + pxNegK * max(|x|, |y|, |z|) */
  float fW, fHalfW, fNegH, fHalfH;
  float pxNegK, pxH0;
};

/* what the 16-bit depth source needs beside the depth image: rs2::pointcloud's pre-computed maps
 * x = (u - ppx) / fx, y = (v - ppy) / fy (L515: no distortion, so they depend on one coordinate each) */
struct DepthSrc
{
  const float *xmap, *ymap;
  float depthUnits;
  int W, H;
  /* row of a point index without an integer division (about twenty instructions on this ISA): row = mulhi(idx, rowMagic) >> 7
   * with rowMagic = ceil(2^39 / W) — exact for idx * W < 2^39 (idx < W * H <= 2^26, W <= 2^13), fits 32 bits for W > 128;
   * 0 = divide (small images) */
  unsigned int rowMagic;
};

/* host + device: the magic of DepthSrc::rowMagic, or 0 when the shortcut does not apply */
__host__ __device__ inline unsigned int depth_row_magic(int W, int H)
{
  if(W <= 128 || W > 8192 || static_cast<long long>(W) * H > (1ll << 26))
    return 0u;
  return static_cast<unsigned int>(((1ull << 39) + static_cast<unsigned long long>(W) - 1ull) / static_cast<unsigned long long>(W));
}

/* all constants of one handle */
struct Params
{
  PointParams pt;
  PixelParams px;
  PreXY pre;
  int W, H, W64;
  int nPoints;
  double r2[4], t2[2], worldZ;
  double xMin, xMax, yMin, yMax, zMin, zMax;
  double xToImage, yToImage, xToWorld, yToWorld, xyRatio;
  int nBins, minHeight, minImgYExtent;
  int maxStepImages;
  /* vertical faces (extension, ssd_set_risers) */
  int risers, riserMinSupport;
  double riserTol, heightInterval;
};

/* strict point-in-quadrilateral test prepared once per quadrilateral
 * (reference quadrilateralTest.cpp:275-451: 3x3 cell map, <= 2 segments tested per cell) */
struct QuadTest
{
  /* the largest cell of the map that holds no segment and counts as inside (the middle of a tread):
   * fx0 <= x < fx1 && fy0 <= y < fy1 decides most points with four compares; empty when fx0 > fx1 */
  double fx0, fx1, fy0, fy1;
  double bxLo, bxUp, byLo, byUp;
  double segK[4], segC[4];
  double yTrans[2];
  double xTrans[3][2];
  unsigned char segSteep[4], segLeftIfPositive[4];
  unsigned char cellMask[3][3];    /* bit s set: segment s decides this cell */
  unsigned char cellConst[3][3];   /* result of a cell without segments */
  unsigned char nCells[3];
  unsigned char nRows, insideIsLeft;
  int err;                         /* 0 or the negative code of the reference's throw */
};

/* A live quadrilateral's four edges as half-planes on the grid of K1's cell boxes (k_quads builds them, k_inquad's cell
 * classification uses them): a box [x0, x1] x [y0, y1] of grid cells lies wholly inside the quadrilateral iff, for every
 * edge s, g[s][0] * xs + g[s][1] * ys + g[s][2] > 0 at the box corner (xs, ys) that minimises it (xs = x0 when
 * g[s][0] > 0, else x1 + 1; likewise ys).  The margin of the boxes' truncation is folded into g[s][2].
 * ok = 0: the reference's cell map does not agree with the geometry somewhere (see build_grid_segs) - do not use. */
struct QuadGridSegs
{
  double g[4][3];
  int ok, pad;
};

/* A live quadrilateral's four edges for k_inquad's per-POINT test in single precision (round 6; ssd_quadtest.h,
 * build_quad_edges): on the centred, normalised coordinates d of the range test (PreXY), e_s = gx[s] d.x + gy[s] d.y + g2[s]
 * with |gx[s]| + |gy[s]| = 1, positive inside.  A point whose smallest e_s exceeds m + (the bound of d for its magnitude) is
 * inside by the reference's test, one whose smallest e_s lies below minus that is outside; the band between takes the
 * reference's doubles.  m = +infinity: single precision says nothing about this quadrilateral (every point takes the doubles). */
struct alignas(16) QuadEdgesF
{
  float gx[4], gy[4], g2[4];
  float m, pad[3];
};

struct PlateauState
{
  int peakBin, binLo, binHi;       /* Plateau::height, chosen pair */
  int effLo, effHi;                /* bins that feed this plateau */
  int nPoints;
  int isStep, outlineFound, valid;
  double quadImg[8];
  double quadWorld[8];
};

/* one riser (extension): the front edge of the upper surface as origin + unit direction + length, and the open
 * height interval of its evidence */
struct RiserState
{
  double ox, oy, ux, uy, len, zLo, zHi;
  double leftX, leftY, rightX, rightY, zBottom, zTop;   /* camera-dependent world, for the result */
};

struct FrameState
{
  /* K1's accumulators.  Zero between calls: k_peaks takes them over into hist / nNonZero below and clears them, so a call
   * needs no memset of the state in front of it (ssd_capi.hip, stateClean) */
  unsigned int histAcc[kMaxBins];
  unsigned int nNonZeroAcc;
  unsigned int hist[kMaxBins];     /* the frame's height histogram (as k_peaks took it) */
  unsigned int nNonZero, nInRange, nOob, status;
  int nPlateaus, firstStep, nStepImages, groundInd, firstValidInd;
  int groundFrontValid;
  unsigned char lut[kMaxBins];     /* bin -> plateau index, 0xff = none */
  /* bounding box of the raw bits of each step image ([kMaxStepImages] = ground image): rows and 64-bit
   * word columns, written by the rasterising kernels, read by the kernels that close and scan the image */
  int imgYMin[kMaxStepImages + 1], imgYMax[kMaxStepImages + 1];
  int imgXMin[kMaxStepImages + 1], imgXMax[kMaxStepImages + 1];
  PlateauState pl[kMaxPlateaus];
  double groundQuadWorld[8];
  /* the live quadrilateral tests, compact (k_quads): slot k belongs to accumulator liveAcc[k] (treads ascending, the
   * ground last); lutLive: height bin -> slot; liveGroups: the groups of 4 height bins of the slot's plateau */
  QuadTest qtLive[kMaxLive];
  QuadGridSegs segLive[kMaxLive];
  QuadEdgesF edgeLive[kMaxLive];   /* the same edges for k_inquad's per-point test in single precision (build_quad_edges) */
  unsigned char liveAcc[kMaxLive];
  int4 liveBox[kMaxLive];          /* thresholds of k_inquad's cell classification on K1's box grid (k_quads: live_box_thresholds) */
  unsigned char lutLive[kMaxBins];
  unsigned int liveGroups[kMaxLive];
  int nLive;
  unsigned char accActive[kMaxPlateaus + 1];
  /* gates of K2 / K4 against K1's tile masks: groups of 8 height bins that hold a bin of a step plateau /
   * of a live quadrilateral; anyActive = some accumulator is live */
  unsigned int wantedSteps, wantedQuads, anyActive;
  /* fixed-point z sums and counts: [kGroundAcc] = the ground points INSIDE the ground quadrilateral; [k] of a tread =
   * its points OUTSIDE its quadrilateral (k_inquad); totZ[slot] = all points of the tread in image slot `slot` (k_raster) */
  long long sumZ[kMaxPlateaus + 1];
  unsigned int cnt[kMaxPlateaus + 1];
  long long totZ[kMaxStepImages];
  /* k_inquad's strip raster: per strip, the bottom-most row (largest) in which a ground point fell on the strip's CENTRE column
   * so far (-1: none; k_quads resets it) - pixels more than two rows above it cannot reach k_final's bottom scan */
  int groundStripMax[kMaxGroundStrips];
  /* Single pass (k_predict -> k_hist -> k_peaks -> k_outline).  k_predict: histogram of a sample of the frame's cells
   * (predHist / predDone are its accumulators, left zero), the bins that may belong to a step plateau, a plane for each:
   * specPlane[bin] (0xff: none).  k_hist rasters the points of those bins into their planes as it counts them: per plane the
   * bounding box of the bits, the sum of round(z * 2^40) and the points outside the image (quirk Q5).  k_peaks, with the
   * exact histogram: every bin of every step plateau has a plane (or no points) -> specOk, the step images' boxes / sums /
   * planes (imgPlane) from the planes'; otherwise the frame goes through k_raster as before.  k_outline merges and clears
   * the planes of its image and clears the planes no image uses (planeUsed == 0). */
  unsigned int predHist[kMaxBins];
  unsigned int predDone;
  unsigned char specPlane[kMaxBins];
  unsigned int predSample[kMaxBins];   /* the sample histogram the table was made of (kept for the tests: ssd_predict.h) */
  int nPlanes;
  int planeBase;                   /* the frame's first plane in the workspace's pool (k_predict; meaningless while nPlanes == 0) */
  int specOk;                      /* every step plateau of the frame covered */
  unsigned int slotCovered;        /* bit s: step image s is made of planes (k_peaks); the others are k_raster's */
  int planeYMin[kMaxPlanes], planeYMax[kMaxPlanes], planeXMin[kMaxPlanes], planeXMax[kMaxPlanes];
  long long planeTotZ[kMaxPlanes];
  unsigned int planeOob[kMaxPlanes];
  unsigned char planeUsed[kMaxPlanes];
  unsigned char imgPlane[kMaxStepImages][2];
  /* vertical faces (extension): written by k_final, accumulated by k_risers */
  int nRisers;
  unsigned int wantedRisers;
  signed char riserOfBin[kMaxBins];      /* bin -> riser whose height interval it overlaps, -1 = none */
  RiserState riser[kMaxRisers];
  long long rSum[kMaxRisers];
  unsigned int rCnt[kMaxRisers];
};

struct DebugFrame
{
  ssd_debug_frame d;
};

} // namespace ssd

#endif /* SSD_DEVICE_H_ */
