/* ssd_launch.h — host-callable launchers of the kernels in ssd_kernels.hip */
#ifndef SSD_LAUNCH_H_
#define SSD_LAUNCH_H_

#include "ssd_device.h"

namespace ssd
{
constexpr int kTileHost = 1024;
/* kTileHost = kThreads * kPts of ssd_kernels.hip: chunk sizes are multiples of it; a block's chunk is at most
 * kMaxTilesPerBlockHost tiles (K1 keeps the chunk's cell masks, K2/K4/K6 the list of its wanted cells, in LDS) */
#ifndef SSD_MAX_TILES
#define SSD_MAX_TILES 32
#endif
constexpr int kMaxTilesPerBlockHost = SSD_MAX_TILES;
constexpr int kMaxTilesPerBlockRasterHost = 128;   /* K2 only */
constexpr int kMaxTilesPerBlockInquadHost = 128;   /* K4 only */
constexpr int kCellHost = 64;               /* points per cell (one gating mask each) */
/* The single pass (k_hist rasters the step plateaus itself) pays a kernel (k_predict) that a few frames do
 * not earn back; its keys hold 13 bits of row and its windows must fit the image. */
/* from 64 XGA frames per batch on (tools/sp_frames.py: 48 frames 3 % slower, 64 frames 4 % faster; VGA between 128 and 192
 * frames, FHD around 24-64): by points */
constexpr long long kSinglePassMinPoints = 64ll * 1024 * 768;
inline bool single_pass_batch(int nframes, int nPoints) { return static_cast<long long>(nframes) * nPoints >= kSinglePassMinPoints; }
constexpr int kSinglePassBackoff = 63;      /* batches run in two passes after one the single pass could not pay on (ssd_fetch_back) */
constexpr int kPredictParts = 4;            /* blocks of k_predict per frame */
inline bool single_pass_geometry(int W, int H) { return W >= 64 && W <= 8192 && H >= 16 && H <= 4096; }

/* The single pass: launch_predict (k_predict; fallback = the batch's list of frames for k_raster - count, then indices: k_predict
 * resets it, k_peaks appends, k_raster reads; sabotage: see k_predict), then launch_hist with planeImg != nullptr (K1 rastering the
 * candidate bins' planes), launch_peaks / launch_raster with the list, launch_outline with the planes. */
void launch_predict(const float *xyz, size_t strideFloats, const Params &P, FrameState *st, int nframes, const DepthSrc *depth,
                    int *fallback, int poolPlanes, int sabotage, hipStream_t s);
void launch_hist(const float *xyz, size_t strideFloats, const Params &P, FrameState *st, uint2 *tileMasks, size_t tileMaskStride,
                 int nframes, int chunkPoints, const DepthSrc *depth, unsigned long long *planeImg, hipStream_t s);
void launch_peaks(const Params &P, FrameState *st, int nframes, DebugFrame *dbg, int *fallback, hipStream_t s);
void launch_raster(const float *xyz, size_t strideFloats, const Params &P, FrameState *st, unsigned long long *stepImg,
                   const uint2 *tileMasks, size_t tileMaskStride, int nframes, int chunkPoints, const DepthSrc *depth, const int *fallback, hipStream_t s);
void launch_outline(const Params &P, FrameState *st, unsigned long long *stepImg, unsigned long long *planeImg, int nframes, DebugFrame *dbg, unsigned long long *dbgImg, hipStream_t s);
void launch_quads(const Params &P, FrameState *st, int nframes, DebugFrame *dbg, hipStream_t s);
void launch_inquad(const float *xyz, size_t strideFloats, const Params &P, FrameState *st, unsigned long long *groundImg,
                   const uint2 *tileMasks, size_t tileMaskStride, int nframes, int chunkPoints, const DepthSrc *depth, hipStream_t s);
void launch_final(const Params &P, FrameState *st, unsigned long long *groundImg, ssd_frame_result *results, int nframes, DebugFrame *dbg, unsigned long long *dbgImg, hipStream_t s);
void launch_risers(const float *xyz, size_t strideFloats, const Params &P, FrameState *st, const uint2 *tileMasks, size_t tileMaskStride,
                   ssd_frame_risers *out, int nframes, int chunkPoints, const DepthSrc *depth, hipStream_t s);
}

#endif /* SSD_LAUNCH_H_ */
