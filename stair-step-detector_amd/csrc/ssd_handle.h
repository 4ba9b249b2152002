/*
 * ssd_handle.h — the state behind an ssd_handle (private to the library; the test-hook library reads it to copy
 * a frame's raw device state out).
 */
#ifndef SSD_HANDLE_H_
#define SSD_HANDLE_H_

#include "ssd_device.h"
#include <vector>

/* Launch-geometry constants of a handle.  The product uses the compiled defaults below (each one measured: ssd_capi.hip,
 * choose_chunk / enqueue_impl).  Only a tools build (make EXTRA=-DSSD_TUNING OUT=../lib_tuning, tools/exp*.sh) reads
 * overrides from the environment (SSD_CHUNK_POINTS, SSD_TARGET_BLOCKS, SSD_K1_BLOCKS_PER_FRAME, SSD_K24_MIN_BLOCKS,
 * SSD_K24_TALL_BLOCKS, SSD_K2_CHUNK_TILES, SSD_K4_CHUNK_TILES, SSD_WIN_SHIFT, SSD_WIN_SHIFT_G), and only once, in ssd_create:
 * no entry point of the product library calls getenv. */
struct ssd_tuning
{
  int chunkPoints = 0;            /* > 0: the general chunk, forced */
  int targetBlocks = 32768;       /* blocks a streaming launch aims at */
  int k1BlocksPerFrame = 256;     /* K1: at most this many blocks per frame (their final atomics share the frame's histogram) */
  int k24MinBlocks = 1536, k24TallBlocks = 6144;
  int k2ChunkTiles = 32, k4ChunkTiles = 16;
  int winShift = 0, winShiftGround = 0;     /* > 0: shape of the waves' LDS image windows, forced */
  int recordPad = 0;              /* cell records per frame added to the stride between the frames' record arrays */
};

/* One complete workspace of a handle: everything a batch in flight owns on the device.  A handle has `depth` of them
 * (ssd_config::batches_in_flight); successive enqueues take them in turn, each on the lane's own stream, so that the launches
 * of one batch fill the gaps the one-block-per-frame kernels of the others leave. */
struct ssd_lane
{
  ssd::FrameState *dState = nullptr;
  unsigned long long *dStepImg = nullptr;
  unsigned long long *dGroundImg = nullptr;
  int *dFallback = nullptr;                 /* single pass: kFallbackList + F ints: frames listed for k_raster, frames without step plateaus, the list */
  unsigned long long *dPlaneImg = nullptr;  /* single pass: [F][kMaxPlanes] bit images, one per candidate height bin (null: the handle never runs it) */
  uint2 *dTileMasks = nullptr;             /* per cell (64 points): which groups of 4 height bins occur; K1 -> K2, K4, K6 */
  uint2 *dTileMasksBase = nullptr;         /* the allocation dTileMasks lies in (the tools' placement hooks put the records elsewhere in a larger one) */
  size_t recordSlack = 0;                   /* bytes of that allocation beyond the records (0 unless a tools hook allocated it) */
  hipStream_t stream = nullptr;             /* the lane's own stream (depth > 1 only; depth 1 runs on the caller's stream) */
  hipEvent_t in = nullptr;                  /* recorded on the caller's stream at the enqueue: the lane's work starts behind it */
  hipEvent_t done = nullptr;                /* recorded behind the lane's last enqueue */
  hipStream_t lastStream = nullptr;         /* depth 1: the stream of the previous call (a switch is ordered by `done`) */
  bool haveLast = false;
  bool stepImagesDirty = false, groundImageDirty = false;   /* a partial run rastered without the stage that consumes (and clears) the bits */
  int dirtyFrames = 0;                      /* leading FrameStates whose K1 accumulators may be non-zero (k_peaks clears them) */
};

constexpr int kMaxLanes = 8;

struct ssd_handle
{
  int device = 0;
  ssd_config cfg{};
  ssd::Params P{};
  ssd_tuning tune{};
  int F = 0;                      /* max frames per batch */
  size_t imgWords = 0;            /* 64-bit words per bit image */
  int depth = 1;                  /* lanes in use */
  ssd_lane lane[kMaxLanes];
  int lastLane = 0;               /* the lane of the last enqueue (debug capture, test hooks, riser fetch read it) */
  unsigned long long laneTurn = 0;
  bool lastPinned = false;        /* the last enqueue was held in lane 0 (debug capture, risers, partial stages) */
  size_t tileMaskStride = 0;      /* cell records per frame */
  size_t recordBytes = 0;         /* one workspace's cell records */
  float *dDepthMaps = nullptr;              /* xmap[W] then ymap[H] (ssd_set_intrinsics) */
  ssd_intrinsics intr{};
  bool haveIntr = false;
  /* result slots (max(2, depth)), used in turn by the enqueues that run the last stage: the device -> pinned-host copy of
   * a batch's results is part of its enqueue, so that the next batches can be enqueued before the results are read */
  int nSlots = 2;
  ssd_frame_result *dResults = nullptr;     /* nSlots x F */
  ssd_frame_result *hResults = nullptr;     /* nSlots x F, pinned */
  ssd_frame_result *hResultsDev = nullptr;  /* the same memory as the kernels address it (small batches write it directly) */
  hipEvent_t resultsReady[kMaxLanes] = {};
  int resultsFrames[kMaxLanes] = {};
  int resultsLane[kMaxLanes] = {};
  unsigned long long finalCount = 0;        /* enqueues that produced results */
  ssd_frame_risers *dRisers = nullptr;      /* vertical faces (extension), allocated by ssd_set_risers */
  ssd_frame_risers *hRisers = nullptr;      /* pinned */
  /* ssd_process_host / ssd_process_depth_host: two device staging buffers, a copy and a compute stream (ssd_capi.hip) */
  void *ingestBuf[2] = { nullptr, nullptr };
  size_t ingestCap = 0;                     /* bytes per buffer */
  hipStream_t ingestCopy = nullptr, ingestCompute = nullptr;
  hipStream_t ingestCopy2 = nullptr;            /* the second half of a large slice: two copy engines (round 6) */
  hipEvent_t ingestCopied[2] = { nullptr, nullptr }, ingestConsumed[2] = { nullptr, nullptr };
  hipEvent_t ingestCopied2[2] = { nullptr, nullptr };
  /* risers of a host-fed batch, slice by slice (the device buffer holds one enqueue's) */
  ssd_frame_risers *hRisersBatch = nullptr; /* pinned */
  int hRisersBatchCap = 0, hRisersBatchFrames = 0;
  ssd::DebugFrame *dDebug = nullptr;
  unsigned long long *dDebugImg = nullptr;
  int debug = 0;                  /* 0 off, 1 records + images (the whole ground image is rastered for it), 2 records only */
  int lastFrames = 0;
  /* single pass (k_hist rasters the step plateaus itself): -1 = whenever a call qualifies (whole pipeline, a batch of at least
   * kSinglePassMinPoints points, vertex input, geometry), 0 = never, 1 = whenever the geometry allows; sabotage: k_predict's (test
   * hooks set both) */
  int singlePassMode = -1, singlePassSabotage = 0;
  int planePool = 0;               /* planes k_predict may hand out per batch (plane_pool_size(F), what each workspace holds; a test hook lowers it) */
  bool lastSinglePass = false;    /* the last enqueue ran it */
  int *hFallback = nullptr;       /* pinned, two per result slot: frames of that batch k_raster had to do, frames without step plateaus (copied with the results) */
  int resultsFallback[kMaxLanes] = {};      /* -1: that slot's batch ran two passes; 0: count on its way; 1: seen by ssd_fetch_back */
  int singlePassBackoff = 0;      /* qualifying batches still to run two passes after a batch the predictor did not cover */
  size_t bytes = 0;
  /* per-stage timing: a ring of event sets, one per enqueue, so that a timed loop never has to synchronise */
  bool timing = false;
  std::vector<hipEvent_t> ev;     /* kTimingSlots x 8 */
  hipEvent_t evPredict[SSD_TIMING_SLOTS] = {};   /* recorded in front of k_predict (single-pass enqueues) */
  bool predictTimed[SSD_TIMING_SLOTS] = {};
  unsigned long long enqueueCount = 0;
  unsigned long long timedFrom = 0;
};

#endif /* SSD_HANDLE_H_ */
