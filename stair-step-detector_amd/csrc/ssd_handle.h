/*
 * ssd_handle.h — the state behind an ssd_handle (private to the library; the test-hook library reads it to copy
 * a frame's raw device state out).
 */
#ifndef SSD_HANDLE_H_
#define SSD_HANDLE_H_

#include "ssd_device.h"
#include <vector>

struct ssd_handle
{
  int device = 0;
  ssd_config cfg{};
  ssd::Params P{};
  int F = 0;                      /* max frames per batch */
  size_t imgWords = 0;            /* 64-bit words per bit image */
  ssd::FrameState *dState = nullptr;
  unsigned long long *dStepImg = nullptr;
  unsigned long long *dGroundImg = nullptr;
  uint2 *dTileMasks = nullptr;             /* per cell (64 points): which groups of 4 height bins occur; K1 -> K2, K4, K6 */
  size_t tileMaskStride = 0;
  float *dDepthMaps = nullptr;              /* xmap[W] then ymap[H] (ssd_set_intrinsics) */
  ssd_intrinsics intr{};
  bool haveIntr = false;
  /* two result slots, used alternately by the enqueues that run the last stage: the device -> pinned-host copy of
   * a batch's results is part of its enqueue, so that the next batch can be enqueued before the results are read */
  ssd_frame_result *dResults = nullptr;     /* 2 x F */
  ssd_frame_result *hResults = nullptr;     /* 2 x F, pinned */
  ssd_frame_result *hResultsDev = nullptr;  /* the same memory as the kernels address it (small batches write it directly) */
  hipEvent_t resultsReady[2] = { nullptr, nullptr };
  int resultsFrames[2] = { 0, 0 };
  unsigned long long finalCount = 0;        /* enqueues that produced results */
  /* the workspace (frame state, bit images, result slots) is single-buffered: an event recorded after every enqueue
   * orders the next one behind it when the caller switches streams */
  hipEvent_t lastDone = nullptr;
  hipStream_t lastStream = nullptr;
  bool haveLast = false;
  ssd_frame_risers *dRisers = nullptr;      /* vertical faces (extension), allocated by ssd_set_risers */
  ssd_frame_risers *hRisers = nullptr;      /* pinned */
  /* ssd_process_host / ssd_process_depth_host: two device staging buffers, a copy and a compute stream (ssd_capi.hip) */
  void *ingestBuf[2] = { nullptr, nullptr };
  size_t ingestCap = 0;                     /* bytes per buffer */
  hipStream_t ingestCopy = nullptr, ingestCompute = nullptr;
  hipEvent_t ingestCopied[2] = { nullptr, nullptr }, ingestConsumed[2] = { nullptr, nullptr };
  ssd::DebugFrame *dDebug = nullptr;
  unsigned long long *dDebugImg = nullptr;
  bool debug = false;
  bool imagesDirty = false;
  int dirtyFrames = 0;                      /* leading FrameStates whose K1 accumulators may be non-zero (k_peaks clears them) */
  int lastFrames = 0;
  size_t bytes = 0;
  /* per-stage timing: a ring of event sets, one per enqueue, so that a timed loop never has to synchronise */
  bool timing = false;
  std::vector<hipEvent_t> ev;     /* kTimingSlots x 8 */
  unsigned long long enqueueCount = 0;
  unsigned long long timedFrom = 0;
};

#endif /* SSD_HANDLE_H_ */
