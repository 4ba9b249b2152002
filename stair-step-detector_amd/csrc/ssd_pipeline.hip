/*
 * ssd_pipeline.hip — batches overlapped across handles (include/ssd_hip.h, "ssd_pipeline_*").
 *
 * One handle runs a batch as a chain of dependent launches; its small kernels (one block per frame or image) leave the GPU
 * nearly empty for about 0.1 ms per batch.  A pipeline owns `depth` handles — each with its own workspace — and as many
 * HIP streams, and deals the submitted batches out round-robin: the launches of one batch fill the gaps of the others
 * (64 frames per batch 126 k -> 199 k frames/s, 1024: 247 k -> 255 k).  Built purely on the handle
 * entry points; results come back in submission order.
 */
#include "../../include/ssd_hip.h"
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

struct ssd_pipeline
{
  int device = 0, depth = 0;
  std::vector<ssd_handle *> handles;
  std::vector<hipStream_t> streams;
  std::vector<hipEvent_t> produced;       /* ssd_pipeline_submit_after: recorded on the producer's stream */
  std::vector<int> frames;                /* frames of the batch each handle holds, 0 = idle */
  unsigned long long submitted = 0, fetched = 0;
  int lastFetched = -1;                   /* handle of the batch ssd_pipeline_next returned last */
};

namespace
{
thread_local std::string g_perr;
int pfail(int code, const std::string &msg)
{
  g_perr = msg;
  return code;
}
}

extern "C"
{

const char *ssd_pipeline_last_error(void)
{
  return g_perr.empty() ? ssd_last_error() : g_perr.c_str();
}

int ssd_pipeline_destroy(ssd_pipeline *p)
{
  if(!p)
    return SSD_OK;
  (void)hipSetDevice(p->device);
  for(ssd_handle *h : p->handles)
    ssd_destroy(h);
  for(hipStream_t s : p->streams)
    if(s) (void)hipStreamDestroy(s);
  for(hipEvent_t e : p->produced)
    if(e) (void)hipEventDestroy(e);
  delete p;
  return SSD_OK;
}

int ssd_pipeline_create(const ssd_config *cfg, const ssd_calibration *cal, int device, int depth, ssd_pipeline **out)
{
  if(!cfg || !cal || !out || depth < 1 || depth > 8)
    return pfail(SSD_E_ARG, "ssd_pipeline_create: bad argument (depth 1..8)");
  *out = nullptr;
  g_perr.clear();
  ssd_pipeline *p = new ssd_pipeline();
  p->device = device;
  p->depth = depth;
  ssd_config one = *cfg;
  one.batches_in_flight = 1;               /* the overlap is across this pipeline's handles: one workspace each */
  for(int k = 0; k < depth; k++)
  {
    ssd_handle *h = nullptr;
    const int rc = ssd_create(&one, cal, device, &h);
    if(rc != SSD_OK)
    {
      ssd_pipeline_destroy(p);
      return rc;                            /* ssd_last_error() has the reason */
    }
    p->handles.push_back(h);
    hipStream_t s = nullptr;
    if(hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
    {
      ssd_pipeline_destroy(p);
      return pfail(SSD_E_HIP, "ssd_pipeline_create: hipStreamCreateWithFlags failed");
    }
    p->streams.push_back(s);
    hipEvent_t e = nullptr;
    if(hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
    {
      ssd_pipeline_destroy(p);
      return pfail(SSD_E_HIP, "ssd_pipeline_create: hipEventCreateWithFlags failed");
    }
    p->produced.push_back(e);
    p->frames.push_back(0);
  }
  *out = p;
  return SSD_OK;
}

int ssd_pipeline_submit_after(ssd_pipeline *p, const void *d_xyz, size_t frame_stride_bytes, int nframes, void *producer_stream, int use_producer)
{
  if(!p)
    return pfail(SSD_E_ARG, "ssd_pipeline_submit: null pipeline");
  g_perr.clear();
  if(p->submitted - p->fetched >= static_cast<unsigned long long>(p->depth))
    return pfail(SSD_E_CAP, "ssd_pipeline_submit: every handle holds an unfetched batch: call ssd_pipeline_next first");
  const int k = static_cast<int>(p->submitted % static_cast<unsigned long long>(p->depth));
  if(use_producer)
  {
    /* the pipeline's streams are its own: order this batch behind whatever the producer's stream holds now */
    if(hipSetDevice(p->device) != hipSuccess || hipEventRecord(p->produced[k], static_cast<hipStream_t>(producer_stream)) != hipSuccess ||
       hipStreamWaitEvent(p->streams[k], p->produced[k], 0) != hipSuccess)
      return pfail(SSD_E_HIP, "ssd_pipeline_submit_after: could not order the batch behind the producer's stream");
  }
  const int rc = ssd_enqueue(p->handles[k], d_xyz, frame_stride_bytes, nframes, p->streams[k]);
  if(rc != SSD_OK)
    return rc;
  p->frames[k] = nframes;
  p->submitted++;
  return SSD_OK;
}

int ssd_pipeline_submit(ssd_pipeline *p, const void *d_xyz, size_t frame_stride_bytes, int nframes)
{
  return ssd_pipeline_submit_after(p, d_xyz, frame_stride_bytes, nframes, nullptr, 0);
}

int ssd_pipeline_pending(const ssd_pipeline *p)
{
  return p ? static_cast<int>(p->submitted - p->fetched) : 0;
}

int ssd_pipeline_next(ssd_pipeline *p, ssd_frame_result *results, int capacity, int *nframes)
{
  if(!p || !results || !nframes)
    return pfail(SSD_E_ARG, "ssd_pipeline_next: null argument");
  g_perr.clear();
  *nframes = 0;
  if(p->submitted == p->fetched)
    return pfail(SSD_E_ARG, "ssd_pipeline_next: nothing submitted");
  const int k = static_cast<int>(p->fetched % static_cast<unsigned long long>(p->depth));
  if(capacity < p->frames[k])
    return pfail(SSD_E_CAP, "ssd_pipeline_next: results array smaller than the oldest batch");
  const int rc = ssd_fetch(p->handles[k], results, p->frames[k], p->streams[k]);
  if(rc != SSD_OK)
    return rc;
  *nframes = p->frames[k];
  p->frames[k] = 0;
  p->fetched++;
  p->lastFetched = k;
  return SSD_OK;
}

int ssd_pipeline_set_timing(ssd_pipeline *p, int enable)
{
  if(!p)
    return pfail(SSD_E_ARG, "ssd_pipeline_set_timing: null pipeline");
  g_perr.clear();
  for(ssd_handle *h : p->handles)
  {
    const int rc = ssd_set_timing(h, enable);
    if(rc != SSD_OK)
      return rc;
  }
  return SSD_OK;
}

int ssd_pipeline_stage_times(ssd_pipeline *p, float ms[7])
{
  if(!p || !ms)
    return pfail(SSD_E_ARG, "ssd_pipeline_stage_times: null argument");
  g_perr.clear();
  if(p->lastFetched < 0 || p->frames[p->lastFetched] != 0)
    return pfail(SSD_E_ARG, "ssd_pipeline_stage_times: no fetched batch whose handle has not been given the next one already");
  return ssd_get_stage_times(p->handles[p->lastFetched], ms);
}

} // extern "C"
