/*
 * ssd_phase.h — instrumentation of the tools builds only (make EXTRA=-DSSD_PHASE_TIMING OUT=../lib_phase; tools/phases.py,
 * tools/blockphases.py): clocks left at marked places of the kernels.  In the product build every macro and method here is
 * empty; the kernels keep only the marks (SSD_PHASE(k, i), BlockPhase::mark), which then compile to nothing.
 * Included by ssd_kernels.hip inside namespace ssd; ssd_phase_readers.h (at the end of that file) holds the two entry points
 * the tools read the clocks through.
 */
#ifndef SSD_PHASE_H_
#define SSD_PHASE_H_

/* Phase clocks of the latency-bound kernels (tools/phases.py; a tools-only build: make EXTRA=-DSSD_PHASE_TIMING
 * OUT=../lib_phase).  Block (0, 0) leaves wall_clock64() — 100 MHz — at the marked places. */
#ifdef SSD_PHASE_TIMING
__device__ unsigned long long g_phase[4][32];
#define SSD_PHASE_IF(cond, k, i) do { if(blockIdx.x == 0 && blockIdx.y == 0 && (cond)) g_phase[k][i] = wall_clock64(); } while(0)
#else
#define SSD_PHASE_IF(cond, k, i) do { } while(0)
#endif
#define SSD_PHASE(k, i) SSD_PHASE_IF(threadIdx.x == 0, k, i)

/* Where a block of a streaming kernel spends its life (tools/blockphases.py; the same tools-only build): lane 0 of EVERY wave adds
 * the wall-clock ticks (100 MHz) between consecutive marks to g_blockphase[kernel][mark]; [kernel][15] counts the waves. */
#ifdef SSD_PHASE_TIMING
__device__ unsigned long long g_blockphase[4][64][8];      /* [kernel][copy = blockIdx.x & 63][mark]; [..][7] counts the waves */
struct BlockPhase
{
  unsigned long long t, acc[7];
  int k;
  __device__ __forceinline__ BlockPhase(int kernel) : t(wall_clock64()), k(kernel)
  {
    for(int i = 0; i < 7; i++) acc[i] = 0ull;
  }
  __device__ __forceinline__ void mark(int i)
  {
    const unsigned long long now = wall_clock64();
    acc[i] += now - t;
    t = now;
  }
  /* at the very end: the wave's sums into one of 64 copies (no hot address while the phases are being measured) */
  __device__ __forceinline__ void finish()
  {
    if((threadIdx.x & 63) == 0)
    {
      unsigned long long *g = g_blockphase[k][blockIdx.x & 63];
      for(int i = 0; i < 7; i++)
        if(acc[i]) atomicAdd(&g[i], acc[i]);
      atomicAdd(&g[7], 1ull);
    }
  }
};
#else
struct BlockPhase
{
  __device__ __forceinline__ BlockPhase(int) { }
  __device__ __forceinline__ void mark(int) { }
  __device__ __forceinline__ void finish() { }
};
#endif

#endif /* SSD_PHASE_H_ */
