// tools/storebench.hip — micro-benchmark (not product code): cost of emitting 1 byte per point beside a 12 B/point read stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while(0)

// MODE 0: no store; 1: dword store per lane per tile; 2: same, nontemporal; 3: stage 4 tiles in LDS, 16-B stores; 4: dword store to a
// buffer that is NOT hipMalloc'ed near... (same); 5: store only every 4th tile a 16-B value made of 4 tiles (registers, strided layout [tile%4])
template<int MODE>
__global__ __launch_bounds__(256) void k(const float *p, unsigned int *out, size_t npts, int chunk, float *sink)
{
  __shared__ unsigned int stage[4][256];
  size_t begin = (size_t)blockIdx.x * chunk, end = begin + chunk < npts ? begin + chunk : npts;
  float acc = 0;
  unsigned int hold[4] = {0, 0, 0, 0};
  int it = 0;
  for(size_t i0 = begin; i0 < end; i0 += 1024, it++)
  {
    size_t idx = i0 + 4 * threadIdx.x;
    float4 a = {0, 0, 0, 0}, b = a, c = a;
    if(idx + 3 < end)
    {
      const float4 *q = (const float4 *)(p + 3 * idx);
      a = q[0]; b = q[1]; c = q[2];
    }
    float s = a.x * 1.0001f + a.y * 0.5f + a.z + a.w * 1.0001f + b.x * 0.5f + b.y + b.z * 1.0001f + b.w * 0.5f + c.x + c.y * 1.0001f + c.z * 0.5f + c.w;
    acc += s;
    unsigned int pack = __float_as_uint(s) | 0x01010101u;
    if(MODE == 1) out[idx >> 2] = pack;
    if(MODE == 2) __builtin_nontemporal_store(pack, &out[idx >> 2]);
    if(MODE == 3)
    {
      stage[it & 3][threadIdx.x] = pack;
      if((it & 3) == 3)
      {
        __syncthreads();
        // 4 tiles = 4096 B contiguous per tile... each tile's 1024 B is contiguous; write tile t by threads 64t..64t+63 as uint4
        const int t = threadIdx.x >> 6, l = threadIdx.x & 63;
        uint4 vq = ((const uint4 *)stage[t])[l];
        ((uint4 *)(out + ((i0 - 3 * 1024 + t * 1024) >> 2)))[l] = vq;
        __syncthreads();
      }
    }
    if(MODE == 5)
    {
      hold[it & 3] = pack;
      if((it & 3) == 3)
      {
        // layout: 4 consecutive tiles interleaved per lane: out16[(i0base/4096)*256 + tid] = {t0,t1,t2,t3}
        ((uint4 *)out)[((i0 - 3 * 1024) >> 12) * 256 + threadIdx.x] = make_uint4(hold[0], hold[1], hold[2], hold[3]);
      }
    }
  }
  if(acc == 1234.5f) sink[0] = acc;
}
template<int MODE> float run(const float *d, unsigned int *out, size_t npts, int chunk, float *sink)
{
  int blocks = (int)((npts + chunk - 1) / chunk);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, out, npts, chunk, sink);
  (void)hipEventRecord(e0);
  for(int r = 0; r < 5; r++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, out, npts, chunk, sink);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}
int main()
{
  const size_t npts = (size_t)1024 * 768 * 1024;
  float *d, *sink; unsigned int *out;
  CK(hipMalloc(&d, npts * 12)); CK(hipMalloc(&sink, 16)); CK(hipMalloc(&out, npts));
  CK(hipMemset(d, 0x3c, npts * 12));
  const int chunk = 49152;
  printf("read only %.3f | dword store %.3f | nontemporal %.3f | LDS-staged 16B %.3f | 4-tile register 16B %.3f ms\n",
         run<0>(d, out, npts, chunk, sink), run<1>(d, out, npts, chunk, sink), run<2>(d, out, npts, chunk, sink),
         run<3>(d, out, npts, chunk, sink), run<5>(d, out, npts, chunk, sink));
  return 0;
}
