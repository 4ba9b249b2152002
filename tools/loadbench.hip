// tools/loadbench.hip — micro-benchmark (not product code): how fast can gfx950 stream AoS float3 points?
// Variants: A = global_load_dwordx3, lane-contiguous 12 B (768 B per wave instruction)
//           B = 3 x global_load_dwordx4 per lane, each lane owns 4 consecutive points (48 B, lane stride 48 B)
//           C = 3 x global_load_dwordx4 wave-contiguous (plain float4 stream; upper bound, no whole points per lane)
//           D = C + LDS transpose so that each lane ends with 4 whole points
// Build: hipcc --offload-arch=gfx950 -O3 tools/loadbench.hip -o gpurun_out/loadbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while(0)
struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };

__global__ __launch_bounds__(256) void kA(const float *p, size_t npts, int chunk, float *out)
{
  size_t begin = (size_t)blockIdx.x * chunk, end = begin + chunk < npts ? begin + chunk : npts;
  float acc = 0;
  for(size_t i0 = begin; i0 < end; i0 += 1024)
  {
    F3 v[4];
#pragma unroll
    for(int j = 0; j < 4; j++) { size_t idx = i0 + j * 256 + threadIdx.x; v[j] = idx < end ? *(const F3 *)(p + 3 * idx) : F3{0, 0, 0}; }
#pragma unroll
    for(int j = 0; j < 4; j++) acc += v[j].x * 1.0001f + v[j].y * 0.5f + v[j].z;
  }
  if(acc == 1234.5f) out[0] = acc;
}
__global__ __launch_bounds__(256) void kB(const float *p, size_t npts, int chunk, float *out)
{
  size_t begin = (size_t)blockIdx.x * chunk, end = begin + chunk < npts ? begin + chunk : npts;
  float acc = 0;
  for(size_t i0 = begin; i0 < end; i0 += 1024)
  {
    size_t idx = i0 + 4 * threadIdx.x;
    float4 a = {0, 0, 0, 0}, b = a, c = a;
    if(idx + 3 < end)
    {
      const float4 *q = (const float4 *)(p + 3 * idx);
      a = q[0]; b = q[1]; c = q[2];
    }
    acc += a.x * 1.0001f + a.y * 0.5f + a.z + a.w * 1.0001f + b.x * 0.5f + b.y + b.z * 1.0001f + b.w * 0.5f + c.x + c.y * 1.0001f + c.z * 0.5f + c.w;
  }
  if(acc == 1234.5f) out[0] = acc;
}
__global__ __launch_bounds__(256) void kC(const float *p, size_t npts, int chunk, float *out)
{
  size_t begin = (size_t)blockIdx.x * chunk, end = begin + chunk < npts ? begin + chunk : npts;
  float acc = 0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for(size_t i0 = begin; i0 < end; i0 += 1024)
  {
    // wave w owns points i0 + 256 w .. +255 = 768 floats = 192 float4: 3 instructions of 64 float4
    const float4 *q = (const float4 *)(p + 3 * (i0 + 256 * wave));
    float4 a = {0, 0, 0, 0}, b = a, c = a;
    if(i0 + 256 * wave + 255 < end) { a = q[lane]; b = q[64 + lane]; c = q[128 + lane]; }
    acc += a.x * 1.0001f + a.y * 0.5f + a.z + a.w * 1.0001f + b.x * 0.5f + b.y + b.z * 1.0001f + b.w * 0.5f + c.x + c.y * 1.0001f + c.z * 0.5f + c.w;
  }
  if(acc == 1234.5f) out[0] = acc;
}
__global__ __launch_bounds__(256) void kD(const float *p, size_t npts, int chunk, float *out)
{
  __shared__ float4 lds[4][192];
  size_t begin = (size_t)blockIdx.x * chunk, end = begin + chunk < npts ? begin + chunk : npts;
  float acc = 0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for(size_t i0 = begin; i0 < end; i0 += 1024)
  {
    const float4 *q = (const float4 *)(p + 3 * (i0 + 256 * wave));
    float4 a = {0, 0, 0, 0}, b = a, c = a;
    if(i0 + 256 * wave + 255 < end) { a = q[lane]; b = q[64 + lane]; c = q[128 + lane]; }
    lds[wave][lane] = a; lds[wave][64 + lane] = b; lds[wave][128 + lane] = c;
    __builtin_amdgcn_wave_barrier();
    a = lds[wave][3 * lane]; b = lds[wave][3 * lane + 1]; c = lds[wave][3 * lane + 2];   // 4 whole points of this lane
    __builtin_amdgcn_wave_barrier();
    acc += a.x * 1.0001f + a.y * 0.5f + a.z + a.w * 1.0001f + b.x * 0.5f + b.y + b.z * 1.0001f + b.w * 0.5f + c.x + c.y * 1.0001f + c.z * 0.5f + c.w;
  }
  if(acc == 1234.5f) out[0] = acc;
}
template<typename K> float run(K k, const float *d, size_t npts, int chunk, float *out, int reps)
{
  int blocks = (int)((npts + chunk - 1) / chunk);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, npts, chunk, out);
  hipEventRecord(e0);
  for(int r = 0; r < reps; r++) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, npts, chunk, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}
int main()
{
  const size_t npts = (size_t)1024 * 768 * 1024;   // 9.66 GB
  float *d, *out;
  CK(hipMalloc(&d, npts * 12)); CK(hipMalloc(&out, 16));
  CK(hipMemset(d, 0x3c, npts * 12));
  const double gb = npts * 12.0 / 1e9;
  for(int chunk : {16384, 49152, 196608, 786432})
  {
    float a = run(kA, d, npts, chunk, out, 5), b = run(kB, d, npts, chunk, out, 5), c = run(kC, d, npts, chunk, out, 5), dd = run(kD, d, npts, chunk, out, 5);
    printf("chunk %7d: A dwordx3 %.3f ms %.0f GB/s | B x4 stride48 %.3f ms %.0f GB/s | C x4 contiguous %.3f ms %.0f GB/s | D x4+LDS transpose %.3f ms %.0f GB/s\n",
           chunk, a, gb / a * 1e3, b, gb / b * 1e3, c, gb / c * 1e3, dd, gb / dd * 1e3);
  }
  return 0;
}
