// tools/cvt_probe.hip — probe (not product code): how v_cvt_pk_u8_f32 and v_cvt_u32_f32 round and saturate on gfx950.
// Build: hipcc --offload-arch=gfx950 -O2 tools/cvt_probe.hip -o build/cvt_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float *in, unsigned *outA, unsigned *outB, int n)
{
  const int i = threadIdx.x;
  if(i < n)
  {
    outA[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0u, 0u);
    unsigned r;
    asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(in[i]));
    outB[i] = r;
  }
}
int main()
{
  const float h[] = { 0.0f, 0.49f, 0.5f, 0.51f, 0.99f, 1.0f, 1.5f, 2.5f, 3.5f, 254.5f, 254.99f, 255.0f, 255.5f, 256.0f, 300.0f, -0.5f, -3.0f, 1e30f, -1e30f, INFINITY, -INFINITY, NAN };
  const int n = sizeof(h) / sizeof(h[0]);
  float *d; unsigned *a, *b; hipMalloc(&d, sizeof(h)); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, a, b, n);
  unsigned ha[64], hb[64]; hipMemcpy(ha, a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, b, n * 4, hipMemcpyDeviceToHost);
  for(int i = 0; i < n; i++) printf("%12g  cvt_pk_u8 %3u   cvt_u32 %u\n", h[i], ha[i], hb[i]);
  return 0;
}
