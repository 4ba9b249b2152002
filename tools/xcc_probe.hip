// tools/xcc_probe.hip — which XCD does workgroup (x, y) of a (1024, 24) grid of 256-thread blocks run on?  K1 deals a frame's blocks to one XCD by
// putting the frame on the fast grid axis, ASSUMING workgroups go round-robin over the 8 XCDs in linear order.  Prints how many blocks sit on XCD
// (linear id % 8 + the first block's XCD) % 8, and how many XCDs a frame's 24 blocks touch.  hipcc --offload-arch=gfx950 -O2 tools/xcc_probe.hip -o /tmp/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
__global__ __launch_bounds__(256, 5) void probe(unsigned char *out, float *sink, int spin)
{
  __shared__ float lds[7600];                                   // ~30 KB, K1's footprint: 5 blocks per CU
  const unsigned int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);     // HW_REG_XCC_ID, bits 3:0
  float a = threadIdx.x;
  for(int i = 0; i < spin; i++)
    a = a * 1.0001f + 0.5f;
  lds[threadIdx.x] = a;
  __syncthreads();
  if(threadIdx.x == 0)
  {
    out[blockIdx.y * gridDim.x + blockIdx.x] = static_cast<unsigned char>(xcc);
    if(lds[17] == 12345.0f) *sink = a;
  }
}
int main(int argc, char **argv)
{
  const int F = 1024, C = 24, spin = argc > 1 ? atoi(argv[1]) : 20000;
  unsigned char *d; float *sink;
  hipMalloc(&d, F * C); hipMalloc(&sink, 4);
  for(int rep = 0; rep < 3; rep++)
  {
    hipMemset(d, 0xff, F * C);
    hipLaunchKernelGGL(probe, dim3(F, C), dim3(256), 0, nullptr, d, sink, spin);
    hipDeviceSynchronize();
    std::vector<unsigned char> h(F * C);
    hipMemcpy(h.data(), d, F * C, hipMemcpyDeviceToHost);
    int hist[16] = {0}, aligned = 0, first = h[0];
    for(int i = 0; i < F * C; i++) { hist[h[i] & 15]++; aligned += (h[i] == (first + i) % 8); }
    int spread[9] = {0};
    for(int f = 0; f < F; f++) { unsigned m = 0; for(int c = 0; c < C; c++) m |= 1u << h[c * F + f]; spread[__builtin_popcount(m)]++; }
    printf("rep %d: first block on XCD %d; blocks on (first + id) %% 8: %d of %d; blocks per XCD:", rep, first, aligned, F * C);
    for(int x = 0; x < 8; x++) printf(" %d", hist[x]);
    printf("; frames by number of XCDs their 24 blocks touch:");
    for(int k = 1; k <= 8; k++) printf(" %d:%d", k, spread[k]);
    printf("\n");
  }
  return 0;
}
