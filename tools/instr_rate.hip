// tools/instr_rate.hip — micro-benchmark (not product code): issue cost of the vector instructions the point loop is made of, on
// gfx950.  Every kernel runs `iters` rounds of 32 INDEPENDENT copies of one instruction per lane (inline asm, so nothing is
// folded), 4 waves per SIMD resident; reported: cycles per wave-instruction per SIMD, relative to v_add_f32 (= 4 on a 16-lane SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 tools/instr_rate.hip -o gpurun_out/instr_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while(0)

#define REP8(x) x x x x x x x x
#define KERNEL(name, decl, body)                                                                   \
  __global__ __launch_bounds__(256) void name(int iters, float *out, float seed)                   \
  {                                                                                                \
    decl                                                                                           \
    for(int i = 0; i < iters; i++) { REP8(body) REP8(body) REP8(body) REP8(body) }                  \
    if(seed == 12345.0f) out[threadIdx.x] = (float)sink;                                           \
  }

KERNEL(k_add_f32, float a = seed + threadIdx.x; float sink = 0;, asm volatile("v_add_f32 %0, %1, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_mul_f64, double a = seed + threadIdx.x; double sink = 0;, asm volatile("v_mul_f64 %0, %1, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_add_f64, double a = seed + threadIdx.x; double sink = 0;, asm volatile("v_add_f64 %0, %1, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_fma_f64, double a = seed + threadIdx.x; double sink = 0;, asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_cvt_f64_f32, float a = seed + threadIdx.x; double sink = 0;, asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_cvt_f32_f64, double a = seed + threadIdx.x; float sink = 0;, asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_cvt_i32_f64, double a = seed + threadIdx.x; int sink = 0;, asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_cvt_u32_f64, double a = seed + threadIdx.x; unsigned sink = 0;, asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_cmp_f64, double a = seed + threadIdx.x; float sink = 0;, asm volatile("v_cmp_lt_f64 vcc, %0, %0" : : "v"(a) : "vcc");)
KERNEL(k_cmp_f32, float a = seed + threadIdx.x; float sink = 0;, asm volatile("v_cmp_lt_f32 vcc, %0, %0" : : "v"(a) : "vcc");)
KERNEL(k_min_u32, unsigned a = threadIdx.x; unsigned sink = 0;, asm volatile("v_min_u32 %0, %1, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_min_u32_dpp, unsigned a = threadIdx.x; unsigned sink = 0;, asm volatile("v_min_u32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "=v"(sink) : "v"(a));)
KERNEL(k_lshl_or, unsigned a = threadIdx.x; unsigned sink = 0;, asm volatile("v_lshl_or_b32 %0, %1, 7, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_lshl_add_u64, unsigned long long a = threadIdx.x; unsigned long long sink = 0;, asm volatile("v_lshl_add_u64 %0, %1, 0, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_pk_min_u16, unsigned a = threadIdx.x; unsigned sink = 0;, asm volatile("v_pk_min_u16 %0, %1, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_cndmask, unsigned a = threadIdx.x; unsigned sink = 0;, asm volatile("v_cndmask_b32 %0, %1, %1, vcc" : "=v"(sink) : "v"(a) : "vcc");)


/* round 5: the single-precision instructions of K1's x / y pre-filter and of its candidate queue */
KERNEL(k_fma_f32, float a = seed + threadIdx.x; float sink = 0;, asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_mul_f32, float a = seed + threadIdx.x; float sink = 0;, asm volatile("v_mul_f32 %0, %1, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_max_f32, float a = seed + threadIdx.x; float sink = 0;, asm volatile("v_max_f32 %0, %1, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_max_f32_abs, float a = seed + threadIdx.x; float sink = 0;, asm volatile("v_max_f32 %0, |%1|, |%1|" : "=v"(sink) : "v"(a));)
KERNEL(k_min_f32_dpp, float a = seed + threadIdx.x; float sink = 0;, asm volatile("v_min_f32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "=v"(sink) : "v"(a));)
KERNEL(k_max3_f32, float a = seed + threadIdx.x; float sink = 0;, asm volatile("v_max3_f32 %0, |%1|, |%1|, |%1|" : "=v"(sink) : "v"(a));)
KERNEL(k_pk_fma_f32, double a = seed + threadIdx.x; double sink = 0;, asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_pk_fma_f32_sel, double a = seed + threadIdx.x; double sink = 0;, asm volatile("v_pk_fma_f32 %0, %1, %1, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(sink) : "v"(a));)
KERNEL(k_pk_add_f32, double a = seed + threadIdx.x; double sink = 0;, asm volatile("v_pk_add_f32 %0, %1, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_pk_mul_f32, double a = seed + threadIdx.x; double sink = 0;, asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_fract_f32, float a = seed + threadIdx.x; float sink = 0;, asm volatile("v_fract_f32 %0, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_cvt_u32_f32, float a = seed + threadIdx.x; unsigned sink = 0;, asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_cvt_pk_u8_f32, float a = seed + threadIdx.x; unsigned sink = 0;, asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(sink) : "v"(a));)
KERNEL(k_mbcnt, unsigned a = threadIdx.x; unsigned sink = 0;, asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_and_b32, unsigned a = threadIdx.x; unsigned sink = 0;, asm volatile("v_and_b32 %0, %1, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_add_u32, unsigned a = threadIdx.x; unsigned sink = 0;, asm volatile("v_add_u32 %0, %1, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_cmp_u32, unsigned a = threadIdx.x; float sink = 0;, asm volatile("v_cmp_lt_u32 vcc, %0, %0" : : "v"(a) : "vcc");)
KERNEL(k_cmp_f32_sgpr, float a = seed + threadIdx.x; float sink = 0;, asm volatile("v_cmp_lt_f32 s[20:21], %0, %0" : : "v"(a) : "s20", "s21");)
KERNEL(k_cmpx_f32, float a = seed + threadIdx.x; float sink = 0;, asm volatile("v_cmp_class_f32 vcc, %0, 3" : : "v"(a) : "vcc");)
KERNEL(k_perm_b32, unsigned a = threadIdx.x; unsigned sink = 0;, asm volatile("v_perm_b32 %0, %1, %1, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_mov_b32, unsigned a = threadIdx.x; unsigned sink = 0;, asm volatile("v_mov_b32 %0, %1" : "=v"(sink) : "v"(a));)
KERNEL(k_cndmask_sgpr, unsigned a = threadIdx.x; unsigned sink = 0;, asm volatile("v_cndmask_b32 %0, %1, %1, s[20:21]" : "=v"(sink) : "v"(a));)
KERNEL(k_cndmask_indep, unsigned a = threadIdx.x; unsigned sink = 0; unsigned s2 = 0; unsigned s3 = 0; unsigned s4 = 0;, asm volatile("v_cndmask_b32 %0, %4, %4, vcc\n v_cndmask_b32 %1, %4, %4, vcc\n v_cndmask_b32 %2, %4, %4, vcc\n v_cndmask_b32 %3, %4, %4, vcc" : "=v"(sink), "=v"(s2), "=v"(s3), "=v"(s4) : "v"(a) : "vcc"); sink += 0;)
KERNEL(k_add_f32_indep, float a = seed + threadIdx.x; float sink = 0; float s2 = 0; float s3 = 0; float s4 = 0;, asm volatile("v_add_f32 %0, %4, %4\n v_add_f32 %1, %4, %4\n v_add_f32 %2, %4, %4\n v_add_f32 %3, %4, %4" : "=v"(sink), "=v"(s2), "=v"(s3), "=v"(s4) : "v"(a)); sink += 0;)
KERNEL(k_add_f64_indep, double a = seed + threadIdx.x; double sink = 0; double s2 = 0; double s3 = 0; double s4 = 0;, asm volatile("v_add_f64 %0, %4, %4\n v_add_f64 %1, %4, %4\n v_add_f64 %2, %4, %4\n v_add_f64 %3, %4, %4" : "=v"(sink), "=v"(s2), "=v"(s3), "=v"(s4) : "v"(a)); sink += 0;)
KERNEL(k_min_u32_indep, unsigned a = threadIdx.x; unsigned sink = 0; unsigned s2 = 0; unsigned s3 = 0; unsigned s4 = 0;, asm volatile("v_min_u32 %0, %4, %4\n v_min_u32 %1, %4, %4\n v_min_u32 %2, %4, %4\n v_min_u32 %3, %4, %4" : "=v"(sink), "=v"(s2), "=v"(s3), "=v"(s4) : "v"(a)); sink += 0;)
__global__ __launch_bounds__(256) void k_ds_write(int iters, float *out, float seed)
{
  __shared__ unsigned int lds[256 * 8];
  unsigned int *mine = lds + threadIdx.x;
  unsigned v = (unsigned)seed + threadIdx.x;
  for(int i = 0; i < iters; i++)
  {
#pragma unroll
    for(int k = 0; k < 32; k++) { asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"((unsigned)(size_t)mine), "v"(v), "n"((k & 7) * 1024) : "memory"); }
  }
  __syncthreads();
  if(seed == 12345.0f) out[threadIdx.x] = (float)lds[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_ds_add(int iters, float *out, float seed)
{
  __shared__ unsigned int lds[128 * 32];
  for(int i = threadIdx.x; i < 128 * 32; i += 256) lds[i] = 0;
  __syncthreads();
  unsigned int *mine = lds + (threadIdx.x & 31);
  unsigned bin = (threadIdx.x * 7 + (unsigned)seed) & 127;
  for(int i = 0; i < iters; i++)
  {
#pragma unroll
    for(int k = 0; k < 32; k++) { atomicAdd(mine + ((bin + k) & 127) * 32, 1u); }
  }
  __syncthreads();
  if(seed == 12345.0f) out[threadIdx.x] = (float)lds[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_ds_read_u8(int iters, float *out, float seed)
{
  __shared__ unsigned char lut[128];
  if(threadIdx.x < 128) lut[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  unsigned idx = (threadIdx.x + (unsigned)seed) & 127, acc = 0;
  for(int i = 0; i < iters; i++)
  {
#pragma unroll
    for(int k = 0; k < 32; k++) { acc += lut[(idx + k * 5 + acc) & 127]; }
  }
  if(seed == 12345.0f) out[threadIdx.x] = (float)acc;
}

typedef void (*kern_t)(int, float *, float);
int main()
{
  float *out; CK(hipMalloc(&out, 4096));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, blocks = cus * 4, iters = 4000;    /* 4 blocks x 4 waves per CU = 4 waves per SIMD */
  const double clk = prop.clockRate * 1e3;
  struct { const char *name; kern_t k; } ks[] = {
    {"v_add_f32", k_add_f32}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_fma_f64", k_fma_f64}, {"v_cvt_f64_f32", k_cvt_f64_f32},
    {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_cvt_u32_f64", k_cvt_u32_f64}, {"v_cmp_lt_f64", k_cmp_f64}, {"v_cmp_lt_f32", k_cmp_f32},
    {"v_min_u32", k_min_u32}, {"v_min_u32_dpp", k_min_u32_dpp}, {"v_lshl_or_b32", k_lshl_or}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_pk_min_u16", k_pk_min_u16},
    {"v_cndmask_b32", k_cndmask},
    {"v_fma_f32", k_fma_f32}, {"v_mul_f32", k_mul_f32}, {"v_max_f32", k_max_f32}, {"v_max_f32 |a|,|a|", k_max_f32_abs}, {"v_min_f32_dpp", k_min_f32_dpp}, {"v_max3_f32 |..|", k_max3_f32},
    {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_fma_f32 op_sel", k_pk_fma_f32_sel}, {"v_pk_add_f32", k_pk_add_f32}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_fract_f32", k_fract_f32},
    {"v_cvt_u32_f32", k_cvt_u32_f32}, {"v_cvt_pk_u8_f32", k_cvt_pk_u8_f32}, {"v_mbcnt_lo_u32_b32", k_mbcnt}, {"v_and_b32", k_and_b32}, {"v_add_u32", k_add_u32},
    {"v_cmp_lt_u32", k_cmp_u32}, {"v_cmp_lt_f32 -> sgpr pair", k_cmp_f32_sgpr}, {"v_cmp_class_f32", k_cmpx_f32}, {"v_perm_b32", k_perm_b32}, {"v_mov_b32", k_mov_b32},
    {"v_cndmask_b32 sgpr mask", k_cndmask_sgpr}, {"v_cndmask_b32 x4 independent dst", k_cndmask_indep}, {"v_add_f32 x4 independent dst", k_add_f32_indep},
    {"v_add_f64 x4 independent dst", k_add_f64_indep}, {"v_min_u32 x4 independent dst", k_min_u32_indep}, {"ds_write_b32", k_ds_write}, {"ds_add_u32 (conflict-free)", k_ds_add}, {"ds_read_u8 dependent chain", k_ds_read_u8} };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%d CUs, %.0f MHz nominal; 4 waves per SIMD, 32 instructions per round\n", cus, clk / 1e6);
  for(auto &k : ks)
  {
    hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, 100, out, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, iters, out, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    /* per SIMD: 4 waves x iters x 32 instructions */
    const double per = ms * 1e-3 * clk / (4.0 * iters * 32.0);
    printf("%-28s %7.3f ms  %6.2f cycles per wave-instruction per SIMD (nominal clock)\n", k.name, ms, per);
  }
  return 0;
}
