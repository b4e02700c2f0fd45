#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
  float x = threadIdx.x * 1e-3f;
  v2f a0 = {x, x + 1}, a1 = {x + 2, x + 3}, a2 = {x + 4, x + 5}, a3 = {x + 6, x + 7};
  v2f a4 = {x, x + 1.5f}, a5 = {x + 2, x + 3.5f}, a6 = {x + 4, x + 5.5f}, a7 = {x + 6, x + 7.5f};
  const v2f m = {0.999f, 1.001f}, c = {1e-3f, 2e-3f};
  float s0 = x, s1 = x + 1, s2 = x + 2, s3 = x + 3, s4 = x + 4, s5 = x + 5, s6 = x + 6, s7 = x + 7;
  for (int i = 0; i < iters; i++)
  {
#pragma unroll
    for (int u = 0; u < 16; u++)
    {
      if (MODE == 0)
      {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
      }
      else
      {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(m.x), "v"(c.x));
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0.x + a1.y + a2.x + a3.y + a4.x + a5.y + a6.x + a7.y + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
}
int main()
{
  float *d; hipMalloc(&d, 1 << 24);
  const int blocks = 256 * 8, iters = 2000;
  for (int mode = 0; mode < 2; mode++)
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++)
    {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
      else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 /*waves*/ * iters * 16 * 8; // wave-instructions
    printf("%s: %.3f ms, %.1f G wave-instr/s, per SIMD (1024): %.2f instr/us -> %.2f cycles/instr at 2.4 GHz\n", mode == 0 ? "v_pk_fma_f32" : "v_fma_f32", ms,
           insts / ms / 1e6, insts / ms / 1e3 / 1024, 2400.0 / (insts / ms / 1e3 / 1024));
  }
  return 0;
}
