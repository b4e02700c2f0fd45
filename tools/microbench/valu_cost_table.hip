// valu_cost_table.hip — what one wave-instruction of every mnemonic in k_descriptor's sample loop costs a gfx950 SIMD, measured:
// 8 independent register chains per wave, 8 waves per SIMD (every CU full), 16 x 8 back-to-back instructions of ONE mnemonic per loop
// iteration. Prints one JSON object {mnemonic: picoseconds per wave-instruction per SIMD} (time, not cycles: the clock the chip holds under
// each instruction differs). tools/descriptor_floor.py prices the kernel's ISA with it.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_valu_cost_table tools/microbench/valu_cost_table.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP8(INSTR) asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(c0), "v"(c1) : "vcc", "scc", "s20", "s21")
#define OP8P(INSTR) asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7) : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(pc0), "v"(pc1))
typedef float f2 __attribute__((ext_vector_type(2)));

#define I0(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I1(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I2(n) "v_add_f32 %" #n ", %" #n ", %9\n"
#define I3(n) "v_sub_f32 %" #n ", %" #n ", %9\n"
#define I4(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define I5(n) "v_fmamk_f32 %" #n ", %" #n ", 0x3fb8aa3b, %9\n"
#define I6(n) "v_fmaak_f32 %" #n ", %" #n ", %8, 0x3d2aaa72\n"
#define I7(n) "v_mul_f32_e64 %" #n ", %8, |%" #n "|\n"
#define I8(n) "v_add_u32 %" #n ", %" #n ", %9\n"
#define I9(n) "v_and_b32 %" #n ", %" #n ", %9\n"
#define I10(n) "v_mov_b32 %" #n ", %8\n"
#define I11(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define I12(n) "v_lshl_add_u32 %" #n ", %" #n ", 1, %9\n"
#define I13(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define I14(n) "v_add_lshl_u32 %" #n ", %" #n ", %9, 1\n"
#define I15(n) "v_mul_u32_u24 %" #n ", %" #n ", %9\n"
#define I16(n) "v_cvt_u32_f32 %" #n ", %" #n "\n"
#define I17(n) "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define I18(n) "v_cvt_f32_i32 %" #n ", %" #n "\n"
#define I19(n) "v_floor_f32 %" #n ", %" #n "\n"
#define I20(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I21(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\n"
#define I22(n) "v_cmp_gt_f32 vcc, %" #n ", %8\n"
#define I23(n) "v_cmp_gt_f32_e64 s[20:21], %" #n ", %8\n"
#define I24(n) "v_cmp_lt_u32 vcc, %" #n ", %8\n"
#define I25(n) "v_cmp_lt_u32_e64 s[20:21], %" #n ", %8\n"
#define I26(n) "v_cmp_class_f32 vcc, %" #n ", %9\n"
#define I27(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define I28(n) "v_sqrt_f32 %" #n ", %" #n "\n"
#define I29(n) "v_div_scale_f32 %" #n ", vcc, %" #n ", %8, %" #n "\n"
#define I30(n) "v_div_fmas_f32 %" #n ", %" #n ", %8, %9\n"
#define I31(n) "v_div_fixup_f32 %" #n ", %" #n ", %8, %9\n"
#define I32(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I33(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define I34(n) "v_pk_add_f32 %" #n ", %" #n ", %9\n"
#define I35(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n s_nop 0\n"
#define I36(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n s_nop 1\n"
#define I37(n) "v_cmp_gt_f32 vcc, %" #n ", %8\n s_nop 1\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
#define I38(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n s_or_b64 s[20:21], s[20:21], vcc\n"
#define I39(n) "v_min_f32 %" #n ", %" #n ", %9\n"
#define I40(n) "v_max3_f32 %" #n ", %" #n ", %8, %9\n"
static const char *NAMES[] = {"v_fma_f32", "v_mul_f32_e32", "v_add_f32_e32", "v_sub_f32_e32", "v_fmac_f32_e32", "v_fmamk_f32", "v_fmaak_f32", "v_mul_f32_e64",
                              "v_add_u32_e32", "v_and_b32_e32", "v_mov_b32_e32", "v_lshlrev_b32_e32", "v_lshl_add_u32", "v_and_or_b32", "v_add_lshl_u32", "v_mul_u32_u24_e32",
                              "v_cvt_u32_f32_e32", "v_cvt_i32_f32_e32", "v_cvt_f32_i32_e32", "v_floor_f32_e32", "v_cndmask_b32_e32", "v_cndmask_b32_e64", "v_cmp_f32_e32", "v_cmp_f32_e64",
                              "v_cmp_u32_e32", "v_cmp_u32_e64", "v_cmp_class_f32_e32", "v_rcp_f32_e32", "v_sqrt_f32_e32", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32",
                              "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "pair:v_fma+s_nop0", "pair:v_fma+s_nop1", "triple:v_cmp+s_nop1+v_cndmask", "pair:v_fma+s_or_b64", "v_min_f32_e32", "v_max3_f32"};
static const int PER[] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
constexpr int NMODES = 41;

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
  float r[8];
  f2 p[8];
  for (int i = 0; i < 8; i++)
    r[i] = threadIdx.x * 1e-3f + i, p[i] = f2{r[i], r[i] + 0.5f};
  float c0 = 0.999f, c1 = 1e-3f;
  f2 pc0 = f2{0.999f, 0.998f}, pc1 = f2{1e-3f, 2e-3f};
  asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b64 vcc, 0x3333" ::: "s20", "s21", "vcc");
  for (int i = 0; i < iters; i++)
  {
#pragma unroll
    for (int u = 0; u < 16; u++)
    {
#define M(N) if (MODE == N) OP8(I##N);
      M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)
      if (MODE == 32) OP8P(I32);
      if (MODE == 33) OP8P(I33);
      if (MODE == 34) OP8P(I34);
      M(35) M(36) M(37) M(38) M(39) M(40)
#undef M
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++)
    s += r[i] + p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// LDS atomics: 64-bit adds, 8 per loop iteration per wave; SPREAD = number of distinct 8-byte addresses the 64 lanes of a wave hit
template <int SPREAD>
__global__ void __launch_bounds__(256) k_lds(float *out, int iters)
{
  __shared__ unsigned long long s_h[4][128];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 512; i += 256)
    (&s_h[0][0])[i] = 0;
  __syncthreads();
  unsigned long long *base = &s_h[wave][(lane * 37) % SPREAD];
  for (int i = 0; i < iters; i++)
  {
#pragma unroll
    for (int u = 0; u < 8; u++)
      atomicAdd(base, (unsigned long long)(i + u));
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = (float)s_h[wave][lane];
}

template <int M>
double run(float *d)
{
  const int blocks = 256 * 8, iters = 1000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0, best = 1e9f;
  for (int rep = 0; rep < 3; rep++)
  {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep)
      best = ms < best ? ms : best;
  }
  // wave-instructions (groups) per SIMD: blocks * 4 waves * iters * 16 * 8 / 1024 SIMDs
  const double per_simd = (double)blocks * 4 * iters * 16 * 8 / 1024.0;
  return best * 1e9 / per_simd; // picoseconds per group per SIMD
}
template <int S>
double run_lds(float *d)
{
  const int blocks = 256 * 8, iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0, best = 1e9f;
  for (int rep = 0; rep < 3; rep++)
  {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_lds<S>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep)
      best = ms < best ? ms : best;
  }
  const double per_cu = (double)blocks * 4 * iters * 8 / 256.0;
  return best * 1e9 / per_cu; // picoseconds per wave-atomic per CU
}

template <int M>
void all(float *d, double *o)
{
  o[M] = run<M>(d);
  fprintf(stderr, "%s %.0f\n", NAMES[M], o[M]);
  if constexpr (M + 1 < NMODES)
    all<M + 1>(d, o);
}

int main()
{
  float *d;
  (void)hipMalloc(&d, 1 << 24);
  static double o[NMODES];
  all<0>(d, o);
  printf("{\"unit\": \"ps per wave-instruction per SIMD (8 waves per SIMD, 8 chains per wave)\"");
  for (int i = 0; i < NMODES; i++)
    printf(", \"%s\": %.0f", NAMES[i], o[i] / PER[i]);
  printf(", \"lds_unit\": \"ps per 64-lane ds_add_u64 per CU (16 waves per CU issuing)\"");
  printf(", \"ds_add_u64@64\": %.0f, \"ds_add_u64@32\": %.0f, \"ds_add_u64@16\": %.0f, \"ds_add_u64@8\": %.0f, \"ds_add_u64@1\": %.0f", run_lds<64>(d), run_lds<32>(d), run_lds<16>(d), run_lds<8>(d), run_lds<1>(d));
  printf("}\n");
  return 0;
}
