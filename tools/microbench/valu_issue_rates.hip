#include <hip/hip_runtime.h>
#include <cstdio>
#define OP8(INSTR) asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(c0), "v"(c1))
#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I_ADD(n) "v_add_f32 %" #n ", %" #n ", %9\n"
#define I_ADDU(n) "v_add_u32 %" #n ", %" #n ", %9\n"
#define I_CVTU(n) "v_cvt_u32_f32 %" #n ", %" #n "\n"
#define I_CVTF(n) "v_cvt_f32_i32 %" #n ", %" #n "\n"
#define I_FLOOR(n) "v_floor_f32 %" #n ", %" #n "\n"
#define I_CND(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_CMP(n) "v_cmp_gt_f32 vcc, %" #n ", %8\n"
#define I_LSHL(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define I_AND(n) "v_and_b32 %" #n ", %" #n ", %9\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define I_SQRT(n) "v_sqrt_f32 %" #n ", %" #n "\n"
#define I_MAX3(n) "v_max3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_CNDS(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\n"
#define I_CND0(n) "v_cndmask_b32 %" #n ", %8, %9, vcc\n"
#define I_PKMUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I_PAIRV(n) "v_cmp_gt_f32 vcc, %" #n ", %8\n s_nop 1\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
#define I_PAIRS(n) "v_cmp_gt_f32 s[20:21], %" #n ", %8\n s_nop 1\n v_cndmask_b32_e64 %" #n ", %" #n ", %9, s[20:21]\n"
#define I_PAIRV2(n) "v_cmp_gt_f32 vcc, %" #n ", %8\n v_add_f32 %" #n ", %" #n ", %9\n v_mul_f32 %" #n ", %" #n ", %8\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
#define I_PAIRS2(n) "v_cmp_gt_f32 s[20:21], %" #n ", %8\n v_add_f32 %" #n ", %" #n ", %9\n v_mul_f32 %" #n ", %" #n ", %8\n v_cndmask_b32_e64 %" #n ", %" #n ", %9, s[20:21]\n"
#define I_PKADD(n) "v_add_f32 %" #n ", %" #n ", %8\n"
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
  float r[8];
  for (int i = 0; i < 8; i++) r[i] = threadIdx.x * 1e-3f + i;
  float c0 = 0.999f, c1 = 1e-3f;
  asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b64 vcc, 0x3333" ::: "s20", "s21", "vcc");
  for (int i = 0; i < iters; i++)
  {
#pragma unroll
    for (int u = 0; u < 16; u++)
    {
      if (MODE == 0) OP8(I_FMA);
      if (MODE == 1) OP8(I_MUL);
      if (MODE == 2) OP8(I_ADD);
      if (MODE == 3) OP8(I_ADDU);
      if (MODE == 4) OP8(I_CVTU);
      if (MODE == 5) OP8(I_CVTF);
      if (MODE == 6) OP8(I_FLOOR);
      if (MODE == 7) OP8(I_CND);
      if (MODE == 8) OP8(I_CMP);
      if (MODE == 9) OP8(I_LSHL);
      if (MODE == 10) OP8(I_AND);
      if (MODE == 11) OP8(I_MOV);
      if (MODE == 12) OP8(I_RCP);
      if (MODE == 13) OP8(I_SQRT);
      if (MODE == 14) OP8(I_MAX3);
      if (MODE == 15) OP8(I_CNDS);
      if (MODE == 17) OP8(I_PAIRV);
      if (MODE == 18) OP8(I_PAIRS);
      if (MODE == 19) OP8(I_PAIRV2);
      if (MODE == 20) OP8(I_PAIRS2);
      if (MODE == 16) OP8(I_CND0);
    }
  }
  float s = 0; for (int i = 0; i < 8; i++) s += r[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int M> void run(float *d, const char *name)
{
  const int blocks = 256 * 8, iters = 1000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; rep++)
  {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double insts = (double)blocks * 4 * iters * 16 * 8;
  printf("%-14s %.3f ms  ratio to 4-cycle model: %.2f cycles/instr/SIMD at 2.4 GHz\n", name, ms, 2400.0 / (insts / ms / 1e3 / 1024));
}
int main()
{
  float *d; (void)hipMalloc(&d, 1 << 24);
  run<0>(d, "v_fma_f32"); run<1>(d, "v_mul_f32"); run<2>(d, "v_add_f32"); run<3>(d, "v_add_u32"); run<4>(d, "v_cvt_u32_f32"); run<5>(d, "v_cvt_f32_i32");
  run<6>(d, "v_floor_f32"); run<7>(d, "v_cndmask_b32"); run<8>(d, "v_cmp_gt_f32"); run<9>(d, "v_lshlrev_b32"); run<10>(d, "v_and_b32"); run<11>(d, "v_mov_b32");
  run<12>(d, "v_rcp_f32"); run<13>(d, "v_sqrt_f32"); run<14>(d, "v_max3_f32"); run<15>(d, "cndmask sgpr"); run<16>(d, "cndmask nodep"); run<17>(d, "cmp+cnd vcc"); run<18>(d, "cmp+cnd sgpr"); run<19>(d, "cmp,add,mul,cnd vcc"); run<20>(d, "cmp,add,mul,cnd sgpr");
  return 0;
}
