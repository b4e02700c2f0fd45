// stream_patterns.hip — what does the memory system give a 1-read-1-write fp32 stream over B x 960 x 1280 planes, by ACCESS PATTERN?
// (the blur launches of pyramid.hip run at 4.4-5.5 TB/s whatever their arithmetic; torch.mul on the same buffers reaches 6.2)
//   linear      : one float4 per thread, consecutive threads consecutive addresses (the torch elementwise pattern)
//   march<TW>   : one wave per TW-column strip marching down a row segment in groups of 8 rows, register prefetch one group ahead,
//                 XCD-contiguous strip order — the blur kernels' pattern without their arithmetic. TW = 128 (b128 loads by 32 lanes,
//                 b64 stores by 64) or 256 (b128 / b128)
//   band<NW>    : NW waves of one workgroup side by side (NW x 256 columns = the image width), marching in step (one barrier per group):
//                 the workgroup reads and writes whole rows
// knobs per run: waves per SIMD (through a dynamic LDS reservation), row segments per image, nt or plain stores.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_stream_patterns tools/microbench/stream_patterns.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_linear(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n4)
{
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4)
  {
    float4 v = in[i];
    v.x *= 1.0001f, v.y *= 1.0001f, v.z *= 1.0001f, v.w *= 1.0001f;
    out[i] = v;
  }
}

struct MArgs
{
  const float *src;
  float *dst;
  int w, h, pitch, seg, rev;
  size_t img_stride;
  int panel;
  int order; // 0: strips fastest, then segments, then images (the blur kernels); 1: strips, images, segments; 2: images fastest; 3: no XCD remap
};

__device__ __forceinline__ void map_block(const MArgs &a, uint32_t &bs, uint32_t &bseg, uint32_t &bimg)
{
  const uint32_t total = gridDim.x * gridDim.y * gridDim.z;
  const uint32_t b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  uint32_t wi = b;
  if ((total & 7u) == 0)
  {
    const uint32_t per = total >> 3, k = b >> 3;
    wi = (b & 7u) * per + (a.rev ? per - 1u - k : k);
  }
  if (a.order == 3)
    wi = b;
  if (a.order == 2)
  {
    bimg = wi % gridDim.z;
    const uint32_t r = wi / gridDim.z;
    bs = r % gridDim.x;
    bseg = r / gridDim.x;
    return;
  }
  bs = wi % gridDim.x;
  const uint32_t r = wi / gridDim.x;
  if (a.order == 1)
  {
    bimg = r % gridDim.z;
    bseg = r / gridDim.z;
    return;
  }
  bseg = r % gridDim.y;
  bimg = r / gridDim.y;
  if (a.order == 5 && (total & 7u) == 0 && (gridDim.z & 7u) == 0)
  {
    // XCD x (= b & 7) still gets a contiguous run of work items, but its images are x, x + 8, x + 16, ...: every XCD sweeps the WHOLE
    // batch range at the same pace instead of an eighth of it each
    const uint32_t per_img = gridDim.z >> 3;          // images per XCD
    const uint32_t x = bimg / per_img, k = bimg % per_img;
    bimg = k * 8u + x;
  }
}

template <int TW, int NT_ST, int NWAVES>
__global__ void __launch_bounds__(64 * NWAVES) k_march(MArgs a)
{
  extern __shared__ float s_dummy[];
  constexpr int NR = 8;
  uint32_t bs, bseg, bimg;
  map_block(a, bs, bseg, bimg);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x0 = (bs * NWAVES + wave) * TW;
  const int y0 = bseg * a.seg, y1 = min(y0 + a.seg, a.h);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(a.src + bimg * a.img_stride), 0, a.pitch * a.h * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)(a.dst + bimg * a.img_stride), 0, a.pitch * a.h * 4, 0x00020000);
  int p4 = a.pitch * 4;
  if (TW == 256)
  {
    unsigned off = x0 + 4 * lane + 3 < a.w ? (unsigned)(x0 + 4 * lane) * 4u : 0x80000000u;
    if (a.panel)
    {
      // panel-major plane: the 256-column panel of a strip is contiguous (row pitch 256 floats), panels follow each other
      p4 = 256 * 4;
      off = (unsigned)((size_t)(x0 / 256) * a.h * 256 * 4) + (unsigned)(4 * lane) * 4u;
    }
    u32x4 pf[NR];
    if (a.panel == 2)
    {
      // store only (the seed launch's traffic: 4 B written per texel, next to nothing read)
      for (int y = y0; y < y1; y += NR)
      {
#pragma unroll
        for (int j = 0; j < NR; j++)
        {
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{(unsigned)y, (unsigned)j, 2u, 3u}, rd, off, (y + j) * p4, NT_ST);
          asm volatile("s_nop 1");
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < NR; j++)
      pf[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, (y0 + j) * p4, 0);
    for (int y = y0; y < y1; y += NR)
    {
      u32x4 cur[NR];
#pragma unroll
      for (int j = 0; j < NR; j++)
        cur[j] = pf[j];
      if (y + NR < y1)
      {
#pragma unroll
        for (int j = 0; j < NR; j++)
          pf[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, (y + NR + j) * p4, 0);
      }
      if (NWAVES > 1)
        __syncthreads();
#pragma unroll
      for (int j = 0; j < NR; j++)
      {
        u32x4 v = cur[j];
        v.x ^= 1u;
        __builtin_amdgcn_raw_buffer_store_b128(v, rd, off, (y + j) * p4, NT_ST);
        asm volatile("s_nop 1");
      }
    }
  }
  else
  {
    const unsigned ldo = (lane < 32 && x0 + 4 * lane + 3 < a.w) ? (unsigned)(x0 + 4 * lane) * 4u : 0x80000000u;
    const unsigned sto = x0 + 2 * lane + 1 < a.w ? (unsigned)(x0 + 2 * lane) * 4u : 0x80000000u;
    u32x4 pf[NR];
#pragma unroll
    for (int j = 0; j < NR; j++)
      pf[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, ldo, (y0 + j) * p4, 0);
    for (int y = y0; y < y1; y += NR)
    {
      u32x4 cur[NR];
#pragma unroll
      for (int j = 0; j < NR; j++)
        cur[j] = pf[j];
      if (y + NR < y1)
      {
#pragma unroll
        for (int j = 0; j < NR; j++)
          pf[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, ldo, (y + NR + j) * p4, 0);
      }
#pragma unroll
      for (int j = 0; j < NR; j++)
      {
        // lanes 0..31 hold the row: lane l stores texels 2l, 2l+1 = half of lane l/2's float4
        const int srcl = lane >> 1;
        const unsigned a0 = __shfl(cur[j].x, srcl), a1 = __shfl(cur[j].y, srcl), a2 = __shfl(cur[j].z, srcl), a3 = __shfl(cur[j].w, srcl);
        const u32x2 v = (lane & 1) ? u32x2{a2, a3} : u32x2{a0, a1};
        __builtin_amdgcn_raw_buffer_store_b64(v, rd, sto, (y + j) * p4, NT_ST);
      }
    }
  }
}

static float median(std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char **argv)
{
  const int B = argc > 1 ? atoi(argv[1]) : 512, H = 960, W = 1280, REPS = 16;
  const size_t n = (size_t)B * H * W;
  float *src, *dst;
  CK(hipMalloc(&src, n * 4));
  CK(hipMalloc(&dst, n * 4));
  CK(hipMemset(src, 1, n * 4));
  CK(hipMemset(dst, 0, n * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double bytes = 8.0 * n;
  auto report = [&](const char *name, std::vector<float> &ms) {
    const float m = median(ms), mn = *std::min_element(ms.begin(), ms.end());
    printf("%-46s %8.1f us (min %8.1f)  %5.0f GB/s  frac %.3f\n", name, m * 1e3, mn * 1e3, bytes / (m * 1e-3) / 1e9, bytes / (m * 1e-3) / 8e12);
    fflush(stdout);
  };
  {
    std::vector<float> ms;
    for (int r = 0; r < REPS; r++)
    {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_linear, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, (const float4 *)src, (float4 *)dst, n / 4);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      if (r >= 2)
        ms.push_back(t);
    }
    report("linear float4, 256 threads, one per thread", ms);
  }
  // production geometry: ONE allocation, image i at i * IS floats, the source plane at offset 0 and the destination plane PS floats behind it
  // (consecutive planes of the same image, as a blur launch of the pyramid sees them). argv: B IS PS [sweep]; IS = 0: two dense buffers
  const size_t IS = argc > 2 ? (size_t)atoll(argv[2]) : 0, PS = argc > 3 ? (size_t)atoll(argv[3]) : (size_t)H * W;
  const size_t pad_img = IS;
  MArgs a{src, dst, W, H, W, 0, 0, (size_t)H * W, 0, 0};
  float *big = nullptr;
  if (IS)
  {
    CK(hipFree(src));
    CK(hipFree(dst));
    CK(hipMalloc(&big, (IS * B + PS + (size_t)H * W) * 4));
    CK(hipMemset(big, 1, (IS * B + PS + (size_t)H * W) * 4));
    a.src = big, a.dst = big + PS, a.img_stride = IS;
  }
  auto run = [&](const char *label, auto kern, int tw, int nwaves, int nseg, int wps, int order) {
    a.seg = ((H + nseg - 1) / nseg + 7) & ~7;
    a.order = order;
    const int strips = (W + tw * nwaves - 1) / (tw * nwaves);
    dim3 grid(strips, (H + a.seg - 1) / a.seg, B);
    const int wg_per_cu = std::max(1, wps * 4 / nwaves);
    size_t lds = (160 * 1024) / wg_per_cu;
    lds = lds > 4096 ? lds - 1024 : lds;
    if (wps >= 8)
      lds = 0;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
    std::vector<float> ms;
    for (int r = 0; r < REPS; r++)
    {
      a.rev = r & 1;
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, grid, dim3(64 * nwaves), lds, 0, a);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      if (r >= 2)
        ms.push_back(t);
    }
    char name[128];
    snprintf(name, sizeof name, "%s nseg=%d order=%d IS=%zu PS=%zu", label, nseg, order, pad_img, PS);
    report(name, ms);
  };
  if (argc > 4)
  {
    if (argc > 5)
    {
      // ONE arena of 2.2x the span, the planes at sliding offsets: does the rate depend on the offset, and with what period?
      const size_t span = IS * B + PS + (size_t)H * W;
      const size_t arena = (size_t)atoll(argv[5]) << 28; // argv[5]: arena size in GiB (floats: GiB << 28)
      float *ar;
      CK(hipMalloc(&ar, arena * 4));
      CK(hipMemset(ar, 1, arena * 4));
      printf("arena at %p, %.1f GB\n", (void *)ar, arena * 4 / 1e9);
      for (size_t off = (size_t)20 << 30; off + span <= arena; off += (size_t)8 << 30) // from 80 GiB on, 32 GiB steps
      {
        a.panel = 2;
        a.src = ar + off, a.dst = ar + off + PS;
        for (int ns : {2, 4, 8})
          for (int wps : {2, 4, 8})
            run("STORE-ONLY march256 (x0.5 for GB/s)", k_march<256, 2, 1>, 256, 1, ns, wps, 0);
        a.panel = 0;
      }
      for (size_t off = (size_t)20 << 30; off + span <= arena; off += (size_t)8 << 30)
      {
        a.src = ar + off, a.dst = ar + off + PS;
        char nm[64];
        snprintf(nm, sizeof nm, "offset %6.0f MB march256", off * 4 / 1048576.0);
        run(nm, k_march<256, 2, 1>, 256, 1, 2, 3, 0);
      }
      return 0;
    }
    // several allocations in one process, tested round-robin three times: does the rate depend on WHICH memory the planes got
    // (sticks to the allocation) or on the chip's state at the time (moves around)?
    std::vector<float *> keep;
    for (int t = 0; t < 5; t++)
    {
      float *b2;
      CK(hipMalloc(&b2, (IS * B + PS + (size_t)H * W) * 4));
      CK(hipMemset(b2, 1, (IS * B + PS + (size_t)H * W) * 4));
      keep.push_back(b2);
      printf("allocation %d at %p\n", t, (void *)b2);
    }
    for (int round = 0; round < 3; round++)
      for (int t = 0; t < 5; t++)
      {
        float *b2 = keep[t];
        a.src = b2, a.dst = b2 + PS;
        char nm[64];
        snprintf(nm, sizeof nm, "round %d alloc %d march256", round, t);
        run(nm, k_march<256, 2, 1>, 256, 1, 2, 3, 0);
        const size_t half = (IS * B / 2) & ~(size_t)1023;
        std::vector<float> ms;
        for (int r = 0; r < REPS; r++)
        {
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(k_linear, dim3((unsigned)((half / 4 + 255) / 256)), dim3(256), 0, 0, (const float4 *)b2, (float4 *)(b2 + half), half / 4);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float tt;
          CK(hipEventElapsedTime(&tt, e0, e1));
          if (r >= 2)
            ms.push_back(tt);
        }
        printf("   linear copy inside this allocation: %.0f GB/s\n", 8.0 * half / (median(ms) * 1e-3) / 1e9);
        if (round == 0)
        {
          // the same allocation under other geometries: another plane pair of the image, a square plane, half the batch
          struct G { const char *nm; int w, h; size_t so, dofs; int b; } gs[] = {
            {"planes 2->3 1280x960", 1280, 960, 2 * PS, 3 * PS, B},
            {"planes 4->5 1280x960", 1280, 960, 4 * PS, 5 * PS, B},
            {"planes 0->1 1152x1110 (pitch 1152)", 1152, 1110, 0, (size_t)1152 * 1110, B},
            {"planes 0->1 1280x960, first half of the batch", 1280, 960, 0, PS, B / 2},
            {"planes 0->1 1280x960, second half of the batch", 1280, 960, (size_t)(B / 2) * IS, (size_t)(B / 2) * IS + PS, B / 2},
          };
          for (const G &g : gs)
          {
            MArgs a2 = a;
            a2.src = b2 + g.so, a2.dst = b2 + g.dofs, a2.w = g.w, a2.h = g.h, a2.pitch = g.w;
            a2.seg = ((g.h + 2 - 1) / 2 + 7) & ~7;
            a2.order = 0;
            dim3 grid((g.w + 255) / 256, (g.h + a2.seg - 1) / a2.seg, g.b);
            std::vector<float> m2;
            for (int r = 0; r < REPS; r++)
            {
              a2.rev = r & 1;
              CK(hipEventRecord(e0));
              hipLaunchKernelGGL((k_march<256, 2, 1>), grid, dim3(64), (size_t)12 * 1024, 0, a2);
              CK(hipEventRecord(e1));
              CK(hipEventSynchronize(e1));
              float tt;
              CK(hipEventElapsedTime(&tt, e0, e1));
              if (r >= 2)
                m2.push_back(tt);
            }
            printf("      %-50s %.0f GB/s\n", g.nm, 8.0 * g.b * g.h * g.w / (median(m2) * 1e-3) / 1e9);
          }
        }
        if (false)
          for (int q = 0; q < 8; q++)
          {
            // eighths of the batch: is the rate a property of the whole allocation or of parts of it?
            a.src = b2 + (size_t)q * (B / 8) * IS, a.dst = b2 + (size_t)q * (B / 8) * IS + PS;
            a.seg = ((H + 2 - 1) / 2 + 7) & ~7;
            a.order = 0;
            dim3 grid((W + 255) / 256, (H + a.seg - 1) / a.seg, B / 8);
            std::vector<float> m2;
            for (int r = 0; r < REPS; r++)
            {
              a.rev = r & 1;
              CK(hipEventRecord(e0));
              hipLaunchKernelGGL((k_march<256, 2, 1>), grid, dim3(64), (size_t)12 * 1024, 0, a);
              CK(hipEventRecord(e1));
              CK(hipEventSynchronize(e1));
              float tt;
              CK(hipEventElapsedTime(&tt, e0, e1));
              if (r >= 2)
                m2.push_back(tt);
            }
            printf("      images %3d..%3d: %.0f GB/s\n", q * (B / 8), (q + 1) * (B / 8) - 1, 8.0 * (B / 8) * H * W / (median(m2) * 1e-3) / 1e9);
          }
      }
    return 0;
  }
  for (int order : {0, 1, 2, 3})
    for (int nseg : {1, 2, 4, 8, 15})
    {
      run("march128 nt", k_march<128, 2, 1>, 128, 1, nseg, 3, order);
      run("march256 nt", k_march<256, 2, 1>, 256, 1, nseg, 3, order);
    }
  return 0;
}
