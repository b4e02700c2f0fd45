// pyramid_fused.hip — the whole scale chain of one octave in one launch (gfx950, wave64).
//
// Replaces, for the default configuration (3 scales per octave, seed sigma 1.6: one-sided tap counts 5,7,9,11,13),
// the five GaussianBlur*.comp H+V dispatch pairs + five DifferenceOfGaussian.comp dispatches of one octave
// (sift_detector.c:927-1001, 1039-1079) and the NEAREST down-sampling blit that seeds the next octave
// (sift_detector.c:1003-1034). The per-scale kernels of pyramid.hip stay the generic path for every other tap set.
//
// Why: the per-scale kernels move 12 B per pixel and scale (read G[s-1], write G[s], write DoG[s-1]); chained in one
// kernel every intermediate scale is consumed from LDS, so an octave pixel costs 4 B read + 40 B written instead of
// 60 B — and one launch instead of six.
//
// Structure: a workgroup = 6 waves = 3 wave pairs owns a strip of TW = 160 output columns (five 128-byte lines) of a
// row segment and marches down it in steps of NR rows. Scale k (1..5, radius R_k = 4,6,8,10,12) is a pipeline stage
// that lives in one wave pair: pair A runs stages 1+5, pair B stages 2+4, pair C stage 3 + the global loads of G0
// (25+9, 21+13, 17 taps: balanced). A stage covers its own output columns plus the halo the later stages still need
// (h_k = sum of the later radii), 2 adjacent pixels per lane:
//   H phase : the NR newest rows of G[k-1] are read from the stage's LDS input ring (ds_read_b64 window of 2R+2
//             floats per lane), blurred horizontally and pushed into a register window of 2R+NR rows
//   V phase : NR output rows from the register window -> global G[k] and DoG[k-1] (own strip columns only) and ->
//             the input ring of stage k+1 (all columns)
// One barrier after each phase; stage k+1 consumes in step t+1 what stage k produced in step t. The input ring of
// stage k is R_k+NR rows deep so that it doubles as the delay line of the DoG centres.
//
// Borders: rows and columns outside the image are *virtual*: G0 is loaded with mirrored-repeat indices and every
// stage simply computes on. Because the taps are symmetric and the pass is evaluated as
// fmaf(t(+i) + t(-i), k[i], acc), the value computed at a virtual position is bit-identical to the value at its
// mirror image (the two operands of each pair sum swap, IEEE addition commutes) — which is exactly what the
// reference's MIRRORED_REPEAT sampler feeds the next scale with. No border special case anywhere.
//
// Arithmetic contract: identical to pyramid.hip / oracle blur_plane (see there), compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "vksift_hip.h"

namespace
{

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int mirror_idx(int i, int n)
{
  if ((unsigned)i < (unsigned)n)
    return i;
  int period = 2 * n;
  int j = i % period;
  if (j < 0)
    j += period;
  return j < n ? j : period - 1 - j;
}

__device__ __forceinline__ int pmod(int v, int m) { return (v + m * 4096) % m; } // v > -4096*m

constexpr int TW = 160;
constexpr int NT1 = 5, NT2 = 7, NT3 = 9, NT4 = 11, NT5 = 13;
constexpr int R1 = NT1 - 1, R2 = NT2 - 1, R3 = NT3 - 1, R4 = NT4 - 1, R5 = NT5 - 1;
// halo still needed after stage k, and the stage widths
constexpr int H5 = 0, H4 = R5, H3 = H4 + R4, H2 = H3 + R3, H1 = H2 + R2, H0 = H1 + R1;
constexpr int W0 = TW + 2 * H0, W1 = TW + 2 * H1, W2 = TW + 2 * H2, W3 = TW + 2 * H3, W4 = TW + 2 * H4, W5 = TW;
static_assert(W0 <= 256 && (W0 % 4) == 0, "stage widths must fit one wave pair at 2 px per lane");
static_assert(((R1 | R2 | R3 | R4 | R5) & 1) == 0, "even radii keep every float2 access 8-byte aligned");

struct FusedArgs
{
  float *g0;            // Gaussian plane 0 of this octave, image 0 (planes k = 1..5 follow at k * plane_stride)
  float *dog0;          // DoG plane 0
  float *next_g0;       // Gaussian plane 0 of the next octave (exact 2:1 sizes only) or null
  uint64_t plane_stride, img_stride, next_img_stride; // floats
  int pitch, next_pitch, next_w, next_h;
  int w, h, seg;
  float k1[NT1], k2[NT2], k3[NT3], k4[NT4], k5[NT5];
};

// Compile-time description of pipeline stage K (scale K of the octave)
template <int K>
struct Stage;
template <>
struct Stage<1>
{
  static constexpr int NT = NT1, HALO = H1, PRE = 0, WIN = W0, WOUT = W1;
};
template <>
struct Stage<2>
{
  static constexpr int NT = NT2, HALO = H2, PRE = H0 - H1, WIN = W1, WOUT = W2;
};
template <>
struct Stage<3>
{
  static constexpr int NT = NT3, HALO = H3, PRE = H0 - H2, WIN = W2, WOUT = W3;
};
template <>
struct Stage<4>
{
  static constexpr int NT = NT4, HALO = H4, PRE = H0 - H3, WIN = W3, WOUT = W4;
};
template <>
struct Stage<5>
{
  static constexpr int NT = NT5, HALO = H5, PRE = H0 - H4, WIN = W4, WOUT = W5;
};

struct Ctx
{
  float *g, *dog, *nxt;
  uint64_t plane_stride;
  int pitch, next_pitch, next_w, next_h;
  int W, x0, y0, y1, vstart, c;
};

// stage K consumes in step t the rows in(t) .. +NR-1 of scale K-1 and emits rows in(t) - R .. +NR-1 of scale K:
//   in(t) = vstart + NR*(t-K) - PRE, valid from vstart + PRE, needed up to y1-1 + HALO + R
template <int K, int NR>
__device__ __forceinline__ bool stage_active(const Ctx &cx, int t, int &in_row)
{
  using S = Stage<K>;
  in_row = cx.vstart + NR * (t - K) - S::PRE;
  return cx.c < S::WOUT && in_row + NR - 1 >= cx.vstart + S::PRE && in_row <= cx.y1 - 1 + S::HALO + (S::NT - 1);
}

// ---- H phase: NR new input rows -> register window; also fetch the DoG centres of the rows the V phase will emit
template <int K, int NR>
__device__ __forceinline__ void stage_h(const Ctx &cx, int t, float2 *__restrict__ wv, float2 *__restrict__ ctr, const float *__restrict__ ring,
                                        const float *__restrict__ k)
{
  using S = Stage<K>;
  constexpr int NT = S::NT, R = NT - 1, D = R + NR, WIN = S::WIN;
  int in_row;
  if (!stage_active<K, NR>(cx, t, in_row))
    return;
  const int c = cx.c;
  int slot = pmod(in_row, D);
  const float k0 = k[0];
#pragma unroll
  for (int j = 0; j < NR; j++)
  {
    const v2f *p = (const v2f *)(ring + slot * WIN + c);
    float v[2 * R + 2];
#pragma unroll
    for (int q = 0; q < R + 1; q++)
    {
      v2f tq = p[q];
      v[2 * q] = tq.x, v[2 * q + 1] = tq.y;
    }
    float acc0 = v[R] * k0, acc1 = v[R + 1] * k0;
#pragma unroll
    for (int i = 1; i < NT; i++)
    {
      acc0 = fmaf(v[R + i] + v[R - i], k[i], acc0);
      acc1 = fmaf(v[R + 1 + i] + v[R + 1 - i], k[i], acc1);
    }
    wv[2 * R + j] = make_float2(acc0, acc1);
    slot = slot + 1 == D ? 0 : slot + 1;
    __builtin_amdgcn_sched_barrier(0);
  }
  slot = pmod(in_row - R, D);
#pragma unroll
  for (int j = 0; j < NR; j++)
  {
    const v2f tq = *(const v2f *)(ring + slot * WIN + c + R);
    ctr[j] = make_float2(tq.x, tq.y);
    slot = slot + 1 == D ? 0 : slot + 1;
  }
}

// ---- V phase: NR output rows from the window -> next stage's ring (all columns) + global G[K], DoG[K-1] (own columns), window slides
template <int K, int NR>
__device__ __forceinline__ void stage_v(const Ctx &cx, int t, float2 *__restrict__ wv, const float2 *__restrict__ ctr, float *__restrict__ ring_out,
                                        const float *__restrict__ k)
{
  using S = Stage<K>;
  constexpr int NT = S::NT, R = NT - 1;
  constexpr int DN = Stage<(K < 5 ? K + 1 : 5)>::NT - 1 + NR; // depth of the next stage's ring
  int in_row;
  if (!stage_active<K, NR>(cx, t, in_row))
    return;
  const int c = cx.c;
  const int out_row = in_row - R;
  if (out_row + NR - 1 >= cx.vstart + S::PRE + R)
  {
    const float k0 = k[0];
    const int x = cx.x0 - S::HALO + c;
    const bool own = c >= S::HALO && c < S::HALO + TW;
    float *gk = cx.g + K * cx.plane_stride;
    float *dk = cx.dog + (K - 1) * cx.plane_stride;
    int slot = K < 5 ? pmod(out_row, DN) : 0;
    size_t orow = (size_t)out_row * cx.pitch + x;
#pragma unroll
    for (int j = 0; j < NR; j++)
    {
      float acc0 = wv[R + j].x * k0, acc1 = wv[R + j].y * k0;
#pragma unroll
      for (int i = 1; i < NT; i++)
      {
        acc0 = fmaf(wv[R + j + i].x + wv[R + j - i].x, k[i], acc0);
        acc1 = fmaf(wv[R + j + i].y + wv[R + j - i].y, k[i], acc1);
      }
      const int y = out_row + j;
      if (K < 5)
      {
        *(v2f *)(ring_out + slot * S::WOUT + c) = v2f{acc0, acc1};
        slot = slot + 1 == DN ? 0 : slot + 1;
      }
      if (own && y >= cx.y0 && y < cx.y1)
      {
        if (x + 1 < cx.W)
        {
          *(float2 *)(gk + orow) = make_float2(acc0, acc1);
          *(float2 *)(dk + orow) = make_float2(acc0 - ctr[j].x, acc1 - ctr[j].y);
        }
        else if (x < cx.W)
        {
          gk[orow] = acc0;
          dk[orow] = acc0 - ctr[j].x;
        }
        // NEAREST 2:1 blit into the next octave: destination (x', y') takes source texel (2x'+1, 2y'+1)
        if (K == 3 && cx.nxt && (y & 1) && (y >> 1) < cx.next_h && (x >> 1) < cx.next_w)
          cx.nxt[(size_t)(y >> 1) * cx.next_pitch + (x >> 1)] = acc1;
      }
      orow += cx.pitch;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int q = 0; q < 2 * R; q++)
    wv[q] = wv[q + NR];
}

template <int NR>
__global__ void __launch_bounds__(384, 3) k_octave_fused(FusedArgs a)
{
  constexpr int D1 = R1 + NR, D2 = R2 + NR, D3 = R3 + NR, D4 = R4 + NR, D5 = R5 + NR;
  __shared__ __attribute__((aligned(16))) float s_in1[D1 * W0]; // G0 rows (loaded from HBM)
  __shared__ __attribute__((aligned(16))) float s_in2[D2 * W1]; // G1
  __shared__ __attribute__((aligned(16))) float s_in3[D3 * W2]; // G2
  __shared__ __attribute__((aligned(16))) float s_in4[D4 * W3]; // G3
  __shared__ __attribute__((aligned(16))) float s_in5[D5 * W4]; // G4

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform by construction: keep the role branches scalar
  const int pair = wave >> 1, hw = wave & 1;

  Ctx cx;
  cx.W = a.w;
  cx.x0 = blockIdx.x * TW;
  cx.y0 = blockIdx.y * a.seg;
  cx.y1 = min(cx.y0 + a.seg, a.h);
  cx.g = a.g0 + (size_t)blockIdx.z * a.img_stride;
  cx.dog = a.dog0 + (size_t)blockIdx.z * a.img_stride;
  cx.nxt = a.next_g0 ? a.next_g0 + (size_t)blockIdx.z * a.next_img_stride : nullptr;
  cx.plane_stride = a.plane_stride;
  cx.pitch = a.pitch, cx.next_pitch = a.next_pitch, cx.next_w = a.next_w, cx.next_h = a.next_h;
  cx.vstart = cx.y0 - H0;
  cx.c = 2 * (64 * hw + lane); // this lane's first column in its stages' local coordinates
  const int H = a.h;

  // last step: stage 5 emits rows y0 - 2*H0 + NR*(t-5) ..
  const int nsteps = (cx.y1 - cx.y0 + 2 * H0 + NR - 1) / NR + 5;

  // Every wave pair runs its own copy of the step loop (same trip count, two barriers per step): the register windows
  // of the other pairs' stages are not live in it.
  if (pair == 0)
  {
    float2 wx[2 * R1 + NR], wy[2 * R5 + NR], c1[NR], c5[NR];
#pragma unroll
    for (int q = 0; q < 2 * R1 + NR; q++)
      wx[q] = make_float2(0.f, 0.f);
#pragma unroll
    for (int q = 0; q < 2 * R5 + NR; q++)
      wy[q] = make_float2(0.f, 0.f);
    for (int t = 0; t < nsteps; t++)
    {
      stage_h<1, NR>(cx, t, wx, c1, s_in1, a.k1);
      stage_h<5, NR>(cx, t, wy, c5, s_in5, a.k5);
      __syncthreads();
      stage_v<1, NR>(cx, t, wx, c1, s_in2, a.k1);
      stage_v<5, NR>(cx, t, wy, c5, nullptr, a.k5);
      __syncthreads();
    }
  }
  else if (pair == 1)
  {
    float2 wx[2 * R2 + NR], wy[2 * R4 + NR], c2[NR], c4[NR];
#pragma unroll
    for (int q = 0; q < 2 * R2 + NR; q++)
      wx[q] = make_float2(0.f, 0.f);
#pragma unroll
    for (int q = 0; q < 2 * R4 + NR; q++)
      wy[q] = make_float2(0.f, 0.f);
    for (int t = 0; t < nsteps; t++)
    {
      stage_h<2, NR>(cx, t, wx, c2, s_in2, a.k2);
      stage_h<4, NR>(cx, t, wy, c4, s_in4, a.k4);
      __syncthreads();
      stage_v<2, NR>(cx, t, wx, c2, s_in3, a.k2);
      stage_v<4, NR>(cx, t, wy, c4, s_in5, a.k4);
      __syncthreads();
    }
  }
  else
  {
    float2 wx[2 * R3 + NR], c3[NR];
#pragma unroll
    for (int q = 0; q < 2 * R3 + NR; q++)
      wx[q] = make_float2(0.f, 0.f);
    // loader: G0 rows vstart + NR*t + j, columns x0-H0 .. x0+TW+H0, one float4 per lane and row, rows split over the two waves
    const int last_in = cx.y1 - 1 + H0;
    const int gx4 = cx.x0 - H0 + 4 * lane;
    const bool loader = lane < W0 / 4;
    const bool vec_ok = gx4 >= 0 && gx4 + 3 < cx.W;
    float4 pf[NR / 2];
    auto load_group = [&](int t) {
#pragma unroll
      for (int q = 0; q < NR / 2; q++)
      {
        const int r = cx.vstart + NR * t + 2 * q + hw;
        pf[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (loader && r <= last_in)
        {
          const float *row = cx.g + (size_t)mirror_idx(r, H) * cx.pitch;
          if (vec_ok)
            pf[q] = *(const float4 *)(row + gx4);
          else
            pf[q] = make_float4(row[mirror_idx(gx4, cx.W)], row[mirror_idx(gx4 + 1, cx.W)], row[mirror_idx(gx4 + 2, cx.W)],
                                row[mirror_idx(gx4 + 3, cx.W)]);
        }
      }
    };
    load_group(0);
    for (int t = 0; t < nsteps; t++)
    {
      stage_h<3, NR>(cx, t, wx, c3, s_in3, a.k3);
      __syncthreads();
      stage_v<3, NR>(cx, t, wx, c3, s_in4, a.k3);
      // hand the prefetched group t to stage 1 (rows vstart + NR*t ..), then prefetch group t+1
      if (loader)
      {
#pragma unroll
        for (int q = 0; q < NR / 2; q++)
        {
          const int r = cx.vstart + NR * t + 2 * q + hw;
          *(v4f *)(s_in1 + pmod(r, D1) * W0 + 4 * lane) = v4f{pf[q].x, pf[q].y, pf[q].z, pf[q].w};
        }
      }
      load_group(t + 1);
      __syncthreads();
    }
  }
}

} // namespace

extern "C"
{
  int vksift_hip_octave_chain_supported(const uint32_t *ntaps, uint32_t nb_scales)
  {
    /* ntaps[s] = one-sided tap count of the blur taking scale s-1 to s, s = 1 .. nb_scales+2 */
    static const uint32_t want[5] = {NT1, NT2, NT3, NT4, NT5};
    if (nb_scales != 3)
      return 0;
    for (int s = 0; s < 5; s++)
      if (ntaps[1 + s] != want[s])
        return 0;
    return 1;
  }

  int vksift_hip_octave_chain(vksift_hip_Plane g0, uint64_t plane_stride, float *dog0, vksift_hip_Plane next_g0, const float *taps, uint32_t taps_stride,
                              uint32_t batch, vksift_hip_stream s)
  {
    FusedArgs a;
    a.g0 = g0.base, a.dog0 = dog0;
    a.plane_stride = plane_stride, a.img_stride = g0.img_stride;
    a.pitch = (int)g0.pitch, a.w = (int)g0.w, a.h = (int)g0.h;
    a.next_g0 = nullptr, a.next_img_stride = 0, a.next_pitch = 0, a.next_w = 0, a.next_h = 0;
    if (next_g0.base)
    {
      if (next_g0.w * 2 != g0.w || next_g0.h * 2 != g0.h)
        return (int)hipErrorInvalidValue; /* only the exact 2:1 blit maps to odd texels; callers use vksift_hip_downsample otherwise */
      a.next_g0 = next_g0.base, a.next_img_stride = next_g0.img_stride, a.next_pitch = (int)next_g0.pitch;
      a.next_w = (int)next_g0.w, a.next_h = (int)next_g0.h;
    }
    for (int i = 0; i < NT1; i++)
      a.k1[i] = taps[1 * taps_stride + i];
    for (int i = 0; i < NT2; i++)
      a.k2[i] = taps[2 * taps_stride + i];
    for (int i = 0; i < NT3; i++)
      a.k3[i] = taps[3 * taps_stride + i];
    for (int i = 0; i < NT4; i++)
      a.k4[i] = taps[4 * taps_stride + i];
    for (int i = 0; i < NT5; i++)
      a.k5[i] = taps[5 * taps_stride + i];

    static int nr = -1, wg_target = -1;
    if (nr < 0)
    {
      const char *e = getenv("VKSIFT_CHAIN_ROWS"); /* rows per pipeline step: 4 or 8 (A/B runs) */
      nr = (e && atoi(e) == 4) ? 4 : 8;
      const char *f = getenv("VKSIFT_CHAIN_WGS");
      wg_target = (f && atoi(f) > 0) ? atoi(f) : 512;
    }
    /* row segments: ~2 workgroups per CU; every segment re-computes up to 2*H0 halo rows in its early stages, keep them long */
    const uint32_t strips = (g0.w + TW - 1) / TW;
    uint32_t nseg = ((uint32_t)wg_target + strips * batch - 1u) / (strips * batch);
    uint32_t max_seg = (g0.h + 127u) / 128u;
    if (nseg > max_seg)
      nseg = max_seg;
    if (nseg < 1)
      nseg = 1;
    uint32_t seg = ((g0.h + nseg - 1u) / nseg + 7u) & ~7u;
    nseg = (g0.h + seg - 1u) / seg;
    a.seg = (int)seg;
    dim3 grid(strips, nseg, batch);
    if (nr == 4)
      hipLaunchKernelGGL(k_octave_fused<4>, grid, dim3(384), 0, (hipStream_t)s, a);
    else
      hipLaunchKernelGGL(k_octave_fused<8>, grid, dim3(384), 0, (hipStream_t)s, a);
    return (int)hipGetLastError();
  }
}
