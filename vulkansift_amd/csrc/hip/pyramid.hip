// pyramid.hip — Gaussian scale-space construction kernels for gfx950 (wave64, LDS-staged, register V-window).
//
// Replaces, in the reference (paths relative to src/vulkansift/):
//   vkCmdCopyBufferToImage + vkCmdBlitImage(LINEAR)   sift_detector.c:881, 909-916   -> k_input_blit / fused into the seed blur
//   GaussianBlur*.comp H + V dispatches               sift_detector.c:927-1001       -> k_blur_lean (k_blur_tile: generic fallback)
//   vkCmdBlitImage(NEAREST) down-sample               sift_detector.c:1003-1034      -> fused into the scale-S blur / k_downsample
//   DifferenceOfGaussian.comp                         sift_detector.c:1039-1079      -> NOT a pass here: D[s] = G[s+1] - G[s] is one
//       fp32 subtraction of two stored planes, so the extrema scan and the refinement (extrema.hip) form it on the fly from the
//       Gaussian planes, bit-identically, and the DoG planes never exist in HBM (20 B per octave pixel less traffic at S = 3).
//       k_dog_plane materialises one layer on demand for vksift_downloadDoGImage.
//
// Arithmetic contract (must stay bit-identical to oracle/sift_oracle.c blur_plane/blit_*):
//   pass:  acc = centre*k0;  acc = fmaf(t(+i) + t(-i), k[i], acc)  for i = 1..n-1 ascending
//   compiled with -ffp-contract=off so nothing else is fused.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "multi.h"
#include "vksift_hip.h"

namespace
{

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct Taps
{
  float k[VKSIFT_HIP_MAX_TAPS];
};

// Scale-space texels are fp32, or (VKSIFT_PYRAMID_PRECISION_FLOAT16, F16 = true) IEEE binary16: stored with round-to-nearest-even
// from the fp32 result, widened exactly on every read; all arithmetic is fp32 either way (the oracle's pyramid_fp16 mode).
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
// fp32 -> binary16 of a value that was ROUNDED TO fp32 first: the compiler otherwise folds "fptrunc(fma(...))" into
// v_fma_mixlo_f16, which rounds the exact fma result to binary16 once — one ulp off the defined semantics (fp32 arithmetic,
// then round to nearest even on store) in rare cases. (With the SLP vectoriser on, the packed fma happened to keep the two
// roundings apart; the scalar form does not.) The empty asm pins the fp32 value in a register.
__device__ __forceinline__ _Float16 to_h(float v)
{
  asm volatile("" : "+v"(v));
  return (_Float16)v;
}
template <bool F16>
__device__ __forceinline__ float ld_px(const float *base, size_t idx)
{
  if (F16)
    return (float)((const _Float16 *)base)[idx];
  return base[idx];
}
template <bool F16>
__device__ __forceinline__ void st_px(float *base, size_t idx, float v)
{
  if (F16)
    ((_Float16 *)base)[idx] = to_h(v);
  else
    base[idx] = v;
}
// image b of a batch: strides are in texels
template <bool F16>
__device__ __forceinline__ float *img_ptr(float *base, size_t texels)
{
  return F16 ? (float *)((_Float16 *)base + texels) : base + texels;
}
template <bool F16>
__device__ __forceinline__ const float *img_ptr(const float *base, size_t texels)
{
  return F16 ? (const float *)((const _Float16 *)base + texels) : base + texels;
}
__device__ __forceinline__ unsigned pack_h2(float a, float b)
{
  const h2v h{to_h(a), to_h(b)};
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ void unpack_h2(unsigned u, float &a, float &b)
{
  const h2v h = __builtin_bit_cast(h2v, u);
  a = (float)h.x, b = (float)h.y;
}

// VK_SAMPLER_ADDRESS_MODE_MIRRORED_REPEAT (sift_detector.c:214-216)
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

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---------------------------------------------------------------------------------------------
// u8 -> fp32 with optional bilinear resize (Vulkan blit coordinate rules, clamp-to-edge).
// One thread per destination pixel; destination rows are written fully coalesced.
// ---------------------------------------------------------------------------------------------
template <bool F16>
__global__ void __launch_bounds__(256) k_input_blit(const uint8_t *__restrict__ src, int sw, int sh, uint64_t src_img_stride, float *__restrict__ dst,
                                                    int dw, int dh, int dpitch, uint64_t dst_img_stride)
{
  int x = blockIdx.x * 64 + (threadIdx.x & 63);
  int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh)
    return;
  const uint8_t *img = src + (size_t)blockIdx.z * src_img_stride;
  float *out = img_ptr<F16>(dst, (size_t)blockIdx.z * dst_img_stride);
  if (dw == sw && dh == sh)
  {
    st_px<F16>(out, (size_t)y * dpitch + x, (float)img[(size_t)y * sw + x] / 255.f);
    return;
  }
  float sx = (float)sw / (float)dw, sy = (float)sh / (float)dh;
  float v = ((float)y + 0.5f) * sy - 0.5f;
  float fy = floorf(v);
  float b = v - fy;
  int y0 = clampi((int)fy, 0, sh - 1), y1 = clampi((int)fy + 1, 0, sh - 1);
  float u = ((float)x + 0.5f) * sx - 0.5f;
  float fx = floorf(u);
  float a = u - fx;
  int x0 = clampi((int)fx, 0, sw - 1), x1 = clampi((int)fx + 1, 0, sw - 1);
  float t00 = (float)img[(size_t)y0 * sw + x0] / 255.f, t10 = (float)img[(size_t)y0 * sw + x1] / 255.f;
  float t01 = (float)img[(size_t)y1 * sw + x0] / 255.f, t11 = (float)img[(size_t)y1 * sw + x1] / 255.f;
  float r0 = fmaf(a, t10, (1.f - a) * t00);
  float r1 = fmaf(a, t11, (1.f - a) * t01);
  st_px<F16>(out, (size_t)y * dpitch + x, fmaf(b, r1, (1.f - b) * r0));
}

// ---------------------------------------------------------------------------------------------
// Exact 2x up-sampling fast path of the input blit: one thread produces 4 consecutive texels of one
// output row (one 16-byte store) from a 2-row x 4-column u8 neighbourhood. For a 2:1 blit the Vulkan
// coordinates u = (x + 0.5)/2 - 0.5 are exact in fp32 and the bilinear weights are exactly 0.25 / 0.75,
// so this evaluates the same fmaf expressions as k_input_blit (bit-identical) with 8 instead of 16
// u8 -> float divisions per 4 texels.
// ---------------------------------------------------------------------------------------------
template <bool F16>
__global__ void __launch_bounds__(256) k_input_blit_2x(const uint8_t *__restrict__ src, int sw, int sh, uint64_t src_img_stride,
                                                       float *__restrict__ dst, int dw, int dh, int dpitch, uint64_t dst_img_stride)
{
  const int xq = blockIdx.x * 64 + (threadIdx.x & 63); // group of 4 output columns
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int x = 4 * xq;
  if (x >= dw || y >= dh)
    return;
  const uint8_t *img = src + (size_t)blockIdx.z * src_img_stride;
  float *out = img_ptr<F16>(dst, (size_t)blockIdx.z * dst_img_stride + (size_t)y * dpitch + x);
  // rows: v = (y + 0.5)/2 - 0.5 -> fy = floor(v), b = v - fy
  const float v = ((float)y + 0.5f) * 0.5f - 0.5f;
  const float fy = floorf(v);
  const float b = v - fy;
  const int y0 = clampi((int)fy, 0, sh - 1), y1 = clampi((int)fy + 1, 0, sh - 1);
  // source columns 2xq-1 .. 2xq+2 cover output columns x .. x+3
  const int c0 = x / 2 - 1;
  float t0[4], t1[4];
#pragma unroll
  for (int k = 0; k < 4; k++)
  {
    const int c = clampi(c0 + k, 0, sw - 1);
    t0[k] = (float)img[(size_t)y0 * sw + c] / 255.f;
    t1[k] = (float)img[(size_t)y1 * sw + c] / 255.f;
  }
  float res[4];
#pragma unroll
  for (int k = 0; k < 4; k++)
  {
    // output column x+k: even -> (0.25, 0.75) over source (c0 + k/2, c0 + k/2 + 1); odd -> (0.75, 0.25) over (c0 + (k+1)/2, +1)
    const int i0 = (k + 1) / 2; // index into t[] of the left source texel
    const float a = (k & 1) ? 0.25f : 0.75f; // weight of the right texel: u - floor(u)
    const float r0 = fmaf(a, t0[i0 + 1], (1.f - a) * t0[i0]);
    const float r1 = fmaf(a, t1[i0 + 1], (1.f - a) * t1[i0]);
    res[k] = fmaf(b, r1, (1.f - b) * r0);
  }
  if (x + 3 < dw && !F16)
    *(float4 *)out = make_float4(res[0], res[1], res[2], res[3]);
  else if (x + 3 < dw)
    *(uint2 *)out = make_uint2(pack_h2(res[0], res[1]), pack_h2(res[2], res[3])); // x is a multiple of 4: 8-byte aligned
  else
    for (int k = 0; k < 4 && x + k < dw; k++)
      st_px<F16>(out, k, res[k]);
}

// ---------------------------------------------------------------------------------------------
// Generic fallback: fused separable blur over a 64x64 output tile staged through LDS (any width, any tap count).
//   LDS: s_src[(64+2R)][SS]  source tile with halo (mirrored at the image border)
//        s_mid[(64+2R)][64]  horizontally blurred rows
// 4 waves; lane = column so every LDS access is stride-1 (conflict-free) and every global row
// access is one 256-byte coalesced segment. Serves the shapes k_blur_lean does not (widths that are not a
// multiple of 4, images narrower than the halo, 1-tap kernels).
// ---------------------------------------------------------------------------------------------
constexpr int TILE = 64;

template <bool F16>
__global__ void __launch_bounds__(256) k_blur_tile(const float *__restrict__ src, uint64_t src_img_stride, int spitch, float *__restrict__ dst,
                                                   uint64_t dst_img_stride, int dpitch, int w, int h, Taps taps, int ntaps)
{
  extern __shared__ float lds[];
  const int R = ntaps - 1;
  const int SW = TILE + 2 * R;  // staged width
  const int SS = SW + 1;        // row stride (odd: keeps the column-strided halo loads off one bank)
  const int SH = TILE + 2 * R;  // staged height
  float *s_src = lds;
  float *s_mid = lds + SH * SS;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int x0 = blockIdx.x * TILE, y0 = blockIdx.y * TILE;
  const float *in = img_ptr<F16>(src, (size_t)blockIdx.z * src_img_stride);

  for (int r = wave; r < SH; r += 4)
  {
    int gy = mirror_idx(y0 - R + r, h);
    for (int c = lane; c < SW; c += 64)
      s_src[r * SS + c] = ld_px<F16>(in, (size_t)gy * spitch + mirror_idx(x0 - R + c, w));
  }
  __syncthreads();

  const float k0 = taps.k[0];
  for (int r = wave; r < SH; r += 4)
  {
    const float *p = s_src + r * SS + lane + R;
    float acc = p[0] * k0;
    for (int i = 1; i < ntaps; i++)
      acc = fmaf(p[i] + p[-i], taps.k[i], acc);
    s_mid[r * TILE + lane] = F16 ? (float)to_h(acc) : acc; // the blur temporary is an image of the pyramid format
  }
  __syncthreads();

  const int gx = x0 + lane;
  float *out = img_ptr<F16>(dst, (size_t)blockIdx.z * dst_img_stride);
  for (int rr = wave * 16; rr < wave * 16 + 16; rr++)
  {
    int gy = y0 + rr;
    if (gy >= h)
      break;
    const float *p = s_mid + (rr + R) * TILE + lane;
    float acc = p[0] * k0;
    for (int i = 1; i < ntaps; i++)
      acc = fmaf(p[i * TILE] + p[-i * TILE], taps.k[i], acc);
    if (gx < w)
      st_px<F16>(out, (size_t)gy * dpitch + gx, acc);
  }
}
// ---------------------------------------------------------------------------------------------
// k_blur_lean — streaming separable blur, the production kernel.
//
// One wave owns a 128-column strip of a row segment [y0, y1) and marches down the "virtual" row sequence
// r = y0-R .. y1-1+R in groups of 8 rows; virtual row r is fed by source row mirror(r), so the mirrored-repeat border
// needs no special case below the loads.
//   stage   : float4 loads of the next group's source rows (prefetched one group ahead, 16 B/lane, coalesced, halo
//             RA = R rounded up to 4, mirrored in x at the image edges) -> LDS
//   H pass  : every lane blurs its 2 pixels of each new row from LDS (ds_read_b64, stride-1 lanes: conflict free)
//             straight into the register window
//   V pass  : the register window holds the 2R+8 most recent H rows of the lane's own two columns; 8 output rows are
//             produced per group, then the window shifts by 8
// Everything is unrolled on the tap count (template): taps sit in SGPRs, inner loops are straight v_pk_add/v_pk_fma
// chains. Built around the instruction budget (a CDNA wave issues at most one instruction per 4 cycles whatever its type):
//   * global accesses are raw buffer instructions: per-lane byte offsets are loop constants, the row offset is one
//     SGPR; lanes that must not load/store (non-staging lanes, columns >= W) carry an out-of-range offset, so the
//     hardware drops them — no exec-mask branches, no 64-bit VALU address arithmetic
//   * mirrored columns are folded into the per-lane load offset (+ a lane-constant "reverse the float4" flag)
//   * the steady state (all 8 rows of a group inside the image and the segment) has no per-row conditions at all
//   * LDS per wave is one 8-row staging group (5 KiB at R = 12): occupancy is set by the registers alone
// HBM traffic per pixel: 4 B read (x (128+2RA)/128 horizontally, x (SEG+2R)/SEG vertically, mostly L2 hits thanks to the
// XCD-contiguous work order) + 4 B written.
// Requirements (checked by the launcher, k_blur_tile serves the rest): W % 4 == 0, a single mirror reflection
// covers every staged column (RA <= W and strips*128 + RA <= 2W).
// ---------------------------------------------------------------------------------------------
struct StreamArgs
{
  const float *src;
  float *dst;
  uint64_t src_img_stride, dst_img_stride;
  int spitch, dpitch;
  int w, h;
  int seg;   // output rows per workgroup (multiple of 8)
  float *ds; // also store the NEAREST 2:1 resample (odd rows, odd columns) of the result here (next octave's seed), or NULL
  uint64_t ds_img_stride;
  int ds_pitch;
  int rev; // walk the work space back to front (vksift_hip_Plane::reverse of the destination)
  Taps taps;
};

// Cache policy of the plane stores: non-temporal. A plane of a 512-frame call is 2.5 GB — nothing of it survives in the 4 MB L2 of an
// XCD until its reader (the next launch) comes by, and written the ordinary way it evicts the source rows the neighbouring strips
// and segments are about to re-read (halo columns, warm-up rows). Measured, same box: octave 0's launches 5.95 -> 5.59 ms per 512
// frames, +1.6 % frames/s; `sc0` (the other candidate) changes nothing.
#ifndef VKSIFT_ST_POLICY
#define VKSIFT_ST_POLICY 2
#endif
constexpr int ST_STREAM = VKSIFT_ST_POLICY; // 2 = the `nt` bit of a gfx950 buffer instruction
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned BUF_OOB = 0x80000000u; // byte offset beyond any plane: loads return 0, stores are dropped

// A 16-byte buffer store whose data registers the NEXT instruction may overwrite. gfx950 reads the store data of a wave over several
// cycles after issue; hipcc (ROCm 7.2) inserts the wait state this needs only when the store has NO scalar offset register
// (GCNHazardRecognizer: "this hazard only exists if the instruction is not using a register in the soffset field"), yet with
// `buffer_store_dwordx4 v[36:39], v88, s[12:15], s68 offen nt` directly followed by `v_pk_mul_f32 v[38:39], ...` the LAST dword of
// lanes 12-15 of every 16 arrived in memory with the next row's value (found by the first bit-comparison of k_blur_wide against
// k_blur_lean, whose stores are 8 bytes wide and have no such hazard). Two wait states behind the store, fenced against the scheduler.
__device__ __forceinline__ void store_b128_stream(u32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, ST_STREAM);
  asm volatile("s_nop 1");
  __builtin_amdgcn_sched_barrier(0);
}

template <bool F16>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const float *base, size_t texel_off, int pitch, int h)
{
  return __builtin_amdgcn_make_buffer_rsrc((void *)img_ptr<F16>(base, texel_off), 0, pitch * h * (F16 ? 2 : 4), 0x00020000);
}

// UPS: the source is the u8 input image at half the resolution; the staged rows are produced on the fly by the exact
// 2:1 LINEAR blit of k_input_blit_2x (same expressions, value/255 by unit_of_byte: the same correctly
// rounded quotients), i.e. vkCmdCopyBufferToImage + vkCmdBlitImage + the seed blur in one pass: the up-sampled plane
// never exists in HBM. a.src then points at the u8 images (a.src_img_stride in bytes), a.spitch is the source width.
// SRC: 0 = a plane of the pyramid; 1 = the u8 input at half the resolution (UPS above); 2 = the u8 input at the plane's own
// resolution (use_input_upsampling = false: vkCmdCopyBufferToImage + the 1:1 blit + the seed blur in one pass, value / 255
// through the same conversion). a.src then points at the u8 images (a.src_img_stride in bytes), a.spitch is the source width.
// byte k of d as float(byte) / 255.f, correctly rounded, without a division and without a table: q = x * fl(1/255) is off by an ulp for 126
// of the 256 values; one residual step — e = fma(-255, q, x), q + e * fl(1/255) — gives the IEEE quotient for all 256 (checked exhaustively:
// tests/test_oracle_host_math.py). v_cvt_f32_ubyteK + 3 VALU per texel instead of 3 VALU + a dependent LDS table read.
template <int K>
__device__ __forceinline__ float unit_of_byte(unsigned d)
{
  const float x = (float)((d >> (8 * K)) & 0xffu), r = 0x1.010102p-8f; // fl(1 / 255)
  const float q = x * r;
  return fmaf(fmaf(-255.f, q, x), r, q);
}
__device__ __forceinline__ void unit_of_bytes(unsigned d, float t[4])
{
  t[0] = unit_of_byte<0>(d), t[1] = unit_of_byte<1>(d), t[2] = unit_of_byte<2>(d), t[3] = unit_of_byte<3>(d);
}

// (the body is shared by k_blur_lean — one plane set per launch, the launch grid is the work grid — and k_blur_lean_multi — several
// octaves' planes in one flat launch, csrc/hip/multi.h: gx/gy/gz and bx/by/bz are then the octave's virtual grid)
template <int NT, int SRC, bool F16>
__device__ __forceinline__ void blur_lean_body(const StreamArgs &a, const uint32_t bx_, const uint32_t by_, const uint32_t bz_, const uint32_t gx_,
                                               const uint32_t gy_, const uint32_t gz_)
{
  constexpr bool UPS = SRC == 1, U8 = SRC == 2;
  constexpr int NR = 8;
  constexpr unsigned EB = F16 ? 2u : 4u; // bytes per texel
  constexpr int R = NT - 1;
  constexpr int RA = (R + 3) & ~3;
  constexpr int TW = 128;
  constexpr int SW = TW + 2 * RA;
  constexpr int NV4 = SW / 4;       // float4 per staged row (<= 42 lanes stage, one float4 per row each)
  constexpr int OFS = (RA - R) & 1; // parity fix so that the H-pass window starts on an even float
  constexpr int NP = R + 1 + OFS;   // float2 pairs read per row in the H pass
  constexpr int C0 = R + OFS;       // index of pixel 0's centre inside the window
  constexpr int NWIN = 2 * R + NR;
  __shared__ __attribute__((aligned(16))) float s_grp[NR * SW];

  const int lane = threadIdx.x;
  const int W = a.w, H = a.h;
  // XCD-aware work mapping: workgroup b is observed to run on XCD b % 8 (each XCD has its own L2). Give every XCD a
  // contiguous range of the (image, segment, strip) space, strips fastest, so that the workgroups that share halo
  // columns and warm-up rows run on the same XCD at about the same time and find them in its L2.
  uint32_t bs = bx_, bseg = by_, bimg = bz_;
  {
    const uint32_t total = gx_ * gy_ * gz_;
    const uint32_t b = bx_ + gx_ * (by_ + gy_ * bz_);
    uint32_t wi = b;
    if ((total & 7u) == 0)
    {
      const uint32_t per = total >> 3, k = b >> 3;
      wi = (b & 7u) * per + (a.rev ? per - 1u - k : k);
    }
    else if (a.rev)
      wi = total - 1u - b;
    if ((total & 7u) == 0 || a.rev)
    {
      bs = wi % gx_;
      const uint32_t r = wi / gx_;
      bseg = r % gy_;
      bimg = r / gy_;
    }
  }
  const int x0 = bs * TW;
  const int y0 = bseg * a.seg;
  const int y1 = min(y0 + a.seg, H);
  const __amdgpu_buffer_rsrc_t rs =
      (UPS || U8) ? __builtin_amdgcn_make_buffer_rsrc((void *)((const uint8_t *)a.src + (size_t)bimg * a.src_img_stride), 0, a.spitch * (UPS ? H / 2 : H), 0x00020000)
                  : plane_rsrc<F16>(a.src, (size_t)bimg * a.src_img_stride, a.spitch, H);
  const __amdgpu_buffer_rsrc_t rd = plane_rsrc<F16>(a.dst, (size_t)bimg * a.dst_img_stride, a.dpitch, H);
  const bool has_ds = SRC == 0 && a.ds != nullptr;
  const __amdgpu_buffer_rsrc_t rds =
      plane_rsrc<F16>(has_ds ? a.ds : a.dst, has_ds ? (size_t)bimg * a.ds_img_stride : 0, has_ds ? a.ds_pitch : a.dpitch, has_ds ? H / 2 : H);

  // ---- lane constants
  const int gx4 = x0 - RA + 4 * lane;
  unsigned ld_off = BUF_OOB;
  bool rev = false;
  if (lane < NV4)
  {
    if (gx4 >= 0 && gx4 + 3 < W)
      ld_off = (unsigned)gx4 * EB;
    else
    {
      ld_off = (unsigned)mirror_idx(gx4 + 3, W) * EB; // the four virtual columns map to m3+3, m3+2, m3+1, m3
      rev = true;
    }
  }
  // UPS: the 4 output columns X..X+3 (X = real column of the float4 after mirroring) come from the source bytes
  // X/2-1 .. X/2+2, clamped to the row: one (byte-aligned) dword load + a lane-constant byte permutation
  unsigned perm_sel = 0x03020100u;
  if (U8 && lane < NV4)
    ld_off /= EB; // one byte per source texel: the (mirrored) column index itself, a multiple of 4
  if (UPS || U8)
  {
    if (UPS && lane < NV4)
    {
      const int X = (int)(ld_off / EB);
      const int sw = a.spitch;
      int cb = X / 2 - 1;
      if (cb < 0)
        cb = 0, perm_sel = 0x02010000u; // (b0, b0, b1, b2)
      else if (cb + 3 > sw - 1)
        cb = sw - 4, perm_sel = 0x03030201u; // (b1, b2, b3, b3)
      ld_off = (unsigned)cb;
    }
  }
  const int px = x0 + 2 * lane;
  const unsigned st_off = px + 1 < W ? (unsigned)px * EB : BUF_OOB;
  const int spitch4 = a.spitch * (int)EB, dpitch4 = a.dpitch * (int)EB; // row pitches in bytes
  const unsigned st_off_ds = px + 1 < W ? (unsigned)(px >> 1) * EB : BUF_OOB; // column px+1 (odd) -> column px/2 of the half-size plane
  const int dspitch4 = a.ds_pitch * (int)EB;
  const float k0 = a.taps.k[0];

  u32x4 pf[NR];
  float vb[NR]; // UPS: vertical weight of the lower source row (wave-uniform)
  // UPS, group away from the top and bottom edges: output rows 2m+1 and 2m+2 interpolate between the SAME two source rows
  // (m, m+1) with weights 0.25 / 0.75, so the 8 rows of a group need 5 or 6 distinct source rows, not 16: each is loaded,
  // converted and interpolated horizontally once (the same expressions on the same values: bit-identical).
  // Groups start on rows of the parity of R (segments start on multiples of 8).
  constexpr int UPS_PAR = R & 1, UPS_NSRC = UPS_PAR ? 5 : 6;
  auto ups_shared = [&](int r0) { return r0 >= 2 && r0 + NR + 2 <= H; };
  // four texels of a source row: one 16-byte load (fp32) or one 8-byte load (fp16: .x, .y carry the four halves)
  auto load4 = [&](int so) -> u32x4 {
    if (F16)
    {
      const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, ld_off, so, 0);
      return u32x4{t.x, t.y, 0u, 0u};
    }
    return __builtin_amdgcn_raw_buffer_load_b128(rs, ld_off, so, 0);
  };
  auto prefetch = [&](int r0) {
    if (UPS && ups_shared(r0))
    {
      const int so0 = (UPS_PAR ? (r0 - 1) / 2 : r0 / 2 - 1) * a.spitch;
#pragma unroll
      for (int s = 0; s < UPS_NSRC; s++)
        pf[s].x = __builtin_amdgcn_raw_buffer_load_b32(rs, ld_off, so0 + s * a.spitch, 0);
    }
    else if (UPS)
    {
      const int sh = H / 2, sw = a.spitch;
#pragma unroll
      for (int j = 0; j < NR; j++)
      {
        const int ru = mirror_idx(r0 + j, H);
        const int m = (ru & 1) ? (ru - 1) / 2 : ru / 2 - 1;
        vb[j] = (ru & 1) ? 0.25f : 0.75f;
        const int ya = m < 0 ? 0 : m, yb_ = m + 1 > sh - 1 ? sh - 1 : m + 1;
        pf[j].x = __builtin_amdgcn_raw_buffer_load_b32(rs, ld_off, ya * sw, 0);
        pf[j].y = __builtin_amdgcn_raw_buffer_load_b32(rs, ld_off, yb_ * sw, 0);
      }
    }
    else if (U8)
    {
#pragma unroll
      for (int j = 0; j < NR; j++)
        pf[j].x = __builtin_amdgcn_raw_buffer_load_b32(rs, ld_off, mirror_idx(r0 + j, H) * a.spitch, 0);
    }
    else if (r0 >= 0 && r0 + NR <= H)
    {
      int so = r0 * spitch4;
#pragma unroll
      for (int j = 0; j < NR; j++, so += spitch4)
        pf[j] = load4(so);
    }
    else
    {
#pragma unroll
      for (int j = 0; j < NR; j++)
        pf[j] = load4(mirror_idx(r0 + j, H) * spitch4);
    }
  };

  int rg = y0 - R; // first virtual row of the current group
  prefetch(rg);
  float hrc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}; // UPS: see the staging of a shared group
  int hrc_from = -(1 << 30);                                      // ... the group they were left by

  float2 wv[NWIN];
#pragma unroll
  for (int k = 0; k < NWIN; k++)
    wv[k] = make_float2(0.f, 0.f);

  for (; rg - R < y1; rg += NR)
  {
    // ---- stage the prefetched group, then prefetch the next one
    __syncthreads();
    if (UPS && ups_shared(rg))
    {
      // the first UPS_NSRC - 4 source rows of this group are the last ones of the previous group (8 output rows = 4 source rows further
      // down): their horizontally interpolated texels are carried over in registers instead of being converted and interpolated
      // again — a third of this launch's conversion work, the same values
      const bool carried = hrc_from == rg - NR;
      hrc_from = rg;
      if (lane < NV4)
      {
        float hr[UPS_NSRC][4];
        auto hrow = [&](int s) {
          const unsigned d = __builtin_amdgcn_perm(pf[s].x, pf[s].x, perm_sel);
          float t[4];
          unit_of_bytes(d, t);
#pragma unroll
          for (int k = 0; k < 4; k++)
          {
            const int i0 = (k + 1) / 2;
            const float aw = (k & 1) ? 0.25f : 0.75f;
            hr[s][k] = fmaf(aw, t[i0 + 1], (1.f - aw) * t[i0]);
          }
        };
        if (carried)
        {
#pragma unroll
          for (int s = 0; s < UPS_NSRC - 4; s++)
#pragma unroll
            for (int k = 0; k < 4; k++)
              hr[s][k] = hrc[s][k];
        }
        else
        {
#pragma unroll
          for (int s = 0; s < UPS_NSRC - 4; s++)
            hrow(s);
        }
#pragma unroll
        for (int s = UPS_NSRC - 4; s < UPS_NSRC; s++)
          hrow(s);
#pragma unroll
        for (int s = 0; s < UPS_NSRC - 4; s++)
#pragma unroll
          for (int k = 0; k < 4; k++)
            hrc[s][k] = hr[4 + s][k];
#pragma unroll
        for (int j = 0; j < NR; j++)
        {
          const int p = UPS_PAR ? j / 2 : (j + 1) / 2;           // rows (p, p+1) of hr
          const float b = ((j + UPS_PAR) & 1) ? 0.25f : 0.75f;   // odd output rows sit nearer the upper source row
          float res[4];
#pragma unroll
          for (int k = 0; k < 4; k++)
          {
            res[k] = fmaf(b, hr[p + 1][k], (1.f - b) * hr[p][k]);
            if (F16)
              res[k] = (float)to_h(res[k]);
          }
          u32x4 v = u32x4{__float_as_uint(res[0]), __float_as_uint(res[1]), __float_as_uint(res[2]), __float_as_uint(res[3])};
          if (rev)
            v = u32x4{v.w, v.z, v.y, v.x};
          *(u32x4 *)(s_grp + j * SW + 4 * lane) = v;
        }
      }
    }
    else if (lane < NV4)
    {
#pragma unroll
      for (int j = 0; j < NR; j++)
      {
        u32x4 v = pf[j];
        if (UPS)
        {
          const unsigned d0 = __builtin_amdgcn_perm(v.x, v.x, perm_sel), d1 = __builtin_amdgcn_perm(v.y, v.y, perm_sel);
          float t0[4], t1[4], res[4];
          unit_of_bytes(d0, t0);
          unit_of_bytes(d1, t1);
          const float b = vb[j];
#pragma unroll
          for (int k = 0; k < 4; k++)
          {
            const int i0 = (k + 1) / 2;              // left source texel of output column X+k
            const float aw = (k & 1) ? 0.25f : 0.75f; // weight of the right one
            const float r0 = fmaf(aw, t0[i0 + 1], (1.f - aw) * t0[i0]);
            const float r1 = fmaf(aw, t1[i0 + 1], (1.f - aw) * t1[i0]);
            res[k] = fmaf(b, r1, (1.f - b) * r0);
            if (F16)
              res[k] = (float)to_h(res[k]); // the blit target is an image of the pyramid format
          }
          v = u32x4{__float_as_uint(res[0]), __float_as_uint(res[1]), __float_as_uint(res[2]), __float_as_uint(res[3])};
        }
        else if (U8)
        {
          float res[4];
          unit_of_bytes(v.x, res);
#pragma unroll
          for (int k = 0; k < 4; k++)
          {
            if (F16)
              res[k] = (float)to_h(res[k]); // the blit target is an image of the pyramid format
          }
          v = u32x4{__float_as_uint(res[0]), __float_as_uint(res[1]), __float_as_uint(res[2]), __float_as_uint(res[3])};
        }
        else if (F16)
        {
          float f0, f1, f2, f3;
          unpack_h2(v.x, f0, f1);
          unpack_h2(v.y, f2, f3);
          v = u32x4{__float_as_uint(f0), __float_as_uint(f1), __float_as_uint(f2), __float_as_uint(f3)};
        }
        if (rev)
          v = u32x4{v.w, v.z, v.y, v.x};
        *(u32x4 *)(s_grp + j * SW + 4 * lane) = v;
      }
    }
    prefetch(rg + NR);
    __syncthreads();

    // ---- horizontal pass of the new rows, into the top of the register window. Two rows at a time: their accumulator
    // chains are independent, so the scheduler can alternate them and the dependent v_pk_add -> v_pk_fma pairs need no
    // wait states (one row alone leaves an s_nop after almost every packed instruction).
    {
      const float *hb = s_grp + (RA - R - OFS) + 2 * lane;
#pragma unroll
      for (int j = 0; j < NR; j += 2)
      {
        const v2f *pa = (const v2f *)(hb + j * SW);
        const v2f *pb = (const v2f *)(hb + (j + 1) * SW);
        float va[2 * NP], vb2[2 * NP];
#pragma unroll
        for (int q = 0; q < NP; q++)
        {
          v2f ta = pa[q], tb = pb[q];
          va[2 * q] = ta.x, va[2 * q + 1] = ta.y;
          vb2[2 * q] = tb.x, vb2[2 * q + 1] = tb.y;
        }
        float a0 = va[C0] * k0, a1 = va[C0 + 1] * k0;
        float b0 = vb2[C0] * k0, b1 = vb2[C0 + 1] * k0;
#pragma unroll
        for (int i = 1; i < NT; i++)
        {
          a0 = fmaf(va[C0 + i] + va[C0 - i], a.taps.k[i], a0);
          a1 = fmaf(va[C0 + 1 + i] + va[C0 + 1 - i], a.taps.k[i], a1);
          b0 = fmaf(vb2[C0 + i] + vb2[C0 - i], a.taps.k[i], b0);
          b1 = fmaf(vb2[C0 + 1 + i] + vb2[C0 + 1 - i], a.taps.k[i], b1);
        }
        if (F16) // the horizontal pass's output is an image of the pyramid format too (the reference's blur temporary)
          a0 = (float)to_h(a0), a1 = (float)to_h(a1), b0 = (float)to_h(b0), b1 = (float)to_h(b1);
        wv[2 * R + j] = make_float2(a0, a1);
        wv[2 * R + j + 1] = make_float2(b0, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- vertical pass: output rows yb .. yb+NR-1
    const int yb = rg - R;
    if (yb + NR > y0)
    {
      auto emit = [&](int j, int so_d, float acc0, float acc1) {
        if (F16)
          __builtin_amdgcn_raw_buffer_store_b32(pack_h2(acc0, acc1), rd, st_off, so_d, ST_STREAM);
        else
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(acc0), __float_as_uint(acc1)}, rd, st_off, so_d, ST_STREAM);
        // vkCmdBlitImage(NEAREST) into the next octave, exact 2:1: destination (x, y) takes source (2x+1, 2y+1). yb is even
        // (segments start on multiples of 8), so the odd rows are the odd j: a compile-time choice in the unrolled loops
        if (has_ds && (j & 1))
        {
          if (F16)
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pack_h2(acc1, 0.f) & 0xffffu), rds, st_off_ds, ((yb + j) >> 1) * dspitch4, ST_STREAM);
          else
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc1), rds, st_off_ds, ((yb + j) >> 1) * dspitch4, ST_STREAM);
        }
      };
      if (yb >= y0 && yb + NR <= y1)
      {
        int so_d = yb * dpitch4;
#pragma unroll
        for (int j = 0; j < NR; j += 2, so_d += 2 * dpitch4)
        {
          float a0 = wv[R + j].x * k0, a1 = wv[R + j].y * k0;
          float b0 = wv[R + j + 1].x * k0, b1 = wv[R + j + 1].y * k0;
#pragma unroll
          for (int i = 1; i < NT; i++)
          {
            a0 = fmaf(wv[R + j + i].x + wv[R + j - i].x, a.taps.k[i], a0);
            a1 = fmaf(wv[R + j + i].y + wv[R + j - i].y, a.taps.k[i], a1);
            b0 = fmaf(wv[R + j + 1 + i].x + wv[R + j + 1 - i].x, a.taps.k[i], b0);
            b1 = fmaf(wv[R + j + 1 + i].y + wv[R + j + 1 - i].y, a.taps.k[i], b1);
          }
          emit(j, so_d, a0, a1);
          emit(j + 1, so_d + dpitch4, b0, b1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      else
      {
        int so_d = yb * dpitch4;
#pragma unroll
        for (int j = 0; j < NR; j++, so_d += dpitch4)
        {
          if (yb + j < y0 || yb + j >= y1)
            continue;
          float acc0 = wv[R + j].x * k0, acc1 = wv[R + j].y * k0;
#pragma unroll
          for (int i = 1; i < NT; i++)
          {
            acc0 = fmaf(wv[R + j + i].x + wv[R + j - i].x, a.taps.k[i], acc0);
            acc1 = fmaf(wv[R + j + i].y + wv[R + j - i].y, a.taps.k[i], acc1);
          }
          emit(j, so_d, acc0, acc1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // ---- slide the window
#pragma unroll
    for (int k = 0; k < 2 * R; k++)
      wv[k] = wv[k + NR];
  }
}

template <int NT, int SRC, bool F16>
__global__ void __launch_bounds__(64) k_blur_lean(StreamArgs a)
{
  blur_lean_body<NT, SRC, F16>(a, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y, gridDim.z);
}

// The same scale of SEVERAL octaves in one launch (round 6): scales S+1 and S+2 of an octave feed nothing but the extrema scan, so they can
// wait until the chain seed -> ... -> scale S has run through every octave and then go as ONE launch per scale over all octaves instead of
// one per octave and scale — for one image 8 dependent launches of 11 us become 4, for a batch the coarse octaves' launches (too small to fill the
// chip) ride in the tail of the larger ones'. Flat grid, every workgroup looks up its octave and its place in that octave's own grid.
template <int NT, bool F16>
__global__ void __launch_bounds__(64) k_blur_lean_multi(Multi<StreamArgs> m)
{
  const VBlock vb = vblock(m);
  blur_lean_body<NT, 0, F16>(m.oct[vb.o], vb.x, vb.y, vb.z, vb.gx, vb.gy, vb.gz);
}

// ---------------------------------------------------------------------------------------------
// k_blur_wide — the strip march of k_blur_lean with FOUR output texels per lane (256-column strips), fp32 plane -> fp32 plane.
//
// Why: the horizontal pass of k_blur_lean reads a (2R + 2)-float window per lane for 2 texels through ds_read2_b64 — the slow LDS
// read form (128 B/clk/CU) and 5-14x read amplification; at 11 and 13 taps the LDS array is busy as long as the VALUs are
// (8 waves x 7 ds_read2_b64 x 8 rows against 2 waves x ~900 VALU per SIMD and group), and the two do not hide behind each other
// with two waves per SIMD. Here a lane owns texels 4L .. 4L+3 of the strip and reads the 16-byte-aligned window
// [4L, 4L + 2 RA + 4) of the staged row with RA/2 + 1 ds_read_b128 (256 B/clk/CU): a quarter of the LDS-array cycles per texel.
// The strip's horizontal halo (2 RA columns) is shared by 256 columns instead of 128, rows are stored 16 bytes per lane
// (half the store instructions), every lane stages (64 + RA/2 float4 per row: the RA/2 beyond the 64 lanes of all 8 rows of a
// group are ONE load + ONE LDS write of up to 48 lanes). The price is the register window: 4 floats x (2R + 8) rows.
// Same operations per texel in the same order as k_blur_lean / blur_plane of the oracle: bit-identical. (The fused up-sampling + seed
// launch was built in this form too, bit-identical at the first run and no faster — 542 vs 530 us per 512 frames: that launch issues
// ~930 VALU instructions per 2048 texels in either form, conversion and interpolation of the u8 source among them — and is not kept.)
// ---------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(64) k_blur_wide(StreamArgs a)
{
  constexpr int NR = 8;
  constexpr int R = NT - 1;
  constexpr int RA = (R + 3) & ~3;
  constexpr int TW = 256;
  constexpr int SW = TW + 2 * RA;
  constexpr int NX = RA / 2;   // float4 columns of a staged row beyond the 64 that the lanes stage themselves
  constexpr int NXT = NX * NR; // ... of a whole group: lane e < NXT stages float4 column 64 + e % NX of row e / NX
  constexpr int NQ = RA / 2 + 1; // ds_read_b128 per row in the horizontal pass
  constexpr int NWIN = 2 * R + NR;
  static_assert(NXT <= 64, "the extra float4 columns of a group are staged by one instruction");
  __shared__ __attribute__((aligned(16))) float s_grp[NR * SW];

  const int lane = threadIdx.x;
  const int W = a.w, H = a.h;
  uint32_t bs = blockIdx.x, bseg = blockIdx.y, bimg = blockIdx.z;
  {
    // XCD-aware work mapping, as k_blur_lean
    const uint32_t total = gridDim.x * gridDim.y * gridDim.z;
    const uint32_t b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    uint32_t wi = b;
    if ((total & 7u) == 0)
    {
      const uint32_t per = total >> 3, k = b >> 3;
      wi = (b & 7u) * per + (a.rev ? per - 1u - k : k);
    }
    else if (a.rev)
      wi = total - 1u - b;
    if ((total & 7u) == 0 || a.rev)
    {
      bs = wi % gridDim.x;
      const uint32_t r = wi / gridDim.x;
      bseg = r % gridDim.y;
      bimg = r / gridDim.y;
    }
  }
  const int x0 = bs * TW;
  const int y0 = bseg * a.seg;
  const int y1 = min(y0 + a.seg, H);
  const __amdgpu_buffer_rsrc_t rs = plane_rsrc<false>(a.src, (size_t)bimg * a.src_img_stride, a.spitch, H);
  const __amdgpu_buffer_rsrc_t rd = plane_rsrc<false>(a.dst, (size_t)bimg * a.dst_img_stride, a.dpitch, H);
  const bool has_ds = a.ds != nullptr;
  const __amdgpu_buffer_rsrc_t rds =
      plane_rsrc<false>(has_ds ? a.ds : a.dst, has_ds ? (size_t)bimg * a.ds_img_stride : 0, has_ds ? a.ds_pitch : a.dpitch, has_ds ? H / 2 : H);
  const int spitch4 = a.spitch * 4, dpitch4 = a.dpitch * 4, dspitch4 = a.ds_pitch * 4; // row pitches in bytes

  // ---- lane constants: the float4 column this lane stages in every row ...
  auto col_off = [&](int gx4, bool &rv) -> unsigned {
    if (gx4 >= 0 && gx4 + 3 < W)
      return (unsigned)gx4 * 4u;
    rv = true;
    return (unsigned)mirror_idx(gx4 + 3, W) * 4u; // the four virtual columns map to m3+3, m3+2, m3+1, m3
  };
  bool rev = false, rev_x = false;
  const unsigned ld_off = col_off(x0 - RA + 4 * lane, rev);
  // ... and the (row, column) beyond the 64th float4 it stages once per group
  const int xr = lane / NX, xq = 64 + lane % NX;
  unsigned ldx_col = BUF_OOB;
  if (lane < NXT)
    ldx_col = col_off(x0 - RA + 4 * xq, rev_x);
  const unsigned ldx_off = lane < NXT ? ldx_col + (unsigned)(xr * spitch4) : BUF_OOB;
  float *const sx = s_grp + xr * SW + 4 * xq;
  const int px = x0 + 4 * lane;
  const unsigned st_off = px + 3 < W ? (unsigned)px * 4u : BUF_OOB;
  const unsigned st_off_ds = px + 3 < W ? (unsigned)(px >> 1) * 4u : BUF_OOB; // columns px+1, px+3 -> px/2, px/2+1 of the half-size plane
  const float k0 = a.taps.k[0];

  u32x4 pf[NR], pfx;
  auto prefetch = [&](int r0) {
    if (r0 >= 0 && r0 + NR <= H)
    {
      int so = r0 * spitch4;
      pfx = __builtin_amdgcn_raw_buffer_load_b128(rs, ldx_off, so, 0);
#pragma unroll
      for (int j = 0; j < NR; j++, so += spitch4)
        pf[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, ld_off, so, 0);
    }
    else
    {
      // a group that crosses the top or bottom edge: every row mirrored on its own (the extra column's row is a lane value)
      const unsigned ox = lane < NXT ? ldx_col + (unsigned)(mirror_idx(r0 + xr, H) * spitch4) : BUF_OOB;
      pfx = __builtin_amdgcn_raw_buffer_load_b128(rs, ox, 0, 0);
#pragma unroll
      for (int j = 0; j < NR; j++)
        pf[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, ld_off, mirror_idx(r0 + j, H) * spitch4, 0);
    }
  };

  int rg = y0 - R; // first virtual row of the current group
  prefetch(rg);

  // The register window is a RING of NWP = NPH * NR rows: in phase P (the P-th group of a round) window row k lives in
  // wv[(k + P * NR) % NWP], so the window slides by renaming — the march loop is unrolled over the NPH phases and every index is a
  // compile-time constant. (The copy form, wv[k] = wv[k + NR] for the 2R rows that stay, was 96 of ~1550 VALU instructions per group at 13
  // taps.) NWIN rounded up to whole groups: 4 spare rows at 11 taps.
  constexpr int NPH = (NWIN + NR - 1) / NR, NWP = NPH * NR;
  static_assert(NPH <= 4, "phases spelled out in the march loop");
  v4f wv[NWP];
#pragma unroll
  for (int k = 0; k < NWP; k++)
    wv[k] = v4f{0.f, 0.f, 0.f, 0.f};

  auto group = [&](auto PH) {
    constexpr int P = decltype(PH)::value;
#define WV(k) wv[((k) + P * NR) % NWP]
    // ---- stage the prefetched group, then prefetch the next one
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NR; j++)
    {
      u32x4 v = pf[j];
      if (rev)
        v = u32x4{v.w, v.z, v.y, v.x};
      *(u32x4 *)(s_grp + j * SW + 4 * lane) = v;
    }
    if (lane < NXT)
    {
      u32x4 v = pfx;
      if (rev_x)
        v = u32x4{v.w, v.z, v.y, v.x};
      *(u32x4 *)sx = v;
    }
    prefetch(rg + NR);
    __syncthreads();

    // ---- horizontal pass of the new rows, into the top of the register window: four independent accumulator chains per row
    {
      const v4f *hb = (const v4f *)(s_grp + 4 * lane);
#pragma unroll
      for (int j = 0; j < NR; j++)
      {
        float va[4 * NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++)
        {
          const v4f t = hb[j * (SW / 4) + q];
          va[4 * q] = t.x, va[4 * q + 1] = t.y, va[4 * q + 2] = t.z, va[4 * q + 3] = t.w;
        }
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
          o[k] = va[RA + k] * k0;
#pragma unroll
        for (int i = 1; i < NT; i++)
#pragma unroll
          for (int k = 0; k < 4; k++)
            o[k] = fmaf(va[RA + k + i] + va[RA + k - i], a.taps.k[i], o[k]);
        WV(2 * R + j) = v4f{o[0], o[1], o[2], o[3]};
      }
    }

    // ---- vertical pass: output rows yb .. yb+NR-1
    const int yb = rg - R;
    if (yb + NR > y0)
    {
      auto vrow = [&](auto J, int so_d) {
        constexpr int j = decltype(J)::value;
        v4f acc = WV(R + j) * k0;
#pragma unroll
        for (int i = 1; i < NT; i++)
        {
          const v4f sm = WV(R + j + i) + WV(R + j - i);
          acc.x = fmaf(sm.x, a.taps.k[i], acc.x);
          acc.y = fmaf(sm.y, a.taps.k[i], acc.y);
          acc.z = fmaf(sm.z, a.taps.k[i], acc.z);
          acc.w = fmaf(sm.w, a.taps.k[i], acc.w);
        }
        store_b128_stream(u32x4{__float_as_uint(acc.x), __float_as_uint(acc.y), __float_as_uint(acc.z), __float_as_uint(acc.w)}, rd, st_off, so_d);
        // vkCmdBlitImage(NEAREST) into the next octave, exact 2:1: destination (x, y) takes source (2x+1, 2y+1); yb is even
        if (has_ds && (j & 1))
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(acc.y), __float_as_uint(acc.w)}, rds, st_off_ds, ((yb + j) >> 1) * dspitch4, ST_STREAM);
      };
      const int so_0 = yb * dpitch4;
#define VROW(J) vrow(std::integral_constant<int, J>{}, so_0 + J * dpitch4);
#define VROWC(J)                            \
  if (yb + J >= y0 && yb + J < y1)          \
    VROW(J)
      if (yb >= y0 && yb + NR <= y1)
      {
        VROW(0) VROW(1) VROW(2) VROW(3) VROW(4) VROW(5) VROW(6) VROW(7)
      }
      else
      {
        VROWC(0) VROWC(1) VROWC(2) VROWC(3) VROWC(4) VROWC(5) VROWC(6) VROWC(7)
      }
#undef VROWC
#undef VROW
    }
#undef WV
  };

  for (;;)
  {
#define PHASE(P)                                      \
  if (P < NPH)                                        \
  {                                                   \
    if (!(rg - R < y1))                               \
      break;                                          \
    group(std::integral_constant<int, (P < NPH ? P : 0)>{}); \
    rg += NR;                                         \
  }
    PHASE(0) PHASE(1) PHASE(2) PHASE(3)
#undef PHASE
  }
}

// ---------------------------------------------------------------------------------------------
// k_blur_pair — TWO consecutive scales in one launch: reads plane s-1 once, writes planes s and s+1 (12 B per texel instead
// of 16 for the two launches of k_blur_lean; the chain of an octave is memory bound in its narrow-tap launches).
//
// Same strip march as k_blur_lean (fp32 planes, plane source). A wave COMPUTES scale s on 128 columns and OWNS the inner
// 128 - 2 HC of them (HC = the second filter's radius rounded up to 4): the outer columns are the horizontal halo the second
// filter needs, recomputed by the neighbouring strips (blurring a mirrored row gives the mirrored blurred row bit for bit —
// symmetric taps, commutative adds — so halo columns beyond the image edge need no special case either). Per group of 8 source
// rows: stage -> H pass 1 -> register window 1 -> V pass 1 = 8 rows of scale s -> (own columns / own rows) to HBM and (all 128
// columns) to a second LDS buffer -> H pass 2 from it -> register window 2 -> V pass 2 = 8 rows of scale s+1, R2 rows behind.
// The segment's march starts R1 + R2 rows early; rows of scale s above / below the segment are the vertical halo of the second
// filter (computed, not stored: the neighbouring segment stores them). Every value that is stored went through exactly the
// operations of the two separate launches, in the same order.
// ---------------------------------------------------------------------------------------------
struct PairArgs
{
  const float *src;
  float *dst1, *dst2;
  uint64_t src_img_stride, dst1_img_stride, dst2_img_stride;
  int spitch, d1pitch, d2pitch;
  int w, h;
  int seg; // output rows per workgroup (multiple of 8)
  int rev; // walk the work space back to front
  Taps t1, t2;
};

template <int NT1, int NT2>
__global__ void __launch_bounds__(64) k_blur_pair(PairArgs a)
{
  constexpr int NR = 8;
  constexpr int R1 = NT1 - 1, R2 = NT2 - 1;
  constexpr int RA1 = (R1 + 3) & ~3;
  constexpr int HC = (R2 + 3) & ~3;
  constexpr int TW = 128, OW = TW - 2 * HC;
  constexpr int SW = TW + 2 * RA1;
  constexpr int NV4 = SW / 4;
  constexpr int OFS1 = (RA1 - R1) & 1, NP1 = R1 + 1 + OFS1, C01 = R1 + OFS1;
  constexpr int OFS2 = R2 & 1, NP2 = R2 + 1 + OFS2, C02 = R2 + OFS2;
  constexpr int PAD = (R2 + OFS2 + 3) & ~3; // floats of padding on both sides of a scale-s row in LDS
  constexpr int G1S = TW + 2 * PAD;         // its row stride
  static_assert(R2 + OFS2 <= PAD, "padding of the second stage's LDS rows");
  constexpr int NWIN1 = 2 * R1 + NR, NWIN2 = 2 * R2 + NR;
  __shared__ __attribute__((aligned(16))) float s_grp[NR * SW];
  __shared__ __attribute__((aligned(16))) float s_g1[NR * G1S];

  const int lane = threadIdx.x;
  const int W = a.w, H = a.h;
  uint32_t bs = blockIdx.x, bseg = blockIdx.y, bimg = blockIdx.z;
  {
    const uint32_t total = gridDim.x * gridDim.y * gridDim.z;
    const uint32_t b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    uint32_t wi = b;
    if ((total & 7u) == 0)
    {
      const uint32_t per = total >> 3, k = b >> 3;
      wi = (b & 7u) * per + (a.rev ? per - 1u - k : k);
    }
    else if (a.rev)
      wi = total - 1u - b;
    bs = wi % gridDim.x;
    const uint32_t r = wi / gridDim.x;
    bseg = r % gridDim.y;
    bimg = r / gridDim.y;
  }
  const int x0 = (int)bs * OW - HC; // first computed column (virtual: negative on the first strip)
  const int y0 = bseg * a.seg;
  const int y1 = min(y0 + a.seg, H);
  const __amdgpu_buffer_rsrc_t rs = plane_rsrc<false>(a.src, (size_t)bimg * a.src_img_stride, a.spitch, H);
  const __amdgpu_buffer_rsrc_t rd1 = plane_rsrc<false>(a.dst1, (size_t)bimg * a.dst1_img_stride, a.d1pitch, H);
  const __amdgpu_buffer_rsrc_t rd2 = plane_rsrc<false>(a.dst2, (size_t)bimg * a.dst2_img_stride, a.d2pitch, H);

  // ---- lane constants
  const int gx4 = x0 - RA1 + 4 * lane;
  unsigned ld_off = BUF_OOB;
  bool rev = false;
  if (lane < NV4)
  {
    if (gx4 >= 0 && gx4 + 3 < W)
      ld_off = (unsigned)gx4 * 4u;
    else
    {
      ld_off = (unsigned)mirror_idx(gx4 + 3, W) * 4u; // the four virtual columns map to m3+3, m3+2, m3+1, m3
      rev = true;
    }
  }
  const int px = x0 + 2 * lane;
  const bool own = 2 * lane >= HC && 2 * lane < TW - HC;
  const unsigned st_off = (own && px >= 0 && px + 1 < W) ? (unsigned)px * 4u : BUF_OOB;
  const int spitch4 = a.spitch * 4, d1pitch4 = a.d1pitch * 4, d2pitch4 = a.d2pitch * 4;
  const float k10 = a.t1.k[0], k20 = a.t2.k[0];

  u32x4 pf[NR];
  auto prefetch = [&](int r0) {
    if (r0 >= 0 && r0 + NR <= H)
    {
      int so = r0 * spitch4;
#pragma unroll
      for (int j = 0; j < NR; j++, so += spitch4)
        pf[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, ld_off, so, 0);
    }
    else
    {
#pragma unroll
      for (int j = 0; j < NR; j++)
        pf[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, ld_off, mirror_idx(r0 + j, H) * spitch4, 0);
    }
  };

  int rg = y0 - R1 - R2; // first virtual source row of the current group
  prefetch(rg);

  float2 wv1[NWIN1], wv2[NWIN2];
#pragma unroll
  for (int k = 0; k < NWIN1; k++)
    wv1[k] = make_float2(0.f, 0.f);
#pragma unroll
  for (int k = 0; k < NWIN2; k++)
    wv2[k] = make_float2(0.f, 0.f);

  for (; rg - R1 - R2 < y1; rg += NR)
  {
    // ---- stage the prefetched group, then prefetch the next one
    __syncthreads();
    if (lane < NV4)
    {
#pragma unroll
      for (int j = 0; j < NR; j++)
      {
        u32x4 v = pf[j];
        if (rev)
          v = u32x4{v.w, v.z, v.y, v.x};
        *(u32x4 *)(s_grp + j * SW + 4 * lane) = v;
      }
    }
    prefetch(rg + NR);
    __syncthreads();

    // ---- first filter, horizontal: the new rows into the top of window 1 (two rows at a time, see k_blur_lean)
    {
      const float *hb = s_grp + (RA1 - R1 - OFS1) + 2 * lane;
#pragma unroll
      for (int j = 0; j < NR; j += 2)
      {
        const v2f *pa = (const v2f *)(hb + j * SW);
        const v2f *pb = (const v2f *)(hb + (j + 1) * SW);
        float va[2 * NP1], vb2[2 * NP1];
#pragma unroll
        for (int q = 0; q < NP1; q++)
        {
          v2f ta = pa[q], tb = pb[q];
          va[2 * q] = ta.x, va[2 * q + 1] = ta.y;
          vb2[2 * q] = tb.x, vb2[2 * q + 1] = tb.y;
        }
        float a0 = va[C01] * k10, a1 = va[C01 + 1] * k10;
        float b0 = vb2[C01] * k10, b1 = vb2[C01 + 1] * k10;
#pragma unroll
        for (int i = 1; i < NT1; i++)
        {
          a0 = fmaf(va[C01 + i] + va[C01 - i], a.t1.k[i], a0);
          a1 = fmaf(va[C01 + 1 + i] + va[C01 + 1 - i], a.t1.k[i], a1);
          b0 = fmaf(vb2[C01 + i] + vb2[C01 - i], a.t1.k[i], b0);
          b1 = fmaf(vb2[C01 + 1 + i] + vb2[C01 + 1 - i], a.t1.k[i], b1);
        }
        wv1[2 * R1 + j] = make_float2(a0, a1);
        wv1[2 * R1 + j + 1] = make_float2(b0, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- first filter, vertical: rows yb1 .. yb1+7 of scale s (all 128 columns: the second filter's horizontal halo included)
    const int yb1 = rg - R1;
    {
      int so_d = yb1 * d1pitch4;
#pragma unroll
      for (int j = 0; j < NR; j += 2, so_d += 2 * d1pitch4)
      {
        float a0 = wv1[R1 + j].x * k10, a1 = wv1[R1 + j].y * k10;
        float b0 = wv1[R1 + j + 1].x * k10, b1 = wv1[R1 + j + 1].y * k10;
#pragma unroll
        for (int i = 1; i < NT1; i++)
        {
          a0 = fmaf(wv1[R1 + j + i].x + wv1[R1 + j - i].x, a.t1.k[i], a0);
          a1 = fmaf(wv1[R1 + j + i].y + wv1[R1 + j - i].y, a.t1.k[i], a1);
          b0 = fmaf(wv1[R1 + j + 1 + i].x + wv1[R1 + j + 1 - i].x, a.t1.k[i], b0);
          b1 = fmaf(wv1[R1 + j + 1 + i].y + wv1[R1 + j + 1 - i].y, a.t1.k[i], b1);
        }
        *(v2f *)(s_g1 + j * G1S + PAD + 2 * lane) = v2f{a0, a1};
        *(v2f *)(s_g1 + (j + 1) * G1S + PAD + 2 * lane) = v2f{b0, b1};
        // the segment's own rows of scale s (wave-uniform tests)
        if (yb1 + j >= y0 && yb1 + j < y1)
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(a0), __float_as_uint(a1)}, rd1, st_off, so_d, ST_STREAM);
        if (yb1 + j + 1 >= y0 && yb1 + j + 1 < y1)
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(b0), __float_as_uint(b1)}, rd1, st_off, so_d + d1pitch4, ST_STREAM);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int k = 0; k < 2 * R1; k++)
      wv1[k] = wv1[k + NR];
    __syncthreads();

    // ---- second filter, horizontal, from the rows of scale s just written to LDS (lanes outside the owned columns read into the
    // padding: their results are never stored)
    {
      const float *hb = s_g1 + (PAD - R2 - OFS2) + 2 * lane;
#pragma unroll
      for (int j = 0; j < NR; j += 2)
      {
        const v2f *pa = (const v2f *)(hb + j * G1S);
        const v2f *pb = (const v2f *)(hb + (j + 1) * G1S);
        float va[2 * NP2], vb2[2 * NP2];
#pragma unroll
        for (int q = 0; q < NP2; q++)
        {
          v2f ta = pa[q], tb = pb[q];
          va[2 * q] = ta.x, va[2 * q + 1] = ta.y;
          vb2[2 * q] = tb.x, vb2[2 * q + 1] = tb.y;
        }
        float a0 = va[C02] * k20, a1 = va[C02 + 1] * k20;
        float b0 = vb2[C02] * k20, b1 = vb2[C02 + 1] * k20;
#pragma unroll
        for (int i = 1; i < NT2; i++)
        {
          a0 = fmaf(va[C02 + i] + va[C02 - i], a.t2.k[i], a0);
          a1 = fmaf(va[C02 + 1 + i] + va[C02 + 1 - i], a.t2.k[i], a1);
          b0 = fmaf(vb2[C02 + i] + vb2[C02 - i], a.t2.k[i], b0);
          b1 = fmaf(vb2[C02 + 1 + i] + vb2[C02 + 1 - i], a.t2.k[i], b1);
        }
        wv2[2 * R2 + j] = make_float2(a0, a1);
        wv2[2 * R2 + j + 1] = make_float2(b0, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- second filter, vertical: rows yb2 .. yb2+7 of scale s+1
    const int yb2 = yb1 - R2;
    if (yb2 + NR > y0)
    {
      int so_d = yb2 * d2pitch4;
#pragma unroll
      for (int j = 0; j < NR; j++, so_d += d2pitch4)
      {
        if (yb2 + j < y0 || yb2 + j >= y1)
          continue;
        float acc0 = wv2[R2 + j].x * k20, acc1 = wv2[R2 + j].y * k20;
#pragma unroll
        for (int i = 1; i < NT2; i++)
        {
          acc0 = fmaf(wv2[R2 + j + i].x + wv2[R2 + j - i].x, a.t2.k[i], acc0);
          acc1 = fmaf(wv2[R2 + j + i].y + wv2[R2 + j - i].y, a.t2.k[i], acc1);
        }
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(acc0), __float_as_uint(acc1)}, rd2, st_off, so_d, ST_STREAM);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int k = 0; k < 2 * R2; k++)
      wv2[k] = wv2[k + NR];
  }
}

// ---------------------------------------------------------------------------------------------
// k_blur_pair_wide — k_blur_pair with four texels per lane on 256-column strips (round 5): the two window reads per row become
// aligned ds_read_b128 (3 + 5 per row for 5 + 7 taps instead of 6 + 8 ds_read_b64 for half the texels), the scale-s rows go to the
// second LDS buffer and to HBM 16 bytes per lane. A wave computes scale s on 256 columns and owns the inner 256 - 2 HC. Same
// operations per texel in the same order: bit-identical to k_blur_pair and to two k_blur_lean launches.
// ---------------------------------------------------------------------------------------------
template <int NT1, int NT2>
__global__ void __launch_bounds__(64) k_blur_pair_wide(PairArgs a)
{
  constexpr int NR = 8;
  constexpr int R1 = NT1 - 1, R2 = NT2 - 1;
  constexpr int RA1 = (R1 + 3) & ~3, RA2 = (R2 + 3) & ~3;
  constexpr int HC = RA2;
  constexpr int TW = 256, OW = TW - 2 * HC;
  constexpr int SW = TW + 2 * RA1;
  constexpr int NX = RA1 / 2, NXT = NX * NR;
  constexpr int NQ1 = RA1 / 2 + 1, NQ2 = RA2 / 2 + 1;
  constexpr int G1S = TW + 2 * RA2; // row stride of the scale-s rows in LDS: RA2 floats of padding on both sides
  constexpr int NWIN1 = 2 * R1 + NR, NWIN2 = 2 * R2 + NR;
  static_assert(NXT <= 64, "the extra float4 columns of a group are staged by one instruction");
  __shared__ __attribute__((aligned(16))) float s_grp[NR * SW];
  __shared__ __attribute__((aligned(16))) float s_g1[NR * G1S];

  const int lane = threadIdx.x;
  const int W = a.w, H = a.h;
  uint32_t bs = blockIdx.x, bseg = blockIdx.y, bimg = blockIdx.z;
  {
    const uint32_t total = gridDim.x * gridDim.y * gridDim.z;
    const uint32_t b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    uint32_t wi = b;
    if ((total & 7u) == 0)
    {
      const uint32_t per = total >> 3, k = b >> 3;
      wi = (b & 7u) * per + (a.rev ? per - 1u - k : k);
    }
    else if (a.rev)
      wi = total - 1u - b;
    bs = wi % gridDim.x;
    const uint32_t r = wi / gridDim.x;
    bseg = r % gridDim.y;
    bimg = r / gridDim.y;
  }
  const int x0 = (int)bs * OW - HC; // first computed column (virtual: negative on the first strip)
  const int y0 = bseg * a.seg;
  const int y1 = min(y0 + a.seg, H);
  const __amdgpu_buffer_rsrc_t rs = plane_rsrc<false>(a.src, (size_t)bimg * a.src_img_stride, a.spitch, H);
  const __amdgpu_buffer_rsrc_t rd1 = plane_rsrc<false>(a.dst1, (size_t)bimg * a.dst1_img_stride, a.d1pitch, H);
  const __amdgpu_buffer_rsrc_t rd2 = plane_rsrc<false>(a.dst2, (size_t)bimg * a.dst2_img_stride, a.d2pitch, H);
  const int spitch4 = a.spitch * 4, d1pitch4 = a.d1pitch * 4, d2pitch4 = a.d2pitch * 4;

  auto col_off = [&](int gx4, bool &rv) -> unsigned {
    if (gx4 >= 0 && gx4 + 3 < W)
      return (unsigned)gx4 * 4u;
    rv = true;
    return (unsigned)mirror_idx(gx4 + 3, W) * 4u; // the four virtual columns map to m3+3, m3+2, m3+1, m3
  };
  bool rev = false, rev_x = false;
  const unsigned ld_off = col_off(x0 - RA1 + 4 * lane, rev);
  const int xr = lane / NX, xq = 64 + lane % NX;
  unsigned ldx_col = BUF_OOB;
  if (lane < NXT)
    ldx_col = col_off(x0 - RA1 + 4 * xq, rev_x);
  const unsigned ldx_off = lane < NXT ? ldx_col + (unsigned)(xr * spitch4) : BUF_OOB;
  float *const sx = s_grp + xr * SW + 4 * xq;
  const int px = x0 + 4 * lane;
  const bool own = 4 * lane >= HC && 4 * lane + 3 < TW - HC;
  const unsigned st_off = (own && px >= 0 && px + 3 < W) ? (unsigned)px * 4u : BUF_OOB;
  const float k10 = a.t1.k[0], k20 = a.t2.k[0];

  u32x4 pf[NR], pfx;
  auto prefetch = [&](int r0) {
    if (r0 >= 0 && r0 + NR <= H)
    {
      int so = r0 * spitch4;
      pfx = __builtin_amdgcn_raw_buffer_load_b128(rs, ldx_off, so, 0);
#pragma unroll
      for (int j = 0; j < NR; j++, so += spitch4)
        pf[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, ld_off, so, 0);
    }
    else
    {
      const unsigned ox = lane < NXT ? ldx_col + (unsigned)(mirror_idx(r0 + xr, H) * spitch4) : BUF_OOB;
      pfx = __builtin_amdgcn_raw_buffer_load_b128(rs, ox, 0, 0);
#pragma unroll
      for (int j = 0; j < NR; j++)
        pf[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, ld_off, mirror_idx(r0 + j, H) * spitch4, 0);
    }
  };

  int rg = y0 - R1 - R2; // first virtual source row of the current group
  prefetch(rg);

  v4f wv1[NWIN1], wv2[NWIN2];
#pragma unroll
  for (int k = 0; k < NWIN1; k++)
    wv1[k] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NWIN2; k++)
    wv2[k] = v4f{0.f, 0.f, 0.f, 0.f};

  for (; rg - R1 - R2 < y1; rg += NR)
  {
    // ---- stage the prefetched group, then prefetch the next one
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NR; j++)
    {
      u32x4 v = pf[j];
      if (rev)
        v = u32x4{v.w, v.z, v.y, v.x};
      *(u32x4 *)(s_grp + j * SW + 4 * lane) = v;
    }
    if (lane < NXT)
    {
      u32x4 v = pfx;
      if (rev_x)
        v = u32x4{v.w, v.z, v.y, v.x};
      *(u32x4 *)sx = v;
    }
    prefetch(rg + NR);
    __syncthreads();

    // ---- first filter, horizontal: the new rows into the top of window 1
    {
      const v4f *hb = (const v4f *)(s_grp + 4 * lane);
#pragma unroll
      for (int j = 0; j < NR; j++)
      {
        float va[4 * NQ1];
#pragma unroll
        for (int q = 0; q < NQ1; q++)
        {
          const v4f t = hb[j * (SW / 4) + q];
          va[4 * q] = t.x, va[4 * q + 1] = t.y, va[4 * q + 2] = t.z, va[4 * q + 3] = t.w;
        }
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
          o[k] = va[RA1 + k] * k10;
#pragma unroll
        for (int i = 1; i < NT1; i++)
#pragma unroll
          for (int k = 0; k < 4; k++)
            o[k] = fmaf(va[RA1 + k + i] + va[RA1 + k - i], a.t1.k[i], o[k]);
        wv1[2 * R1 + j] = v4f{o[0], o[1], o[2], o[3]};
      }
    }
    // ---- first filter, vertical: rows yb1 .. yb1+7 of scale s (all 256 columns: the second filter's horizontal halo included)
    const int yb1 = rg - R1;
    {
      int so_d = yb1 * d1pitch4;
#pragma unroll
      for (int j = 0; j < NR; j++, so_d += d1pitch4)
      {
        v4f acc = wv1[R1 + j] * k10;
#pragma unroll
        for (int i = 1; i < NT1; i++)
        {
          const v4f sm = wv1[R1 + j + i] + wv1[R1 + j - i];
          acc.x = fmaf(sm.x, a.t1.k[i], acc.x);
          acc.y = fmaf(sm.y, a.t1.k[i], acc.y);
          acc.z = fmaf(sm.z, a.t1.k[i], acc.z);
          acc.w = fmaf(sm.w, a.t1.k[i], acc.w);
        }
        *(v4f *)(s_g1 + j * G1S + RA2 + 4 * lane) = acc;
        // the segment's own rows of scale s (wave-uniform test)
        if (yb1 + j >= y0 && yb1 + j < y1)
          store_b128_stream(u32x4{__float_as_uint(acc.x), __float_as_uint(acc.y), __float_as_uint(acc.z), __float_as_uint(acc.w)}, rd1, st_off, so_d);
      }
    }
#pragma unroll
    for (int k = 0; k < 2 * R1; k++)
      wv1[k] = wv1[k + NR];
    __syncthreads();

    // ---- second filter, horizontal, from the rows of scale s just written to LDS (lanes outside the owned columns read into the
    // padding: their results are never stored)
    {
      const v4f *hb = (const v4f *)(s_g1 + 4 * lane);
#pragma unroll
      for (int j = 0; j < NR; j++)
      {
        float va[4 * NQ2];
#pragma unroll
        for (int q = 0; q < NQ2; q++)
        {
          const v4f t = hb[j * (G1S / 4) + q];
          va[4 * q] = t.x, va[4 * q + 1] = t.y, va[4 * q + 2] = t.z, va[4 * q + 3] = t.w;
        }
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
          o[k] = va[RA2 + k] * k20;
#pragma unroll
        for (int i = 1; i < NT2; i++)
#pragma unroll
          for (int k = 0; k < 4; k++)
            o[k] = fmaf(va[RA2 + k + i] + va[RA2 + k - i], a.t2.k[i], o[k]);
        wv2[2 * R2 + j] = v4f{o[0], o[1], o[2], o[3]};
      }
    }
    // ---- second filter, vertical: rows yb2 .. yb2+7 of scale s+1
    const int yb2 = yb1 - R2;
    if (yb2 + NR > y0)
    {
      int so_d = yb2 * d2pitch4;
#pragma unroll
      for (int j = 0; j < NR; j++, so_d += d2pitch4)
      {
        if (yb2 + j < y0 || yb2 + j >= y1)
          continue;
        v4f acc = wv2[R2 + j] * k20;
#pragma unroll
        for (int i = 1; i < NT2; i++)
        {
          const v4f sm = wv2[R2 + j + i] + wv2[R2 + j - i];
          acc.x = fmaf(sm.x, a.t2.k[i], acc.x);
          acc.y = fmaf(sm.y, a.t2.k[i], acc.y);
          acc.z = fmaf(sm.z, a.t2.k[i], acc.z);
          acc.w = fmaf(sm.w, a.t2.k[i], acc.w);
        }
        store_b128_stream(u32x4{__float_as_uint(acc.x), __float_as_uint(acc.y), __float_as_uint(acc.z), __float_as_uint(acc.w)}, rd2, st_off, so_d);
      }
    }
#pragma unroll
    for (int k = 0; k < 2 * R2; k++)
      wv2[k] = wv2[k + NR];
  }
}

// ---------------------------------------------------------------------------------------------
// k_octave_chain — the WHOLE scale-space of the trailing octaves whose planes fit the LDS, one workgroup per image.
//
// A coarse octave (160x120 and below for a 640x480 input) is a few thousand texels: each of its blur launches costs the
// launch-to-launch latency (~10 us of a single-image detection, 5 launches per octave) to move 77 KB. Here one 1024-thread
// workgroup keeps the current scale (A) and the horizontal-pass temporary (B) of the octave in LDS and walks
// scale 1 .. S+2 of every octave of the run: H pass A -> B, V pass B -> A and to the layer in memory; after scale S every
// thread takes its share of the nearest-neighbour seed of the next octave into registers (dst(x, y) = src(floor((x + .5) * sw / dw), ..):
// k_downsample's rule) and plants it in A when the octave is done. Per texel and pass exactly blur_plane's operations
// (oracle/sift_oracle.c, GaussianBlur*.comp:32-44): acc = c * k0; acc = fma(t(+i) + t(-i), k[i], acc), i ascending, mirrored-repeat
// borders — bit-identical to the per-scale launches.
//   H pass: a thread takes 4 consecutive texels of a row: its 4 + 2 RA inputs are ds_read_b128 in the interior (7 reads per 4
//           outputs at R = 12 instead of 25 per output), mirrored scalar reads in the border groups
//   V pass: a thread takes CH_RUN consecutive rows of one column: CH_RUN + 2 R reads per CH_RUN outputs, lanes on consecutive columns
// fp32 pyramids only; w % 4 == 0; 2 * w * h floats <= the LDS budget; the seed of the next octave <= CH_SEED texels per thread.
// ---------------------------------------------------------------------------------------------
constexpr int CH_MAX_OCT = 4, CH_MAX_LAYERS = 8, CH_THREADS = 1024, CH_RUN = 5, CH_SEED = 5;
constexpr int CH_LDS_FLOATS = 19200 * 2; // 150 KiB of the CU's 160
struct ChainOct
{
  float *layer[CH_MAX_LAYERS]; // image 0
  int w, h, pitch;
  unsigned long long img_stride; // texels
};
struct ChainArgs
{
  ChainOct o[CH_MAX_OCT];
  int n_oct, n_layers, S;
  int nt[CH_MAX_LAYERS];
  float k[CH_MAX_LAYERS][VKSIFT_HIP_MAX_TAPS];
};

// one reflection of the mirrored-repeat addressing: all a run needs when every filter radius is below the plane's sides (checked by the launcher)
__device__ __forceinline__ int mirror1(int i, int n) { return i < 0 ? -1 - i : (i >= n ? 2 * n - 1 - i : i); }
// it / d for 0 <= it < 2^20 and 1 <= d <= 2^10 without an integer division: (it + 0.5) / d is at least 0.5 / d away from an integer
__device__ __forceinline__ int div_small(int it, float inv_d) { return (int)(((float)it + 0.5f) * inv_d); }

template <int NT>
__device__ __forceinline__ void chain_h(const float *__restrict__ A, float *__restrict__ B, int w, int h, const float *__restrict__ k)
{
  constexpr int R = NT - 1, RA = (R + 3) & ~3, NV = 4 + 2 * RA, NB = RA / 4; // NB groups at each end of a row need mirrored inputs
  const int groups = w >> 2;
  const int gi = groups > 2 * NB ? groups - 2 * NB : 0; // interior groups per row
  auto finish = [&](const float *v, int row, int x0) {
    v4f out;
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
      float acc = v[RA + j] * k[0];
#pragma unroll
      for (int i = 1; i <= R; i++)
        acc = fmaf(v[RA + j + i] + v[RA + j - i], k[i], acc);
      out[j] = acc;
    }
    *(v4f *)(B + row * w + x0) = out;
  };
  // interior groups and border groups are separate index spaces: a wave never runs both input paths
  if (gi > 0)
  {
    const float inv = 1.f / (float)gi;
    for (int it = threadIdx.x; it < gi * h; it += CH_THREADS)
    {
      const int row = div_small(it, inv), x0 = (NB + it - row * gi) * 4;
      const float *p = A + row * w + x0 - RA;
      float v[NV];
#pragma unroll
      for (int q = 0; q < NV / 4; q++)
      {
        const v4f t = *(const v4f *)(p + 4 * q);
        v[4 * q] = t.x, v[4 * q + 1] = t.y, v[4 * q + 2] = t.z, v[4 * q + 3] = t.w;
      }
      finish(v, row, x0);
    }
  }
  {
    const int nb = groups - gi;
    const float inv = 1.f / (float)nb;
    for (int it = threadIdx.x; it < nb * h; it += CH_THREADS)
    {
      const int row = div_small(it, inv), b = it - row * nb;
      const int x0 = (gi > 0 && b >= NB ? gi + b : b) * 4;
      const float *p = A + row * w;
      float v[NV];
#pragma unroll
      for (int q = RA - R; q < NV - (RA - R); q++)
        v[q] = p[mirror1(x0 - RA + q, w)];
      finish(v, row, x0);
    }
  }
}

template <int NT>
__device__ __forceinline__ void chain_v(const float *__restrict__ B, float *__restrict__ A, float *__restrict__ out, int pitch, int w, int h,
                                        const float *__restrict__ k)
{
  constexpr int R = NT - 1, NV = CH_RUN + 2 * R;
  const int runs = (h + CH_RUN - 1) / CH_RUN, items = runs * w;
  const float inv = 1.f / (float)w;
  for (int it = threadIdx.x; it < items; it += CH_THREADS)
  {
    const int run = div_small(it, inv), x = it - run * w, r0 = run * CH_RUN;
    float v[NV];
    if (r0 - R >= 0 && r0 + CH_RUN + R <= h) // (a wave spans at most two runs)
    {
#pragma unroll
      for (int q = 0; q < NV; q++)
        v[q] = B[(r0 - R + q) * w + x];
    }
    else
    {
#pragma unroll
      for (int q = 0; q < NV; q++)
        v[q] = B[mirror1(min(r0 - R + q, h - 1 + R), h) * w + x]; // rows past h + R - 1 feed only outputs past the last row
    }
#pragma unroll
    for (int j = 0; j < CH_RUN; j++)
    {
      float acc = v[R + j] * k[0];
#pragma unroll
      for (int i = 1; i <= R; i++)
        acc = fmaf(v[R + j + i] + v[R + j - i], k[i], acc);
      if (r0 + j < h)
      {
        A[(r0 + j) * w + x] = acc;
        out[(size_t)(r0 + j) * pitch + x] = acc;
      }
    }
  }
}

// any tap count: straight from LDS, one texel per step (configurations whose kernels are not instantiated below)
__device__ __forceinline__ void chain_h_any(const float *__restrict__ A, float *__restrict__ B, int w, int h, const float *__restrict__ k, int nt)
{
  for (int it = threadIdx.x; it < w * h; it += CH_THREADS)
  {
    const int row = it / w, x = it - row * w;
    const float *p = A + row * w;
    float acc = p[x] * k[0];
    for (int i = 1; i < nt; i++)
      acc = fmaf(p[mirror_idx(x + i, w)] + p[mirror_idx(x - i, w)], k[i], acc);
    B[it] = acc;
  }
}
__device__ __forceinline__ void chain_v_any(const float *__restrict__ B, float *__restrict__ A, float *__restrict__ out, int pitch, int w, int h,
                                            const float *__restrict__ k, int nt)
{
  for (int it = threadIdx.x; it < w * h; it += CH_THREADS)
  {
    const int row = it / w, x = it - row * w;
    float acc = B[it] * k[0];
    for (int i = 1; i < nt; i++)
      acc = fmaf(B[mirror_idx(row + i, h) * w + x] + B[mirror_idx(row - i, h) * w + x], k[i], acc);
    A[it] = acc;
    out[(size_t)row * pitch + x] = acc;
  }
}

__global__ void __launch_bounds__(CH_THREADS) k_octave_chain(ChainArgs a)
{
  extern __shared__ float lds[];
  float *A = lds, *B = lds + a.o[0].w * a.o[0].h;
  const int tid = threadIdx.x;
  float seed[CH_SEED];
  {
    // scale 0 of the first octave of the run is in memory (written by the launch that blurred scale S of the octave before it)
    const ChainOct &o = a.o[0];
    const float *src = o.layer[0] + (size_t)blockIdx.x * o.img_stride;
    const int groups = o.w >> 2;
    for (int it = tid; it < groups * o.h; it += CH_THREADS)
    {
      const int row = it / groups, x0 = (it - row * groups) * 4;
      *(v4f *)(A + row * o.w + x0) = *(const v4f *)(src + (size_t)row * o.pitch + x0);
    }
  }
  for (int oi = 0; oi < a.n_oct; oi++)
  {
    const ChainOct &o = a.o[oi];
    const int w = o.w, h = o.h;
    if (oi > 0)
    {
      // plant the seed taken at scale S of the octave above (and store it: the keypoint stages read layer 0 too)
      float *l0 = o.layer[0] + (size_t)blockIdx.x * o.img_stride;
#pragma unroll
      for (int q = 0; q < CH_SEED; q++)
      {
        const int idx = tid + q * CH_THREADS;
        if (idx < w * h)
        {
          const int y = idx / w, x = idx - y * w;
          A[idx] = seed[q];
          l0[(size_t)y * o.pitch + x] = seed[q];
        }
      }
    }
    __syncthreads();
    for (int s = 1; s < a.n_layers; s++)
    {
      const float *k = a.k[s];
      float *out = o.layer[s] + (size_t)blockIdx.x * o.img_stride;
      const int nt = a.nt[s];
      switch (nt)
      {
      case 5: chain_h<5>(A, B, w, h, k); break;
      case 7: chain_h<7>(A, B, w, h, k); break;
      case 9: chain_h<9>(A, B, w, h, k); break;
      case 11: chain_h<11>(A, B, w, h, k); break;
      case 13: chain_h<13>(A, B, w, h, k); break;
      default: chain_h_any(A, B, w, h, k, nt); break;
      }
      __syncthreads();
      switch (nt)
      {
      case 5: chain_v<5>(B, A, out, o.pitch, w, h, k); break;
      case 7: chain_v<7>(B, A, out, o.pitch, w, h, k); break;
      case 9: chain_v<9>(B, A, out, o.pitch, w, h, k); break;
      case 11: chain_v<11>(B, A, out, o.pitch, w, h, k); break;
      case 13: chain_v<13>(B, A, out, o.pitch, w, h, k); break;
      default: chain_v_any(B, A, out, o.pitch, w, h, k, nt); break;
      }
      __syncthreads();
      if (s == a.S && oi + 1 < a.n_oct)
      {
        const int dw = a.o[oi + 1].w, dh = a.o[oi + 1].h;
        const float sx = (float)w / (float)dw, sy = (float)h / (float)dh;
#pragma unroll
        for (int q = 0; q < CH_SEED; q++)
        {
          const int idx = tid + q * CH_THREADS;
          seed[q] = 0.f;
          if (idx < dw * dh)
          {
            const int y = idx / dw, x = idx - y * dw;
            const int yy = clampi((int)floorf(((float)y + 0.5f) * sy), 0, h - 1);
            const int xx = clampi((int)floorf(((float)x + 0.5f) * sx), 0, w - 1);
            seed[q] = A[yy * w + xx];
          }
        }
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Nearest-neighbour resample (2:1 -> odd source texels), one thread per destination pixel.
// ---------------------------------------------------------------------------------------------
template <bool F16>
__global__ void __launch_bounds__(256) k_downsample(const float *__restrict__ src, uint64_t src_img_stride, int sw, int sh, int spitch,
                                                    float *__restrict__ dst, uint64_t dst_img_stride, int dw, int dh, int dpitch)
{
  int x = blockIdx.x * 64 + (threadIdx.x & 63);
  int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh)
    return;
  float sx = (float)sw / (float)dw, sy = (float)sh / (float)dh;
  int yy = clampi((int)floorf(((float)y + 0.5f) * sy), 0, sh - 1);
  int xx = clampi((int)floorf(((float)x + 0.5f) * sx), 0, sw - 1);
  const float *in = img_ptr<F16>(src, (size_t)blockIdx.z * src_img_stride);
  float *out = img_ptr<F16>(dst, (size_t)blockIdx.z * dst_img_stride);
  st_px<F16>(out, (size_t)y * dpitch + x, ld_px<F16>(in, (size_t)yy * spitch + xx)); // widening + narrowing a binary16 value is the identity
}

// DifferenceOfGaussian.comp:13-17 for ONE layer of ONE image, dense w x h output: only vksift_downloadDoGImage needs a DoG plane
// in memory (the detection path forms the differences in registers).
// hi == nullptr: the Gaussian layer `lo` itself, widened to fp32 (vksift_downloadScaleSpaceImage of an fp16 pyramid).
template <bool F16>
__global__ void __launch_bounds__(256) k_dog_plane(const float *__restrict__ lo, const float *__restrict__ hi, int w, int h, int pitch, float *__restrict__ out)
{
  int x = blockIdx.x * 64 + (threadIdx.x & 63);
  int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h)
    return;
  const float l = ld_px<F16>(lo, (size_t)y * pitch + x);
  float d = hi ? ld_px<F16>(hi, (size_t)y * pitch + x) - l : l;
  if (F16 && hi)
    d = (float)to_h(d); // the DoG image of an fp16 pyramid is a binary16 image too
  out[(size_t)y * w + x] = d;
}

} // namespace

extern "C"
{

  int vksift_hip_input_blit(const uint8_t *src, uint32_t sw, uint32_t sh, uint64_t src_img_stride, vksift_hip_Plane dst, uint32_t batch, vksift_hip_stream s)
  {
    if (dst.w == 2 * sw && dst.h == 2 * sh)
    {
      dim3 grid2(((dst.w + 3) / 4 + 63) / 64, (dst.h + 3) / 4, batch);
      if (dst.fp16)
        hipLaunchKernelGGL(k_input_blit_2x<true>, grid2, dim3(256), 0, (hipStream_t)s, src, (int)sw, (int)sh, src_img_stride, dst.base, (int)dst.w, (int)dst.h,
                           (int)dst.pitch, dst.img_stride);
      else
        hipLaunchKernelGGL(k_input_blit_2x<false>, grid2, dim3(256), 0, (hipStream_t)s, src, (int)sw, (int)sh, src_img_stride, dst.base, (int)dst.w, (int)dst.h,
                           (int)dst.pitch, dst.img_stride);
      return (int)hipGetLastError();
    }
    dim3 grid((dst.w + 63) / 64, (dst.h + 3) / 4, batch);
    if (dst.fp16)
      hipLaunchKernelGGL(k_input_blit<true>, grid, dim3(256), 0, (hipStream_t)s, src, (int)sw, (int)sh, src_img_stride, dst.base, (int)dst.w, (int)dst.h,
                         (int)dst.pitch, dst.img_stride);
    else
      hipLaunchKernelGGL(k_input_blit<false>, grid, dim3(256), 0, (hipStream_t)s, src, (int)sw, (int)sh, src_img_stride, dst.base, (int)dst.w, (int)dst.h,
                         (int)dst.pitch, dst.img_stride);
    return (int)hipGetLastError();
  }

  static int blur_tile_launch(vksift_hip_Plane src, vksift_hip_Plane dst, const Taps &t, uint32_t ntaps, uint32_t batch, vksift_hip_stream s)
  {
    int R = (int)ntaps - 1;
    int SH = TILE + 2 * R, SS = TILE + 2 * R + 1;
    size_t lds_bytes = sizeof(float) * ((size_t)SH * SS + (size_t)SH * TILE);
    static bool lds_attr_set = false;
    if (!lds_attr_set)
    {
      /* largest tile (R = 19) needs 68 KiB of the CU's 160 KiB LDS: above the 64 KiB default opt-in limit */
      const int Rm = VKSIFT_HIP_MAX_TAPS - 1;
      const int max_bytes = (int)(sizeof(float) * ((TILE + 2 * Rm) * (TILE + 2 * Rm + 1) + (TILE + 2 * Rm) * TILE));
      hipError_t ae = hipFuncSetAttribute((const void *)k_blur_tile<false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_bytes);
      if (ae == hipSuccess)
        ae = hipFuncSetAttribute((const void *)k_blur_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_bytes);
      if (ae != hipSuccess)
        return (int)ae;
      lds_attr_set = true;
    }
    dim3 grid((src.w + TILE - 1) / TILE, (src.h + TILE - 1) / TILE, batch);
    if (src.fp16)
      hipLaunchKernelGGL(k_blur_tile<true>, grid, dim3(256), lds_bytes, (hipStream_t)s, src.base, src.img_stride, (int)src.pitch, dst.base, dst.img_stride,
                         (int)dst.pitch, (int)src.w, (int)src.h, t, (int)ntaps);
    else
      hipLaunchKernelGGL(k_blur_tile<false>, grid, dim3(256), lds_bytes, (hipStream_t)s, src.base, src.img_stride, (int)src.pitch, dst.base, dst.img_stride,
                         (int)dst.pitch, (int)src.w, (int)src.h, t, (int)ntaps);
    return (int)hipGetLastError();
  }

  /* Row segments of the streaming kernel: enough workgroups to give every CU ~40 waves over the launch (2560 long-lived waves
   * left the slowest CU to set the time), but segments long enough that the 2R-row warm-up stays a small fraction; launches
   * that cannot fill the GPU anyway (small octaves, small batches) are latency bound and take shorter marches. */
  /* shortest march (output rows per wave) a launch of this size is cut into: launches that cannot fill the chip are bound by the length of
   * one wave's march (2R warm-up rows + its own), not by bandwidth */
  static uint32_t march_rows(uint32_t waves64)
  {
    const int t = vksift_hip_tune_get(VKSIFT_TUNE_MIN_MARCH);
    /* (one 640x480 image: 8-row marches 0.302-0.313 ms against 0.314-0.326 with 16 rows, four alternations of 4 000 detections each) */
    const uint32_t smallest = t > 0 ? (uint32_t)t : 8u;
    return waves64 >= 2048u ? 64u : (waves64 >= 512u ? 32u : (waves64 >= 192u ? 16u : smallest));
  }

  static dim3 stream_grid(uint32_t w, uint32_t h, uint32_t batch, uint32_t wg_target, int *seg_out, uint32_t strip_w = 128u)
  {
    const uint32_t strips = (w + strip_w - 1u) / strip_w;
    if (vksift_hip_tune_get(VKSIFT_TUNE_WG_TARGET) > 0)
      wg_target = (uint32_t)vksift_hip_tune_get(VKSIFT_TUNE_WG_TARGET);
    uint32_t nseg = (wg_target + strips * batch - 1u) / (strips * batch);
    const uint32_t waves64 = strips * batch * ((h + 63u) / 64u);
    const uint32_t seg_rows = march_rows(waves64);
    const uint32_t max_seg = (h + seg_rows - 1u) / seg_rows;
    if (nseg > max_seg)
      nseg = max_seg;
    if (nseg < 1)
      nseg = 1;
    const uint32_t seg = ((h + nseg - 1u) / nseg + 7u) & ~7u;
    nseg = (h + seg - 1u) / seg;
    *seg_out = (int)seg;
    return dim3(strips, nseg, batch);
  }

  constexpr int WIDE_DEFAULT_MASK = (1 << 9) | (1 << 11) | (1 << 13);
  /* which kernel a blur of this shape takes: 0 the generic tile kernel, 1 the two-texel strip march (k_blur_lean), 2 the four-texel one (k_blur_wide) */
  static int blur_form(const vksift_hip_Plane &src, const vksift_hip_Plane &dst, uint32_t ntaps, uint32_t batch)
  {
    const uint32_t ra = ((ntaps - 1u) + 3u) & ~3u;
    const uint32_t nstrips = (src.w + 127u) / 128u;
    const bool lean = ntaps >= 2 && (src.w % 4u) == 0 && ra <= src.w && nstrips * 128u + ra <= 2u * src.w;
    if (!lean)
      return 0;
    /* four texels per lane on 256-column strips (k_blur_wide): fp32 planes whose width wastes little of the last strip */
    const int wide_mask = vksift_hip_tune_get(VKSIFT_TUNE_WIDE_MASK) >= 0 ? vksift_hip_tune_get(VKSIFT_TUNE_WIDE_MASK) : WIDE_DEFAULT_MASK;
    const uint32_t wstrips = (src.w + 255u) / 256u;
    /* (launches that cannot fill the chip — a single image, the coarse octaves of a small batch — are latency bound and want the larger
     * number of shorter-lived waves the 128-column strips give them: one 640x480 image 0.355 -> 0.378 ms with wide strips everywhere) */
    const bool fills = (uint64_t)wstrips * batch * ((src.h + 63u) / 64u) >= 2048u || vksift_hip_tune_get(VKSIFT_TUNE_WIDE_MASK) >= 0;
    if (fills && !src.fp16 && !dst.fp16 && ((wide_mask >> ntaps) & 1) && (src.w % 4u) == 0 && ra <= src.w && wstrips * 256u + ra <= 2u * src.w &&
        wstrips * 256u - src.w <= 64u && ((src.pitch | dst.pitch) & 3u) == 0)
      return 2;
    return 1;
  }
  int vksift_hip_blur_form(vksift_hip_Plane src, vksift_hip_Plane dst, uint32_t ntaps, uint32_t batch) { return blur_form(src, dst, ntaps, batch); }

  static int blur_impl(vksift_hip_Plane src, vksift_hip_Plane dst, vksift_hip_Plane ds, const float *taps, uint32_t ntaps, uint32_t batch, vksift_hip_stream s)
  {
    if (ntaps < 1 || ntaps > VKSIFT_HIP_MAX_TAPS || src.base == dst.base || dst.base == NULL)
      return (int)hipErrorInvalidValue;
    Taps t;
    for (uint32_t i = 0; i < VKSIFT_HIP_MAX_TAPS; i++)
      t.k[i] = i < ntaps ? taps[i] : 0.f;
    static int force_tile = -1;
    if (force_tile < 0)
    {
      const char *e = getenv("VKSIFT_BLUR_KERNEL"); /* "tile": the generic fallback kernel everywhere (debugging) */
      force_tile = (e && e[0] == 't') ? 1 : 0;
    }
    const uint32_t ra = ((ntaps - 1u) + 3u) & ~3u;
    const int form = force_tile ? 0 : blur_form(src, dst, ntaps, batch);
    if (form == 0)
      return ds.base ? -1 : blur_tile_launch(src, dst, t, ntaps, batch, s);

    StreamArgs a;
    a.ds = ds.base, a.ds_img_stride = ds.img_stride, a.ds_pitch = (int)ds.pitch;
    a.src = src.base, a.dst = dst.base;
    a.src_img_stride = src.img_stride, a.dst_img_stride = dst.img_stride;
    a.spitch = (int)src.pitch, a.dpitch = (int)dst.pitch;
    a.w = (int)src.w, a.h = (int)src.h;
    a.rev = (int)dst.reverse;
    a.taps = t;
    hipStream_t hs = (hipStream_t)s;
    (void)ra;
    if (form == 2)
    {
      /* (from 11 taps on half as many, twice as long marches: 512 x 1280x960, tools/blur_ab.py: 11 taps 928 -> 905 us, 13 taps 988 -> 939 us) */
      const dim3 wgrid = stream_grid(src.w, src.h, batch, ntaps >= 11u ? 5120u : 10240u, &a.seg, 256u);
      switch (ntaps)
      {
#define VKSIFT_CASE(N)                                              \
  case N:                                                           \
    hipLaunchKernelGGL((k_blur_wide<N>), wgrid, dim3(64), 0, hs, a); \
    return (int)hipGetLastError();
        VKSIFT_CASE(5) VKSIFT_CASE(7) VKSIFT_CASE(9) VKSIFT_CASE(11) VKSIFT_CASE(13)
#undef VKSIFT_CASE
      default:
        break;
      }
    }
    /* (11 taps and more: fewer, longer marches — the 2R-row warm-up of a 13-tap segment is 12 rows, and these launches are the VALU
     * co-limited ones. 512 x 640x480 planes, tools/blur_ab.py: 11 taps 267 -> 256 us, 13 taps 301 -> 287 us with 240-row instead of 120-row
     * segments; 320x240: 13 taps 95 -> 87 us; 9 taps and fewer: no difference) */
    const dim3 grid = stream_grid(src.w, src.h, batch, ntaps >= 11u ? 3072u : 10240u, &a.seg);
    switch (ntaps)
    {
#define VKSIFT_CASE(N)                                                        \
  case N:                                                                     \
    if (src.fp16)                                                             \
      hipLaunchKernelGGL((k_blur_lean<N, 0, true>), grid, dim3(64), 0, hs, a);  \
    else                                                                      \
      hipLaunchKernelGGL((k_blur_lean<N, 0, false>), grid, dim3(64), 0, hs, a); \
    break;
      VKSIFT_CASE(2) VKSIFT_CASE(3) VKSIFT_CASE(4) VKSIFT_CASE(5) VKSIFT_CASE(6) VKSIFT_CASE(7) VKSIFT_CASE(8) VKSIFT_CASE(9) VKSIFT_CASE(10)
      VKSIFT_CASE(11) VKSIFT_CASE(12) VKSIFT_CASE(13) VKSIFT_CASE(14) VKSIFT_CASE(15) VKSIFT_CASE(16) VKSIFT_CASE(17) VKSIFT_CASE(18)
      VKSIFT_CASE(19) VKSIFT_CASE(20)
#undef VKSIFT_CASE
    default:
      return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
  }

  /* One scale of n octaves (src[i] -> dst[i], same taps) in ONE launch of the two-texel strip march; -1 (nothing launched) when a plane is
   * not covered by that kernel, the texel types differ or the tap count has no multi-octave instantiation: the caller then takes
   * vksift_hip_blur per plane. Bit-identical to those launches (the same kernel body). */
  int vksift_hip_blur_multi(const vksift_hip_Plane *src, const vksift_hip_Plane *dst, uint32_t n, const float *taps, uint32_t ntaps, uint32_t batch,
                            vksift_hip_stream s)
  {
    if (n == 0 || batch == 0)
      return 0;
    if (n > (uint32_t)MULTI_MAX || ntaps < 2 || ntaps > VKSIFT_HIP_MAX_TAPS || getenv("VKSIFT_BLUR_KERNEL"))
      return -1;
    if (ntaps != 9 && ntaps != 11 && ntaps != 13 && ntaps != 15)
      return -1;
    Multi<StreamArgs> m;
    m.n = 0;
    const uint32_t ra = ((ntaps - 1u) + 3u) & ~3u;
    for (uint32_t i = 0; i < n; i++)
    {
      const vksift_hip_Plane &p = src[i], &d = dst[i];
      const uint32_t nstrips = (p.w + 127u) / 128u;
      if (p.base == NULL || d.base == NULL || p.base == d.base || p.fp16 != src[0].fp16 || d.fp16 != src[0].fp16 || d.w != p.w || d.h != p.h || (p.w % 4u) != 0 ||
          ra > p.w || nstrips * 128u + ra > 2u * p.w)
        return -1;
      StreamArgs a;
      a.ds = NULL, a.ds_img_stride = 0, a.ds_pitch = 0;
      a.src = p.base, a.dst = d.base;
      a.src_img_stride = p.img_stride, a.dst_img_stride = d.img_stride;
      a.spitch = (int)p.pitch, a.dpitch = (int)d.pitch;
      a.w = (int)p.w, a.h = (int)p.h;
      a.rev = (int)d.reverse;
      for (uint32_t k = 0; k < VKSIFT_HIP_MAX_TAPS; k++)
        a.taps.k[k] = k < ntaps ? taps[k] : 0.f;
      const dim3 g = stream_grid(p.w, p.h, batch, ntaps >= 11u ? 3072u : 10240u, &a.seg);
      if (!multi_add(m, a, g.x, g.y, g.z))
        return -1;
    }
    const dim3 grid(m.start[m.n]);
    const bool f16 = src[0].fp16 != 0;
    switch (ntaps)
    {
#define VKSIFT_CASE(N)                                                                      \
  case N:                                                                                   \
    if (f16)                                                                                \
      hipLaunchKernelGGL((k_blur_lean_multi<N, true>), grid, dim3(64), 0, (hipStream_t)s, m);  \
    else                                                                                    \
      hipLaunchKernelGGL((k_blur_lean_multi<N, false>), grid, dim3(64), 0, (hipStream_t)s, m); \
    break;
      VKSIFT_CASE(9) VKSIFT_CASE(11) VKSIFT_CASE(13) VKSIFT_CASE(15)
#undef VKSIFT_CASE
    default:
      return -1;
    }
    return (int)hipGetLastError();
  }

  int vksift_hip_blur_pair(vksift_hip_Plane src, vksift_hip_Plane dst1, vksift_hip_Plane dst2, const float *taps1, uint32_t ntaps1, const float *taps2,
                           uint32_t ntaps2, uint32_t batch, vksift_hip_stream s)
  {
    static int pair_env = -1;
    if (pair_env < 0)
    {
      const char *e = getenv("VKSIFT_BLUR_PAIR"); /* 0: every scale its own launch (A/B runs, tests) */
      pair_env = e ? atoi(e) : 1;
    }
    const bool combo = ntaps1 == 5 && ntaps2 == 7; /* scales 1 and 2 of the default configuration (3 scales per octave, sampler-interpolated taps) */
    if (!pair_env || !combo || src.fp16 || dst1.fp16 || dst2.fp16 || src.base == NULL || dst1.base == NULL || dst2.base == NULL || getenv("VKSIFT_BLUR_KERNEL"))
      return -1;
    const uint32_t W = src.w, H = src.h;
    const uint32_t r2 = ntaps2 - 1u, hc = (r2 + 3u) & ~3u, ow = 128u - 2u * hc, ra1 = ((ntaps1 - 1u) + 3u) & ~3u;
    const uint32_t strips = (W + ow - 1u) / ow;
    /* one mirror reflection has to cover every staged column and every virtual row of a march */
    if ((W % 4u) != 0 || hc + ra1 > W || (strips - 1u) * ow + 128u + ra1 > 2u * W + hc || H < 64u || dst1.w != W || dst2.w != W || dst1.h != H || dst2.h != H)
      return -1;
    {
      /* four texels per lane on 256-column strips (k_blur_pair_wide) for launches that fill the chip */
      const uint32_t hcw = (r2 + 3u) & ~3u, oww = 256u - 2u * hcw, wstrips = (W + oww - 1u) / oww;
      const int pw = vksift_hip_tune_get(VKSIFT_TUNE_PAIR_FORM); /* 0: built-in, 1: two texels per lane, 2: four */
      /* ... of planes at least 1024 texels wide: on 512 x 640x480 planes the two-texel form is the faster one (377 against 387 us), on 320x240
       * by 9 % (109 / 119 us; tools/blur_ab.py) */
      const bool fills = (uint64_t)wstrips * batch * ((H + 63u) / 64u) >= 2048u && W >= 1024u;
      if (pw != 1 && (fills || pw == 2) && hcw + ra1 <= W && (wstrips - 1u) * oww + 256u + ra1 <= 2u * W + hcw && ((src.pitch | dst1.pitch | dst2.pitch) & 3u) == 0)
      {
        PairArgs aw;
        aw.src = src.base, aw.dst1 = dst1.base, aw.dst2 = dst2.base;
        aw.src_img_stride = src.img_stride, aw.dst1_img_stride = dst1.img_stride, aw.dst2_img_stride = dst2.img_stride;
        aw.spitch = (int)src.pitch, aw.d1pitch = (int)dst1.pitch, aw.d2pitch = (int)dst2.pitch;
        aw.w = (int)W, aw.h = (int)H;
        aw.rev = (int)dst2.reverse;
        for (uint32_t i = 0; i < VKSIFT_HIP_MAX_TAPS; i++)
          aw.t1.k[i] = i < ntaps1 ? taps1[i] : 0.f, aw.t2.k[i] = i < ntaps2 ? taps2[i] : 0.f;
        uint32_t nsegw = (10240u + wstrips * batch - 1u) / (wstrips * batch);
        const uint32_t max_segw = (H + 63u) / 64u;
        nsegw = nsegw > max_segw ? max_segw : (nsegw < 1u ? 1u : nsegw);
        const uint32_t segw = ((H + nsegw - 1u) / nsegw + 7u) & ~7u;
        nsegw = (H + segw - 1u) / segw;
        aw.seg = (int)segw;
        hipLaunchKernelGGL((k_blur_pair_wide<5, 7>), dim3(wstrips, nsegw, batch), dim3(64), 0, (hipStream_t)s, aw);
        return (int)hipGetLastError();
      }
    }
    PairArgs a;
    a.src = src.base, a.dst1 = dst1.base, a.dst2 = dst2.base;
    a.src_img_stride = src.img_stride, a.dst1_img_stride = dst1.img_stride, a.dst2_img_stride = dst2.img_stride;
    a.spitch = (int)src.pitch, a.d1pitch = (int)dst1.pitch, a.d2pitch = (int)dst2.pitch;
    a.w = (int)W, a.h = (int)H;
    a.rev = (int)dst2.reverse;
    for (uint32_t i = 0; i < VKSIFT_HIP_MAX_TAPS; i++)
      a.t1.k[i] = i < ntaps1 ? taps1[i] : 0.f, a.t2.k[i] = i < ntaps2 ? taps2[i] : 0.f;
    /* row segments as stream_grid(): the strip count differs (128 - 2 HC owned columns per wave) */
    uint32_t nseg = (10240u + strips * batch - 1u) / (strips * batch);
    /* launches that cannot fill the GPU (a single image, the coarse octaves of a small batch) are latency bound: shorter marches,
     * as stream_grid() — one 640x480 image: 4 pair launches of 27 us each with 64-row segments */
    const uint32_t waves64 = strips * batch * ((H + 63u) / 64u);
    const uint32_t seg_rows = march_rows(waves64);
    const uint32_t max_seg = (H + seg_rows - 1u) / seg_rows;
    if (nseg > max_seg)
      nseg = max_seg;
    if (nseg < 1)
      nseg = 1;
    const uint32_t seg = ((H + nseg - 1u) / nseg + 7u) & ~7u;
    nseg = (H + seg - 1u) / seg;
    a.seg = (int)seg;
    const dim3 grid(strips, nseg, batch);
    hipLaunchKernelGGL((k_blur_pair<5, 7>), grid, dim3(64), 0, (hipStream_t)s, a);
    return (int)hipGetLastError();
  }

  int vksift_hip_blur(vksift_hip_Plane src, vksift_hip_Plane dst, const float *taps, uint32_t ntaps, uint32_t batch, vksift_hip_stream s)
  {
    const vksift_hip_Plane none = {NULL, 0, 0, 0, 0, 0, 0};
    return blur_impl(src, dst, none, taps, ntaps, batch, s);
  }

  int vksift_hip_blur_downsample(vksift_hip_Plane src, vksift_hip_Plane dst, vksift_hip_Plane next, const float *taps, uint32_t ntaps, uint32_t batch,
                                 vksift_hip_stream s)
  {
    if (next.base == NULL || dst.base == NULL || next.w * 2u != src.w || next.h * 2u != src.h)
      return -1;
    return blur_impl(src, dst, next, taps, ntaps, batch, s);
  }

  int vksift_hip_seed_upsampled(const uint8_t *src, uint32_t sw, uint32_t sh, uint64_t src_img_stride, vksift_hip_Plane dst, const float *taps, uint32_t ntaps,
                                uint32_t batch, vksift_hip_stream s)
  {
    const uint32_t W = dst.w, H = dst.h;
    const uint32_t ra = ((ntaps - 1u) + 3u) & ~3u;
    const uint32_t strips = (W + 127u) / 128u;
    if (ntaps < 2 || ntaps > 12 || W != 2 * sw || H != 2 * sh || (W % 4u) != 0 || sw < 4 || ra > W || strips * 128u + ra > 2u * W)
      return -1; /* not applicable: the caller runs vksift_hip_input_blit + vksift_hip_blur */
    StreamArgs a;
    a.ds = NULL, a.ds_img_stride = 0, a.ds_pitch = 0;
    a.src = (const float *)src, a.dst = dst.base;
    a.src_img_stride = src_img_stride, a.dst_img_stride = dst.img_stride;
    a.spitch = (int)sw, a.dpitch = (int)dst.pitch;
    a.w = (int)W, a.h = (int)H;
    a.rev = (int)dst.reverse;
    for (uint32_t i = 0; i < VKSIFT_HIP_MAX_TAPS; i++)
      a.taps.k[i] = i < ntaps ? taps[i] : 0.f;
    const uint32_t seed_wg = vksift_hip_tune_get(VKSIFT_TUNE_SEED_WG) > 0 ? (uint32_t)vksift_hip_tune_get(VKSIFT_TUNE_SEED_WG) : 10240u;
    uint32_t nseg = (seed_wg + strips * batch - 1u) / (strips * batch); /* as the other launches (stream_grid): 2560 long-lived waves left the tail to a few CUs */
    const uint32_t waves64 = strips * batch * ((H + 63u) / 64u);
    const uint32_t seg_rows = march_rows(waves64); /* latency-bound launches: shorter marches (stream_grid) */
    uint32_t max_seg = (H + seg_rows - 1u) / seg_rows;
    if (nseg > max_seg)
      nseg = max_seg;
    if (nseg < 1)
      nseg = 1;
    uint32_t seg = ((H + nseg - 1u) / nseg + 7u) & ~7u;
    nseg = (H + seg - 1u) / seg;
    a.seg = (int)seg;
    dim3 grid(strips, nseg, batch);
    switch (ntaps)
    {
#define VKSIFT_CASE(N)                                                                        \
  case N:                                                                                     \
    if (dst.fp16)                                                                             \
      hipLaunchKernelGGL((k_blur_lean<N, 1, true>), grid, dim3(64), 0, (hipStream_t)s, a);   \
    else                                                                                      \
      hipLaunchKernelGGL((k_blur_lean<N, 1, false>), grid, dim3(64), 0, (hipStream_t)s, a);  \
    break;
      VKSIFT_CASE(2) VKSIFT_CASE(3) VKSIFT_CASE(4) VKSIFT_CASE(5) VKSIFT_CASE(6) VKSIFT_CASE(7) VKSIFT_CASE(8) VKSIFT_CASE(9) VKSIFT_CASE(10)
      VKSIFT_CASE(11) VKSIFT_CASE(12)
#undef VKSIFT_CASE
    default:
      return -1;
    }
    return (int)hipGetLastError();
  }

  int vksift_hip_seed_direct(const uint8_t *src, uint32_t sw, uint32_t sh, uint64_t src_img_stride, vksift_hip_Plane dst, const float *taps, uint32_t ntaps,
                             uint32_t batch, vksift_hip_stream s)
  {
    const uint32_t W = dst.w, H = dst.h;
    const uint32_t ra = ((ntaps - 1u) + 3u) & ~3u;
    const uint32_t strips = (W + 127u) / 128u;
    if (ntaps < 2 || ntaps > VKSIFT_HIP_MAX_TAPS || W != sw || H != sh || (W % 4u) != 0 || ra > W || strips * 128u + ra > 2u * W)
      return -1; /* not applicable: the caller runs vksift_hip_input_blit + vksift_hip_blur */
    StreamArgs a;
    a.ds = NULL, a.ds_img_stride = 0, a.ds_pitch = 0;
    a.src = (const float *)src, a.dst = dst.base;
    a.src_img_stride = src_img_stride, a.dst_img_stride = dst.img_stride;
    a.spitch = (int)sw, a.dpitch = (int)dst.pitch;
    a.w = (int)W, a.h = (int)H;
    a.rev = (int)dst.reverse;
    for (uint32_t i = 0; i < VKSIFT_HIP_MAX_TAPS; i++)
      a.taps.k[i] = i < ntaps ? taps[i] : 0.f;
    const dim3 grid = stream_grid(W, H, batch, 10240u, &a.seg);
    switch (ntaps)
    {
#define VKSIFT_CASE(N)                                                                        \
  case N:                                                                                     \
    if (dst.fp16)                                                                             \
      hipLaunchKernelGGL((k_blur_lean<N, 2, true>), grid, dim3(64), 0, (hipStream_t)s, a);      \
    else                                                                                      \
      hipLaunchKernelGGL((k_blur_lean<N, 2, false>), grid, dim3(64), 0, (hipStream_t)s, a);     \
    break;
      VKSIFT_CASE(2) VKSIFT_CASE(3) VKSIFT_CASE(4) VKSIFT_CASE(5) VKSIFT_CASE(6) VKSIFT_CASE(7) VKSIFT_CASE(8) VKSIFT_CASE(9) VKSIFT_CASE(10)
      VKSIFT_CASE(11) VKSIFT_CASE(12)
#undef VKSIFT_CASE
    default:
      return -1;
    }
    return (int)hipGetLastError();
  }

  int vksift_hip_downsample(vksift_hip_Plane src, vksift_hip_Plane dst, uint32_t batch, vksift_hip_stream s)
  {
    dim3 grid((dst.w + 63) / 64, (dst.h + 3) / 4, batch);
    if (src.fp16)
      hipLaunchKernelGGL(k_downsample<true>, grid, dim3(256), 0, (hipStream_t)s, src.base, src.img_stride, (int)src.w, (int)src.h, (int)src.pitch, dst.base,
                         dst.img_stride, (int)dst.w, (int)dst.h, (int)dst.pitch);
    else
      hipLaunchKernelGGL(k_downsample<false>, grid, dim3(256), 0, (hipStream_t)s, src.base, src.img_stride, (int)src.w, (int)src.h, (int)src.pitch, dst.base,
                         dst.img_stride, (int)dst.w, (int)dst.h, (int)dst.pitch);
    return (int)hipGetLastError();
  }

  int vksift_hip_octave_chain(const vksift_hip_Plane *layers, uint32_t n_oct, uint32_t n_layers, uint32_t S, const float *taps, const uint32_t *ntaps,
                              uint32_t batch, vksift_hip_stream s)
  {
    if (n_oct < 1 || n_oct > (uint32_t)CH_MAX_OCT || n_layers < 2 || n_layers > (uint32_t)CH_MAX_LAYERS || S >= n_layers || batch < 1)
      return -1;
    ChainArgs a;
    a.n_oct = (int)n_oct, a.n_layers = (int)n_layers, a.S = (int)S;
    for (uint32_t o = 0; o < n_oct; o++)
    {
      const vksift_hip_Plane &p0 = layers[o * n_layers];
      if (p0.fp16 || (p0.w & 3u) || (p0.pitch & 3u) || p0.w < 8u || p0.h < 8u || (uint64_t)p0.w * p0.h * 2u > (uint64_t)CH_LDS_FLOATS)
        return -1;
      if (o > 0 && ((uint64_t)p0.w * p0.h > (uint64_t)CH_SEED * CH_THREADS || p0.w > layers[(o - 1) * n_layers].w || p0.h > layers[(o - 1) * n_layers].h))
        return -1;
      a.o[o].w = (int)p0.w, a.o[o].h = (int)p0.h, a.o[o].pitch = (int)p0.pitch, a.o[o].img_stride = p0.img_stride;
      for (uint32_t l = 0; l < n_layers; l++)
      {
        const vksift_hip_Plane &p = layers[o * n_layers + l];
        if (p.w != p0.w || p.h != p0.h || p.pitch != p0.pitch || p.img_stride != p0.img_stride || p.fp16 || ((uintptr_t)p.base & 15u))
          return -1;
        a.o[o].layer[l] = p.base;
      }
    }
    uint32_t min_side = 0xFFFFFFFFu;
    for (uint32_t o = 0; o < n_oct; o++)
    {
      min_side = a.o[o].w < (int)min_side ? (uint32_t)a.o[o].w : min_side;
      min_side = a.o[o].h < (int)min_side ? (uint32_t)a.o[o].h : min_side;
    }
    for (uint32_t l = 0; l < n_layers; l++)
    {
      if (ntaps[l] < 1 || ntaps[l] > VKSIFT_HIP_MAX_TAPS || (l > 0 && ntaps[l] > min_side))
        return -1; /* (a radius that reaches past one reflection: the per-scale launches handle it) */
      a.nt[l] = (int)ntaps[l];
      for (uint32_t i = 0; i < VKSIFT_HIP_MAX_TAPS; i++)
        a.k[l][i] = i < ntaps[l] ? taps[l * VKSIFT_HIP_MAX_TAPS + i] : 0.f;
    }
    static bool lds_attr_set = false;
    if (!lds_attr_set)
    {
      const hipError_t ae = hipFuncSetAttribute((const void *)k_octave_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(CH_LDS_FLOATS * sizeof(float)));
      if (ae != hipSuccess)
        return (int)ae;
      lds_attr_set = true;
    }
    const size_t lds_bytes = sizeof(float) * 2u * (size_t)a.o[0].w * a.o[0].h;
    hipLaunchKernelGGL(k_octave_chain, dim3(batch), dim3(CH_THREADS), lds_bytes, (hipStream_t)s, a);
    return (int)hipGetLastError();
  }

  int vksift_hip_dog_plane(const float *lo, const float *hi, uint32_t w, uint32_t h, uint32_t pitch, uint32_t fp16, float *out_dense, vksift_hip_stream s)
  {
    dim3 grid((w + 63) / 64, (h + 3) / 4, 1);
    if (fp16)
      hipLaunchKernelGGL(k_dog_plane<true>, grid, dim3(256), 0, (hipStream_t)s, lo, hi, (int)w, (int)h, (int)pitch, out_dense);
    else
      hipLaunchKernelGGL(k_dog_plane<false>, grid, dim3(256), 0, (hipStream_t)s, lo, hi, (int)w, (int)h, (int)pitch, out_dense);
    return (int)hipGetLastError();
  }
}
