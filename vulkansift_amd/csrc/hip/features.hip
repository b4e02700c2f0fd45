// features.hip — keypoint orientation assignment and 4x4x8 descriptor extraction (gfx950, wave64).
//
// Replaces ComputeOrientation.comp (dispatch sift_detector.c:1191-1241) and ComputeDescriptors.comp
// (sift_detector.c:1243-1259). The reference launches them with vkCmdDispatchIndirect, one small
// workgroup per keypoint; HIP has no indirect dispatch, so both kernels are persistent grid-stride
// loops that read the keypoint count from HBM. One 64-lane wave owns one keypoint; histograms are
// accumulated with integer LDS atomics (ds_add_u32) on the reference's fixed-point scale, so the
// result does not depend on the order in which lanes arrive (bit-reproducible run to run).
//
// Extra orientations are appended through a prefix scan (k_orientation_finalize) instead of the
// reference's global atomicAdd (ComputeOrientation.comp:170-183): order = (keypoint, histogram bin).
//
// All per-pixel arithmetic mirrors oracle/sift_oracle.c (orc_orientations / orc_descriptor) in its
// "det" math mode operation for operation; exp/atan2/sin/cos come from detmath.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../detmath.h"
#include "multi.h"
#include "vksift_hip.h"

namespace
{

#define PI_F 3.14159265358979323846f

struct GaussView
{
  const float *base; // Gaussian layer 0 of this image/octave
  int w, h, pitch;
  size_t plane;
};

// One texel of the keypoint's Gaussian layer through its buffer resource. voff / soff are the byte offsets of the fp32 layout;
// a binary16 pyramid (VKSIFT_PYRAMID_PRECISION_FLOAT16) halves them and widens the texel exactly.
// (F16 is a template parameter of the kernels: as a run-time flag it cost the VALU-bound descriptor kernel 60 %)
template <bool F16>
__device__ __forceinline__ float tap_ld(const __amdgpu_buffer_rsrc_t rs, unsigned voff, int soff)
{
  if (F16)
    return (float)__builtin_bit_cast(_Float16, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, voff >> 1, soff >> 1, 0));
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
// layer `layer` of image `img` (strides in texels)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t layer_rsrc(const float *gauss, size_t texel_off, int pitch, int h, bool f16)
{
  const void *p = f16 ? (const void *)((const _Float16 *)gauss + texel_off) : (const void *)(gauss + texel_off);
  return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, pitch * h * (f16 ? 2 : 4), 0x00020000);
}

// imageLoad with robust out-of-bounds behaviour (returns 0) — quirk Q2 relies on it.
__device__ __forceinline__ float ldg(const GaussView &g, const float *layer, int x, int y)
{
  if ((unsigned)x >= (unsigned)g.w || (unsigned)y >= (unsigned)g.h)
    return 0.f;
  return layer[(size_t)y * g.pitch + x];
}

struct FeatArgs
{
  const float *gauss;
  uint32_t fp16; // binary16 texels
  int w, h, pitch;
  uint64_t plane_stride, img_stride;
  uint8_t *feats;
  uint64_t feat_img_stride;
  uint32_t cap;
  uint32_t *found;
  uint32_t found_img_stride;
  float *ori_ang;
  uint32_t *ori_cnt;
  uint64_t ori_img_stride;
  uint32_t max_keep; // orientations kept per keypoint (1..18)
  uint32_t use_vlfeat;
  const float *desc_fp_tab;
  uint32_t desc_fp_tab_len;
  uint32_t sec; // section of the SIFT buffer this octave fills (k_descriptor's dense rows)
};

// ComputeDescriptors.comp:160-171 / ComputeOrientation.comp:100-104 wrap an angle with "if (t < 0) t += 2 pi; else if
// (t > 2 pi) t -= 2 pi". Both places only ever see t <= 2 pi — atan2 returns (-pi, pi], and the relative orientation is the
// difference of two angles of [0, 2 pi] (the keypoint orientations are (k/2 + 0.5) * 2 pi / 36 with k/2 <= 35.5) — so the
// second branch is dead and the wrap is one add, one compare, one select.
__device__ __forceinline__ float wrap_2pi(float t)
{
  const float up = t + 2.f * PI_F;
  return t < 0 ? up : t;
}

// sqrtf(x) and a / b as hipcc expands them, minus the parts that only matter outside the range the caller has checked:
// * sqrt_inrange: v_sqrt_f32 (<= 1 ulp), then one step down / up by the signs of two exact residuals — the compiler's own correctly rounded
//   expansion without the 2^32 input scaling for x < 2^-96 and without the class test for 0 / inf behind it (x = 0: both residual tests fail
//   on NaN / zero, the result is the 0 v_sqrt_f32 returned);
// * div_inrange: v_rcp_f32, one Newton step, quotient, two remainder corrections — the compiler's expansion without v_div_scale (operands
//   with exponents far apart or near the ends of the range), v_div_fmas' rescaling and v_div_fixup (zeros, infinities, NaN). With
//   2^-64 <= a <= b <= 2^2 (or a = 0) no intermediate leaves the normal range, so every step computes what the scaled sequence computes.
// Both return the bits of the IEEE operation (the descriptor parity tests run both paths against the oracle's sqrtf and '/').
// 14 instead of 26 and 8 instead of 11 instructions; k_descriptor is bound by VALU issue (profiles/r05_descriptor_floor.md).
__device__ __forceinline__ float sqrt_inrange(float x)
{
  const float s = __builtin_amdgcn_sqrtf(x);
  const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
  const float rd = fmaf(-sd, s, x), ru = fmaf(-su, s, x);
  float r = rd <= 0.f ? sd : s;
  r = ru > 0.f ? su : r;
  return r;
}
__device__ __forceinline__ float div_inrange(float a, float b)
{
  float r = __builtin_amdgcn_rcpf(b);
  const float e = fmaf(-b, r, 1.f);
  r = fmaf(e, r, r);
  float q = a * r;
  float m = fmaf(-b, q, a);
  q = fmaf(m, r, q);
  m = fmaf(-b, q, a);
  return fmaf(m, r, q);
}
// v is 0 or at least 2^(E - 127) (v >= 0): one subtract, one unsigned compare
template <uint32_t E>
__device__ __forceinline__ bool below_range(float v) { return __float_as_uint(v) - 1u < (E << 23) - 1u; }

// x / (2*pi): the 3-operation form of dm_div_2pi (detmath.h) without its tests. The caller checks the range (non-zero magnitudes below 2^-96
// take the general form); a zero comes out as +0 whatever its sign, and floor and remainder of a zero bin coordinate are the same for both.
__device__ __forceinline__ float div_2pi_inrange(float x)
{
  const float c2 = 2.f * PI_F, rc = 0x1.45f306p-3f;
  const float q = x * rc;
  return fmaf(fmaf(-q, c2, x), rc, q);
}

// Angle and length of a gradient for both per-keypoint kernels (ComputeOrientation.comp:102-106, ComputeDescriptors.comp:146-162): atan2 and
// sqrt. INRANGE: the short forms; *odd is set where their range conditions do not hold (the caller then repeats the step with the general
// forms: a wave-uniform branch that real images take in regions darker than ~2^-30, if at all; black synthetic backgrounds do, in the
// tails of the blurs: tests/test_gpu_descriptor_ranges.py)
template <bool INRANGE>
__device__ __forceinline__ void grad_polar(float gradX, float gradY, float *ori, float *len, bool *odd)
{
  const float g2 = (gradX * gradX) + (gradY * gradY);
  if (INRANGE)
  {
    const float ax = fabsf(gradX), ay = fabsf(gradY);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    *ori = dm_atan2f_ratio(div_inrange(mn, mx == 0.f ? 1.f : mx), ax, ay, gradX, gradY);
    *len = sqrt_inrange(g2);
    // mx >= 2^-48 (then g2 >= 2^-96 and no square underflowed) or 0; mn >= 2^-64 or 0. (Tested on mx, not on g2: a denormal mx beside
    // mn = 0 squares to an exact 0, which a test of g2 lets through to rcp(denormal) = inf.)
    *odd = below_range<127 - 48>(mx) || below_range<127 - 64>(mn);
  }
  else
  {
    *ori = dm_atan2f(gradY, gradX);
    *len = sqrtf(g2);
  }
}

// -------------------------------------------------------------------------------------------------
// Orientation histogram (ComputeOrientation.comp:52-186). grid = (blocks, batch); 4 keypoints/block.
// -------------------------------------------------------------------------------------------------
template <bool IMG_FAST, bool F16>
__global__ void __launch_bounds__(256) k_orientation(Multi<FeatArgs> m)
{
  const VBlock vb = vblock(m); // virtual grid (images, blocks) when IMG_FAST, (blocks, images) otherwise
  const FeatArgs &a = m.oct[vb.o];
  // One wave per keypoint, four independent waves per block: every wave owns its histogram and LDS executes the DS
  // operations of a wave in program order, so no workgroup barrier is needed (a wave with a 15x15 window does not wait
  // for a neighbour with a 29x29 one).
  __shared__ uint32_t s_hist[4][36];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = (int)(IMG_FAST ? vb.x : vb.y);
  const uint32_t found = a.found[(size_t)b * a.found_img_stride];
  const uint32_t n0 = found < a.cap ? found : a.cap;
  GaussView g{a.gauss + (size_t)b * a.img_stride, a.w, a.h, a.pitch, (size_t)a.plane_stride};
  uint8_t *feats = a.feats + (size_t)b * a.feat_img_stride;
  uint32_t *hist = s_hist[wave];

  const uint32_t bk = IMG_FAST ? vb.y : vb.x, nbk = IMG_FAST ? vb.gy : vb.gx;
  for (uint32_t k = bk * 4 + wave; k < n0; k += nbk * 4)
  {
    if (lane < 36)
      hist[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    float fp = 0.f;
    {
      const float *rec = (const float *)(feats + (size_t)k * 164);
      const float scale_x = rec[2], scale_y = rec[3];
      const uint32_t scale_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)((const uint32_t *)rec)[4]); // one keypoint per wave
      const int octave_idx = ((const int *)rec)[5];
      const float sigma = rec[6];
      // planes stay below 2 GiB (vksift_hip_extract_keypoints refuses sides of 16384 and more)
      const __amdgpu_buffer_rsrc_t rs = layer_rsrc(a.gauss, (size_t)b * a.img_stride + (size_t)scale_idx * g.plane, g.pitch, g.h, F16);

      // sigma / 2^octave_idx as a product with 2^-octave_idx: the same real number, rounded once either way
      float lambda = 1.5f * (sigma * dm_pow2i(-octave_idx));
      int r = (int)floorf(3 * lambda);
      float es = -1.f / (2.f * lambda * lambda);
      // the window's largest squared distance is 2 (r + 1)^2 (offsets of at most r + 0.5 + 0.5 per axis): can es * d2 pass the clamp of dm_expf?
      const bool may_clamp = (2.f * (float)(r + 1) * (float)(r + 1)) * es < -87.f;

      // fixed-point scale: M = (sum_i e^{es i^2})^2 * sqrt(2)  (separable form of :75-79, same order as the oracle's det mode)
      float gsum = 1.f;
      for (int cb = 0; cb <= r; cb += 64)
      {
        int i = cb + lane;
        float e = (i >= 1 && i <= r) ? dm_expf(es * (float)(i * i)) : 0.f;
        int lim = r - cb < 63 ? r - cb : 63;
        for (int j = 0; j <= lim; j++)
        {
          float ej = __shfl(e, j, 64);
          if (cb + j >= 1)
            gsum += 2.f * ej;
        }
      }
      float M = (gsum * gsum) * sqrtf(2.f);
      fp = (float)(1u << (uint32_t)(30 - dm_ceil_log2f(M)));

      float rsx = roundf(scale_x), rsy = roundf(scale_y);
      int box = 2 * r + 1;
      int npix = box * box;
      // lane l visits window pixels l, l + 64, ...: (row, column) advance by (64 / box, 64 % box) with at most one carry
      // a / box for 0 <= a <= 64 as (int)((a + 0.5) * rcp(box)): (a + 0.5) / box stays 0.5 / box away from every integer — for box < 32768
      // far more than the error of v_rcp_f32 and one product — 4 instructions instead of the ~25 of a 32-bit integer division
      const float rbox = __builtin_amdgcn_rcpf((float)box);
      const bool small_box = box < 32768;
      const int step_q = small_box ? (int)(64.5f * rbox) : 64 / box, step_r = 64 - step_q * box;
      int py = small_box ? (int)(((float)lane + 0.5f) * rbox) : lane / box, px = lane - py * box;
      // the contribution of one window texel given its gradient taps (ComputeOrientation.comp:102-121)
      auto accumulate = [&](float sdx2, float tr, float tl, float td, float tu) {
        const float gradX = 0.5f * (tr - tl);
        const float gradY = 0.5f * (td - tu);
        // dm_expf_nb_nonpos(sdx2 * es): its clamp as one v_max, its zero result only for keypoints whose window can reach the clamp at all
        // (exponents below -87.3: never with lambda >= 0.08, i.e. for any sigma a SIFT configuration produces)
        const float xe = sdx2 * es; /* d2 >= 0 > es */
        float e = dm_expf_core_nonpos(fmaxf(xe, -87.3f));
        if (may_clamp)
          e = xe < -87.3f ? 0.f : e;
        float ori, len, fbin;
        bool odd;
        grad_polar<true>(gradX, gradY, &ori, &len, &odd);
        float x36 = wrap_2pi(ori) * 36.f;
        fbin = div_2pi_inrange(x36); // == ori * 36 / (2 pi)
        odd |= below_range<127 - 96>(fabsf(x36));
        if (__builtin_expect(__ballot(odd) != 0ull, 0))
        {
          grad_polar<false>(gradX, gradY, &ori, &len, nullptr);
          x36 = wrap_2pi(ori) * 36.f;
          fbin = dm_div_2pi(x36);
        }
        const float mag = e * len;
        // ComputeOrientation.comp:107-112 wraps bin < 0 and bin >= 36; the angle is in [0, 2 pi] here (wrap_2pi), its quotient by 2 pi is
        // the correctly rounded one of a non-negative number: only 36 itself can occur
        int bin = (int)fbin;
        bin = bin >= 36 ? bin - 36 : bin;
        atomicAdd(&hist[bin], (uint32_t)(mag * fp));
      };
      const int cxi = (int)rsx, cyi = (int)rsy;
      const int pitch4 = g.pitch * 4;
      // (wave-uniform: one keypoint per wave) the whole window and its taps inside the image interior — all but the keypoints
      // within r + 1 texels of a border: no texel is skipped (quirk Q2's test needs a texel outside the interior) and no tap needs
      // the out-of-image handling, which is a third of the general loop's instructions
      if (cxi - r >= 1 && cxi + r <= g.w - 2 && cyi - r >= 1 && cyi + r <= g.h - 2)
      {
        for (int pix = lane; pix < npix; pix += 64)
        {
          const int dy = py - r, dx = px - r;
          px += step_r, py += step_q;
          if (px >= box)
            px -= box, py++;
          const float sdx = (rsx + (float)dx) - scale_x;
          const float sdy = (rsy + (float)dy) - scale_y;
          // byte offset of texel (x - 1, y - 1): the four taps are immediate / scalar offsets from it (scalar offsets are unsigned)
          const unsigned v0 = (__umul24((unsigned)(cyi + dy - 1), (unsigned)g.pitch) + (unsigned)(cxi + dx - 1)) * 4u;
          accumulate((sdx * sdx) + (sdy * sdy), tap_ld<F16>(rs, v0 + 8u, pitch4), tap_ld<F16>(rs, v0, pitch4), tap_ld<F16>(rs, v0 + 4u, 2 * pitch4), tap_ld<F16>(rs, v0 + 4u, 0));
        }
      }
      else
        for (int pix = lane; pix < npix; pix += 64)
        {
          const int dy = py - r, dx = px - r;
          px += step_r, py += step_q;
          if (px >= box)
            px -= box, py++;
          int gx = cxi + dx, gy = cyi + dy;
          float sdx = (rsx + (float)dx) - scale_x;
          float sdy = (rsy + (float)dy) - scale_y;
          float d2 = (sdx * sdx) + (sdy * sdy);
          if ((gx < 1 || gx >= (g.w - 1) || gy < 1 || gy >= (g.h - 1)) && (d2 > (float)(r * r)))
            continue; // quirk Q2 ('&&')
          // imageLoad semantics (0 outside the image, quirk Q2 relies on it) through the layer's buffer resource: a tap
          // outside the image gets an out-of-range offset, which the hardware answers with 0 — no branches, 32-bit offsets
          const bool xin = (unsigned)gx < (unsigned)g.w, yin = (unsigned)gy < (unsigned)g.h;
          // (each row offset from its own, in-range, row index: the 24-bit multiply must not see a negative row)
          const unsigned o0 = (__umul24((unsigned)gy, (unsigned)g.pitch) + (unsigned)gx) * 4u;
          const unsigned o_r = (yin && (unsigned)(gx + 1) < (unsigned)g.w) ? o0 + 4u : 0x80000000u;
          const unsigned o_l = (yin && (unsigned)(gx - 1) < (unsigned)g.w) ? o0 - 4u : 0x80000000u;
          const unsigned o_d = (xin && (unsigned)(gy + 1) < (unsigned)g.h) ? (__umul24((unsigned)(gy + 1), (unsigned)g.pitch) + (unsigned)gx) * 4u : 0x80000000u;
          const unsigned o_u = (xin && (unsigned)(gy - 1) < (unsigned)g.h) ? (__umul24((unsigned)(gy - 1), (unsigned)g.pitch) + (unsigned)gx) * 4u : 0x80000000u;
          accumulate(d2, tap_ld<F16>(rs, o_r, 0), tap_ld<F16>(rs, o_l, 0), tap_ld<F16>(rs, o_d, 0), tap_ld<F16>(rs, o_u, 0));
        }
    }
    __builtin_amdgcn_wave_barrier();
    {
      // 3 x 2 box-filter passes in registers (lane i holds bin i), :130-147
      const int li = lane < 36 ? lane : 0;
      uint32_t hv = hist[li];
      const int lm = (li + 35) % 36, lp = (li + 1) % 36;
#pragma unroll
      for (int it = 0; it < 6; it++)
      {
        uint32_t hm = __shfl(hv, lm, 64), hp = __shfl(hv, lp, 64);
        hv = (uint32_t)dm_div_3((float)(hm + hv + hp)); // == / 3.f for every integer-valued float up to 2^32 (detmath.h)
      }
      uint32_t hm = __shfl(hv, lm, 64), hp = __shfl(hv, lp, 64);
      uint32_t mx = lane < 36 ? hv : 0u;
#pragma unroll
      for (int dlt = 32; dlt >= 1; dlt >>= 1)
      {
        uint32_t t = __shfl_xor(mx, dlt, 64);
        mx = t > mx ? t : mx;
      }
      bool peak = lane < 36 && ((float)hv >= (0.8f * (float)mx)) && (hv > hm) && (hv > hp);
      unsigned long long pm = __ballot(peak);
      uint32_t npk = (uint32_t)__popcll(pm);
      if (peak)
      {
        uint32_t rank = (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
        if (rank < a.max_keep)
        {
          uint32_t num = hm - hp;                 // quirk Q3: uint32 wrap-around
          uint32_t den = hm - (2u * hv) + hp;
          float idx = (float)lane + 0.5f * ((float)num / (float)den);
          float ang = (idx + 0.5f) * (2.f * PI_F) / 36.f;
          a.ori_ang[((size_t)b * a.ori_img_stride + k) * VKSIFT_HIP_MAX_ORI + rank] = ang;
        }
      }
      if (lane == 0)
        a.ori_cnt[(size_t)b * a.ori_img_stride + k] = npk < a.max_keep ? npk : a.max_keep;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// -------------------------------------------------------------------------------------------------
// Write main orientations in place and append the extra-orientation copies in (keypoint, bin) order
// (ComputeOrientation.comp:170-183, made deterministic). One 1024-thread block per image.
// -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_orientation_finalize(Multi<FeatArgs> m)
{
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  const VBlock vb = vblock(m); // virtual grid (images)
  const FeatArgs &a = m.oct[vb.o];
  const int b = (int)vb.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t found = a.found[(size_t)b * a.found_img_stride];
  const uint32_t n0 = found < a.cap ? found : a.cap;
  uint8_t *feats = a.feats + (size_t)b * a.feat_img_stride;
  const uint32_t *cnt = a.ori_cnt + (size_t)b * a.ori_img_stride;
  const float *ang = a.ori_ang + (size_t)b * a.ori_img_stride * VKSIFT_HIP_MAX_ORI;
  if (threadIdx.x == 0)
    carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n0; base += 1024)
  {
    uint32_t k = base + threadIdx.x;
    uint32_t c = k < n0 ? cnt[k] : 0u;
    uint32_t v = c > 1 ? c - 1 : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1)
    {
      uint32_t t = __shfl_up(incl, dlt, 64);
      if (lane >= dlt)
        incl += t;
    }
    if (lane == 63)
      wave_tot[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0;
    for (int wv = 0; wv < wave; wv++)
      wave_base += wave_tot[wv];
    uint32_t carry = carry_s;
    uint32_t excl = carry + wave_base + incl - v;
    if (k < n0 && c >= 1)
    {
      uint32_t *rec = (uint32_t *)(feats + (size_t)k * 164);
      rec[7] = __float_as_uint(ang[(size_t)k * VKSIFT_HIP_MAX_ORI]);
      for (uint32_t j = 1; j < c; j++)
      {
        uint32_t idx = found + excl + (j - 1);
        if (idx < a.cap)
        {
          uint32_t *dst = (uint32_t *)(feats + (size_t)idx * 164);
#pragma unroll
          for (int q = 0; q < 9; q++)
            dst[q] = rec[q];
          dst[7] = __float_as_uint(ang[(size_t)k * VKSIFT_HIP_MAX_ORI + j]);
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 1023)
      carry_s = carry + wave_base + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0)
    a.found[(size_t)b * a.found_img_stride] = found + carry_s;
}

// -------------------------------------------------------------------------------------------------
// Descriptor (ComputeDescriptors.comp:84-274). grid = (blocks, batch); 4 keypoints per block.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ int smod8(int v) { return v & 7; } // floored modulo for a power of two (quirk Q5, OpSMod)


// Per-pixel descriptor contribution (ComputeDescriptors.comp:139-197) for window offset (cdx, cdy).
struct DescCtx
{
  __amdgpu_buffer_rsrc_t rs; // the keypoint's Gaussian layer
  int pitch, pitch4;
  float scale_x, scale_y, rsx, rsy, kcos, ksin, kori, fp, bin_scale;
};

// The contribution of one window pixel in two straight-line halves (no branch in either, so that the scheduler can
// interleave the halves of two samples): desc_sample() = taps, gradient, angle, weight; desc_scatter() = bins + atomics.
struct DescSample
{
  float ox, oy, mag;
  float xb; // 8 * relative orientation, before the division by 2*pi
};

// the four taps of a sample: up (x, y-1), left (x-1, y), right (x+1, y), down (x, y+1)
struct DescTaps
{
  float up, lf, rt, dn;
};

// byte offset of texel (ix - 1, iy - 1) of the sample at window offset (cdx, cdy): the taps are immediate / scalar offsets from it
// (the window is clipped to the image interior, so every tap of a live sample is in range; sides < 16384: 24-bit factors)
__device__ __forceinline__ unsigned desc_tap_base(const DescCtx &c, int cdx, int cdy)
{
  const int ix = (int)c.rsx + cdx, iy = (int)c.rsy + cdy;
  return (__umul24((unsigned)(iy - 1), (unsigned)c.pitch) + (unsigned)(ix - 1)) * 4u;
}

template <bool F16>
__device__ __forceinline__ DescTaps desc_taps(const DescCtx &c, unsigned v0)
{
  DescTaps t;
  t.up = tap_ld<F16>(c.rs, v0 + 4u, 0);
  t.lf = tap_ld<F16>(c.rs, v0, c.pitch4);
  t.rt = tap_ld<F16>(c.rs, v0 + 8u, c.pitch4);
  t.dn = tap_ld<F16>(c.rs, v0 + 4u, 2 * c.pitch4);
  return t;
}

// Two samples of one lane, the second the right-hand neighbour of the first (the common case: a lane walks along a row): their
// eight taps are the texels (x, x+1) of the rows above and below and (x-1 .. x+2) of the row itself — three loads (8, 16, 8
// bytes; dword-aligned, which is all a buffer load needs) instead of eight. The descriptor kernel keeps the address path of the
// texture unit full (SQ_VMEM_TA_ADDR_FIFO_FULL 29 % of its busy cycles with one load per tap).
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void desc_taps_pair(const DescCtx &c, unsigned v0, DescTaps &a, DescTaps &b)
{
  const u32x2_t up = __builtin_amdgcn_raw_buffer_load_b64(c.rs, v0 + 4u, 0, 0);
  const u32x4_t mid = __builtin_amdgcn_raw_buffer_load_b128(c.rs, v0, c.pitch4, 0);
  const u32x2_t dn = __builtin_amdgcn_raw_buffer_load_b64(c.rs, v0 + 4u, 2 * c.pitch4, 0);
  a.up = __uint_as_float(up.x), b.up = __uint_as_float(up.y);
  a.lf = __uint_as_float(mid.x), b.lf = __uint_as_float(mid.y);
  a.rt = __uint_as_float(mid.z), b.rt = __uint_as_float(mid.w);
  a.dn = __uint_as_float(dn.x), b.dn = __uint_as_float(dn.y);
}

// INRANGE: see grad_polar
template <bool INRANGE>
__device__ __forceinline__ DescSample desc_sample(const DescCtx &c, int cdx, int cdy, const DescTaps &t, bool *odd)
{
  const float es = -1.f / (2.f * 2 * 2);
  float sdx = (c.rsx + (float)cdx) - c.scale_x;
  float sdy = (c.rsy + (float)cdy) - c.scale_y;
  DescSample r;
  r.ox = c.kcos * sdx + c.ksin * sdy;
  r.oy = c.kcos * sdy - c.ksin * sdx;
  float gradX = 0.5f * (t.rt - t.lf);
  float gradY = 0.5f * (t.dn - t.up);
  float ori, root;
  grad_polar<INRANGE>(gradX, gradY, &ori, &root, odd);
  ori = wrap_2pi(ori);
  ori = wrap_2pi(ori - c.kori);
  // |ox|, |oy| < 4 for every enumerated sample: the exponent is in [-4, 0], no range handling needed
  r.mag = dm_expf_core_nonpos(es * ((r.ox * r.ox) + (r.oy * r.oy))) * root;
  r.xb = ori * c.bin_scale; // +8 (VLFeat order) or -8: (-ori) * 8 == ori * (-8) exactly
  return r;
}

// Histogram layout for 64-bit atomics: a cell is eight 8-byte slots, slot b holding the sums for the bin pair (b, b + 1 mod 8). The two
// orientation bins a sample touches are hb and hb + 1 (mod 8), so each of the four cells takes ONE ds_add_u64 into slot hb mod 8 instead of
// two ds_add_u32 (the sums stay below 2^32 by construction of the fixed-point scale, so the low half never carries into the high half),
// and the slot's byte offset is (hb & 7) * 8 — no parity split. Cell = 64 bytes. The epilogue adds the two halves that belong to a bin.
// The 4x4 grid sits inside a 6x8 array of cells (grid cell (cx, cy) = array cell (cx + 1, cy + 1)), so the 2x2 cells of a
// sample share ONE address register — the other three are immediate offsets of the ds_add_u64, and the lower corner of a sample
// one cell outside the grid still gives a non-negative address. Cells outside the grid (ComputeDescriptors.comp:189 drops
// them: 36 % of all cell updates, the enumerated footprint is 5x5 cells) are not executed at all: the add sits under the
// exec mask of its in-grid test. (Round 3's form redirected them to a per-lane dummy slot — four selects and three more
// address computations per sample, and an LDS atomic for every dropped cell. Letting them land in the border cells of
// the array instead, without any test, was measured 1.3 % SLOWER despite 6 % fewer instructions: the dropped updates then collide
// on 20 border cells like the kept ones do on the grid, and the kernel is as sensitive to same-address LDS atomics as to
// VALU issue.) Only the 16 grid cells are ever touched.
constexpr int DESC_PAD = 1, DESC_ROW_CELLS = 8; // rows of 8 cells: a grid column keeps the LDS banks it had in the dense 4x4 layout
constexpr int DESC_HIST_WORDS = (4 + 2 * DESC_PAD) * DESC_ROW_CELLS * 16; // 6 rows x 8 cells x (8 + 8) words = 3 KiB
constexpr int DESC_WORK_WORDS = DESC_HIST_WORDS;
__device__ __forceinline__ int desc_cell_word(int cell) // first word of grid cell cy * 4 + cx
{
  return (((cell >> 2) + DESC_PAD) * DESC_ROW_CELLS + (cell & 3) + DESC_PAD) * 16;
}
__device__ __forceinline__ uint32_t desc_hist_read(const uint32_t *s_work, int t) // descriptor element t = cell * 8 + bin
{
  const int cw = desc_cell_word(t >> 3), bin = t & 7;
  // slot b of a cell holds (bin b, bin b + 1 mod 8): bin = the low half of its own slot + the high half of the slot below
  return s_work[cw + 2 * bin] + s_work[cw + 2 * ((bin + 7) & 7) + 1];
}

__device__ __forceinline__ void desc_scatter(const DescCtx &c, const DescSample &r, float fbin, bool live, uint32_t *s_work)
{
  float fhx = r.ox + 2.f, fhy = r.oy + 2.f;
  // (float)(int)floorf(v) == floorf(v) for these small values: the remainders take the floor as it is, one conversion less each
  const float flx = floorf(fhx - 0.5f), fly = floorf(fhy - 0.5f), flb = floorf(fbin);
  int hx = (int)flx, hy = (int)fly, hb = (int)flb;
  float rhx = fhx - (flx + 0.5f), rhy = fhy - (fly + 0.5f), rb = fbin - flb;
  // ((w * wb) * mag) * fp == (w * wb) * (mag * fp) bit for bit: fp is a power of two (2^0 .. 2^31), so mag * fp is exact and
  // scaling by it commutes with the rounding of the product — except where (w * wb) * mag is subnormal, and there both
  // forms are far below 1 and convert to 0. One multiply per sample instead of one per bin value (8).
  // (Packed v_pk_mul_f32 for the weight products was measured 9 % slower despite 6 % fewer instructions.)
  const float magfp = r.mag * c.fp;
  const unsigned b0 = (unsigned)smod8(hb);
  const unsigned pair = b0 * 8u; // byte offset of the (hb, hb + 1) slot inside a cell
  // in-grid tests of the two columns and the two rows (a dead sample — a lane whose run has ended — fails all of them);
  // the address of a lane that fails is never used
  const bool okx[2] = {live && (unsigned)hx < 4u, live && (unsigned)(hx + 1) < 4u};
  const bool oky[2] = {(unsigned)hy < 4u, (unsigned)(hy + 1) < 4u};
  char *base = (char *)s_work + ((unsigned)((hy + DESC_PAD) * (DESC_ROW_CELLS * 64) + (hx + DESC_PAD) * 64) + pair);
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
    {
      const float w = fabsf(1.f - (float)i - rhx) * fabsf(1.f - (float)j - rhy);
      const uint32_t v0 = (uint32_t)((w * fabsf(1.f - 0.f - rb)) * magfp);
      const uint32_t v1 = (uint32_t)((w * fabsf(1.f - 1.f - rb)) * magfp);
      if (okx[i] && oky[j])
        atomicAdd((unsigned long long *)(base + (j * DESC_ROW_CELLS * 64 + i * 64)), ((unsigned long long)v1 << 32) | v0);
    }
}

// One 256-thread workgroup per keypoint (the window of a coarse-scale keypoint has up to ~7.5k pixels; a single wave
// would make it the critical path of the whole launch). About half of the window falls outside the rotated 4x4 grid:
// the reference evaluates atan/exp for those pixels and then drops the contribution (ComputeDescriptors.comp:189);
// here they are never visited (analytic row spans, see below), which changes no bit. The expensive part (gradient,
// atan2, exp, 8 fixed-point atomics) runs on full 64-lane batches.
constexpr int DESC_MAX_ROWS = 256; // window rows handled per pass (R <= 127: every stock configuration); taller windows take several passes

template <int NWV, bool IMG_FAST, bool F16>
__global__ void __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu((NWV == 2 && !F16) ? 8 : 6, 8)))
k_descriptor(Multi<FeatArgs> m, vksift_hip_DenseRows dr)
{
  const VBlock vb = vblock(m); // virtual grid (images, blocks) when IMG_FAST, (blocks, images) otherwise
  const FeatArgs &a = m.oct[vb.o];
  constexpr int NT_ = 64 * NWV;
  __shared__ __attribute__((aligned(8))) uint32_t s_work[DESC_WORK_WORDS]; // see desc_scatter
  __shared__ int s_row_lo[DESC_MAX_ROWS];
  __shared__ uint32_t s_row_cnt[DESC_MAX_ROWS];
  __shared__ uint32_t s_row_pre[NWV][DESC_MAX_ROWS + 1];
  const int tid = threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = (int)(IMG_FAST ? vb.x : vb.y);
  const uint32_t found = a.found[(size_t)b * a.found_img_stride];
  const uint32_t n1 = found < a.cap ? found : a.cap;
  GaussView g{a.gauss + (size_t)b * a.img_stride, a.w, a.h, a.pitch, (size_t)a.plane_stride};
  uint8_t *feats = a.feats + (size_t)b * a.feat_img_stride;
  // Dense rows for the matcher (vksift_hip_DenseRows): this section's features follow the stored features of the sections in front of
  // it. All counters are final here (the orientation launches of every octave precede this one); uniform scalar loads.
  // (the row offset waits in LDS for the epilogue: nothing of this stays in registers across the sample loop)
  __shared__ uint32_t s_row0;
  // (posting belongs to single-image detections, which take the four-wave form: the two-wave instantiation of batches — the kernel that
  // sits on its instruction floor with 64 VGPRs — does not carry that code)
  constexpr bool CAN_POST = NWV == 4;
  if (dr.desc || (CAN_POST && dr.post))
  {
    const uint32_t *fsec = a.found + (size_t)b * a.found_img_stride - a.sec; // counter of section 0 of this image's buffer
    uint32_t row0 = 0, total = 0;
    // (constant indices into the by-value argument: a run-time index would send the struct through scratch memory — 8 bytes of scratch
    // cost this kernel 4 %, tests/test_kernel_resources.py)
#pragma unroll
    for (uint32_t j = 0; j < 16u; j++)
      if (j < dr.nsec)
      {
        const uint32_t f = fsec[j], n = f < dr.sec_cap[j] ? f : dr.sec_cap[j];
        row0 += j < a.sec ? n : 0u;
        total += n;
      }
    if (tid == 0)
      s_row0 = row0; // read behind the barriers of the keypoint loop
    // the buffer's row count and the zero rows of quirk Q6: first workgroup of section 0's share of this image
    if (a.sec == 0 && (IMG_FAST ? vb.y : vb.x) == 0)
    {
      if (CAN_POST && dr.post && (uint32_t)tid < dr.found_post_n)
        dr.found_post[(size_t)b * dr.found_post_n + tid] = fsec[tid]; // the counters travel with the posted records
      if (dr.desc && tid == 0)
        dr.n[(size_t)b * dr.n_img_stride] = total;
      if (dr.desc && total < 2u && tid < 64)
      {
        uint32_t *z = (uint32_t *)(dr.desc + (size_t)b * dr.desc_img_stride);
        if (tid >= (int)total * 32)
          z[tid] = 0u;
        if (tid < 2 && tid >= (int)total)
          dr.norm[(size_t)b * dr.norm_img_stride + tid] = 128u * 128u * 128u;
      }
    }
  }

  for (uint32_t k = IMG_FAST ? vb.y : vb.x; k < n1; k += IMG_FAST ? vb.gy : vb.gx)
  {
    __syncthreads(); // the previous keypoint's epilogue has read the histogram
    for (int i = tid; i < 256; i += NT_)
      s_work[desc_cell_word(i >> 4) + (i & 15)] = 0; // the 16 grid cells; made visible by the barrier behind the row-span pass below

    const float *rec = (const float *)(feats + (size_t)k * 164);
    const uint32_t scale_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)((const uint32_t *)rec)[4]); // uniform: keeps the resource in SGPRs
    const int octave_idx = ((const int *)rec)[5];
    const float sigma = rec[6];
    DescCtx c;
    c.scale_x = rec[2], c.scale_y = rec[3], c.kori = rec[7];
    // planes stay below 2 GiB (vksift_hip_extract_keypoints refuses sides of 16384 and more): 32-bit byte offsets
    c.rs = layer_rsrc(a.gauss, (size_t)b * a.img_stride + (size_t)scale_idx * g.plane, g.pitch, g.h, F16);
    c.pitch = g.pitch, c.pitch4 = g.pitch * 4;
    c.bin_scale = a.use_vlfeat ? 8.f : -8.f;
    float scale_factor = dm_pow2i(octave_idx);
    float lambda = 3.0f * (sigma / scale_factor);
    float radius = sqrtf(2.f) * lambda * 5.f * 0.5f;
    int R = (int)floorf(radius + 0.5f);
    float sn, cs;
    dm_sincosf(c.kori, &sn, &cs);
    c.kcos = cs / lambda, c.ksin = sn / lambda;
    uint32_t ti = (uint32_t)(R / 2);
    c.fp = a.desc_fp_tab[ti < a.desc_fp_tab_len ? ti : a.desc_fp_tab_len - 1];
    c.rsx = roundf(c.scale_x), c.rsy = roundf(c.scale_y);

    // The window [-R, R]^2 is clipped to the image interior up front (ComputeDescriptors.comp:151 skips the border texels).
    const int cxi = (int)c.rsx, cyi = (int)c.rsy;
    const int dx0 = max(-R, 1 - cxi), dx1 = min(R, g.w - 2 - cxi);
    const int dy0 = max(-R, 1 - cyi), dy1 = min(R, g.h - 2 - cyi);
    const int bh = dy1 - dy0 + 1;
    // Only pixels whose rotated position (ox, oy) lies within the 5x5-cell footprint |ox|, |oy| < 2.5 of the 4x4 grid can
    // touch a bin (about half of the window). Instead of testing every pixel, each window row gets its span of such
    // pixels analytically: ox and oy are linear in dx, so {dx : |kcos*fx + ksin*fy| < T and |kcos*fy - ksin*fx| < T} is an
    // interval. The spans are approximate (fused arithmetic) and conservative (margin): desc_accumulate re-derives the
    // cells exactly and drops the out-of-grid ones, so a false positive costs a sample slot, never a bit.
    // Sample enumeration: the concatenated spans form the index space [0, N); wave v owns [v, v+1) * N / NWV and inside it
    // lane l walks its own contiguous run — at any step the 64 lanes sit a run length apart and spread over all 16 spatial
    // cells, so the 8 fixed-point LDS atomics of a step hit mostly distinct addresses (row-adjacent samples pile onto
    // 1-2 cells and serialise: measured 8 % slower). Integer adds commute: the result does not depend on the order.
    const float offx = c.rsx - c.scale_x, offy = c.rsy - c.scale_y;
    const float T = 2.5f + 0.01f;
    // (reciprocals by v_rcp_f32, 1 ulp: the spans are conservative by 0.01 in T and 0.01 px at both ends, six orders of magnitude more than
    // that error on coordinates below 2^14 — the four IEEE divisions per row cost every wave 44 instructions per keypoint, round 6)
    const float ia0 = __builtin_amdgcn_rcpf(c.kcos), ia1 = __builtin_amdgcn_rcpf(-c.ksin);
    for (int rb = 0; rb < bh; rb += DESC_MAX_ROWS)
    {
      const int nrows = min(DESC_MAX_ROWS, bh - rb);
      __syncthreads();
      for (int row = tid; row < nrows; row += NT_)
      {
        const float fy = (float)(dy0 + rb + row) + offy;
        float lo = (float)dx0 - 0.5f, hi = (float)dx1 + 0.5f; // in dx
        // |a * fx + bb| < T for (a, bb) = (kcos, ksin*fy) and (-ksin, kcos*fy); fx = dx + offx
#pragma unroll
        for (int e = 0; e < 2; e++)
        {
          const float aa = e == 0 ? c.kcos : -c.ksin, ia = e == 0 ? ia0 : ia1;
          const float bb = e == 0 ? c.ksin * fy : c.kcos * fy;
          if (fabsf(aa) > 1e-12f)
          {
            const float ctr = -bb * ia - offx, half = T * fabsf(ia);
            lo = fmaxf(lo, ctr - half);
            hi = fminf(hi, ctr + half);
          }
          else if (!(fabsf(bb) < T))
            hi = lo - 2.f; // empty
        }
        const int ilo = (int)ceilf(lo - 0.01f), ihi = (int)floorf(hi + 0.01f);
        const int xl = max(ilo, dx0), xh = min(ihi, dx1);
        s_row_lo[row] = xl;
        s_row_cnt[row] = xh >= xl ? (uint32_t)(xh - xl + 1) : 0u;
      }
      __syncthreads();
      {
        // exclusive scan of the row counts (nrows <= DESC_MAX_ROWS), total in s_row_pre[nrows]. Every wave computes it for
        // itself from the counts (s_row_cnt is read-only here) into its own copy of the prefix array: no barrier afterwards.
        uint32_t *pre = s_row_pre[wave];
        uint32_t carry = 0;
        for (int base = 0; base < nrows; base += 64)
        {
          const int i = base + lane;
          const uint32_t v = i < nrows ? s_row_cnt[i] : 0u;
          uint32_t incl = v;
#pragma unroll
          for (int dlt = 1; dlt < 64; dlt <<= 1)
          {
            const uint32_t t = __shfl_up(incl, dlt, 64);
            if (lane >= dlt)
              incl += t;
          }
          if (i < nrows)
            pre[i] = carry + incl - v;
          carry += __shfl(incl, 63, 64);
        }
        if (lane == 0)
          pre[nrows] = carry;
      }
      __builtin_amdgcn_wave_barrier();
      const uint32_t *s_pre = s_row_pre[wave];
      const uint32_t N = s_pre[nrows];
      // The N samples need T = ceil(N / 64) wave-steps; they are dealt out to the waves in whole steps (the first T % NWV
      // waves take one more), so only the last step of the last wave has idle lanes. Splitting N into NWV equal parts
      // first and rounding each up to whole steps cost up to NWV - 1 extra steps (11 % of a typical 25-step keypoint).
      const uint32_t T = (N + 63u) / 64u, tq = T / NWV, tr = T % NWV;
      uint32_t run = tq + ((uint32_t)wave < tr ? 1u : 0u); // samples per lane = steps of this wave
      const uint32_t step0 = (uint32_t)wave * tq + min((uint32_t)wave, tr);
      uint32_t sidx = min(N, step0 * 64u + (uint32_t)lane * run);
      uint32_t send = min(N, sidx + run);
      // locate the row of the first sample of this lane's run: largest row with pre[row] <= sidx
      int row = 0;
      if (sidx < send)
      {
        int lo_r = 0, hi_r = nrows - 1;
        while (lo_r < hi_r)
        {
          const int mid = (lo_r + hi_r + 1) >> 1;
          if (s_pre[mid] <= sidx)
            lo_r = mid;
          else
            hi_r = mid - 1;
        }
        row = lo_r;
        while (s_pre[row + 1] <= sidx) // skip empty rows that share the same prefix value
          row++;
      }
      int cdx = s_row_lo[row] + (int)(sidx - s_pre[row]);
      uint32_t row_end = s_pre[row + 1];
      // Two samples per iteration: their (straight-line) contribution code is independent, so the scheduler interleaves
      // the two dependency chains instead of padding every compare -> select and transcendental with wait states. A lane
      // whose run has ended carries a dead sample: the taps go through the buffer resource (any offset is safe) and
      // no contribution is executed.
      auto next_sample = [&](int &sx, int &sy, bool &live) {
        live = sidx < send;
        if (live && sidx >= row_end)
        {
          do
            row++;
          while (s_pre[row + 1] <= sidx);
          row_end = s_pre[row + 1];
          cdx = s_row_lo[row];
        }
        sx = cdx, sy = dy0 + rb + row;
        cdx++, sidx++;
      };
      uint32_t it = 0;
      for (; it + 1 < run; it += 2)
      {
        int sx[2], sy[2];
        bool live[2];
        next_sample(sx[0], sy[0], live[0]);
        next_sample(sx[1], sy[1], live[1]);
        DescTaps t0, t1;
        const unsigned v0 = desc_tap_base(c, sx[0], sy[0]);
        if (F16)
        {
          t0 = desc_taps<F16>(c, v0);
          t1 = desc_taps<F16>(c, desc_tap_base(c, sx[1], sy[1]));
        }
        else
        {
          // (a dead second sample takes whatever the pair load returns: it contributes nothing; the loads go
          // through the buffer resource, so the texels past a row end are harmless too)
          desc_taps_pair(c, v0, t0, t1);
          if (live[1] && !(sy[1] == sy[0] && sx[1] == sx[0] + 1))
            t1 = desc_taps<F16>(c, desc_tap_base(c, sx[1], sy[1])); // the run crossed into the next row span
        }
        bool o0, o1;
        DescSample s0 = desc_sample<true>(c, sx[0], sy[0], t0, &o0), s1 = desc_sample<true>(c, sx[1], sy[1], t1, &o1);
        float f0 = div_2pi_inrange(s0.xb), f1 = div_2pi_inrange(s1.xb);
        o0 |= below_range<127 - 96>(fabsf(s0.xb)), o1 |= below_range<127 - 96>(fabsf(s1.xb));
        if (__builtin_expect(__ballot(o0 || o1) != 0ull, 0))
        {
          // some lane holds a value outside the ranges of the short forms (see desc_sample): the general forms for this step
          s0 = desc_sample<false>(c, sx[0], sy[0], t0, nullptr), s1 = desc_sample<false>(c, sx[1], sy[1], t1, nullptr);
          f0 = dm_div_2pi(s0.xb), f1 = dm_div_2pi(s1.xb);
        }
        desc_scatter(c, s0, f0, live[0], s_work);
        desc_scatter(c, s1, f1, live[1], s_work);
      }
      if (it < run) // odd number of steps
      {
        int sx, sy;
        bool live;
        next_sample(sx, sy, live);
        const DescSample s0 = desc_sample<false>(c, sx, sy, desc_taps<F16>(c, desc_tap_base(c, sx, sy)), nullptr);
        desc_scatter(c, s0, dm_div_2pi(s0.xb), live, s_work);
      }
    }
    __syncthreads();
    if (wave == 0)
    {
      // (an opaque copy of the lane index for the whole epilogue: addresses and masks derived from it are recomputed per keypoint instead
      // of being hoisted out of the keypoint loop into registers the sample loop cannot spare — hoisted, one of them was spilled, and any
      // scratch costs this kernel per wave launched: tests/test_kernel_resources.py)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      // normalise -> clamp at 0.2*norm -> renormalise -> x512 -> u8 (:200-265)
      uint32_t w0 = desc_hist_read(s_work, ln), w1 = desc_hist_read(s_work, ln + 64);
      uint32_t acc = w0 * w0 + w1 * w1;
#pragma unroll
      for (int dlt = 32; dlt >= 1; dlt >>= 1)
        acc += __shfl_xor(acc, dlt, 64);
      float norm = sqrtf((float)acc);
      uint32_t lim = (uint32_t)(norm * 0.2f);
      w0 = w0 < lim ? w0 : lim;
      w1 = w1 < lim ? w1 : lim;
      acc = w0 * w0 + w1 * w1;
#pragma unroll
      for (int dlt = 32; dlt >= 1; dlt >>= 1)
        acc += __shfl_xor(acc, dlt, 64);
      norm = sqrtf((float)acc);
      float scale = 512.f / norm;
      float v0 = (float)w0 * scale, v1 = (float)w1 * scale;
      uint32_t b0 = (v0 != v0) ? 0u : (v0 < 0.f ? 0u : (v0 > 255.f ? 255u : (uint32_t)v0));
      uint32_t b1 = (v1 != v1) ? 0u : (v1 < 0.f ? 0u : (v1 > 255.f ? 255u : (uint32_t)v1));
      // pack 4 consecutive bytes per dword: lanes 4q..4q+3 hold bytes of dword q (first half) / q+16 (second half)
      uint32_t sh = (uint32_t)(ln & 3) * 8u;
      uint32_t p0 = b0 << sh, p1 = b1 << sh;
      p0 |= __shfl_xor(p0, 1, 64);
      p0 |= __shfl_xor(p0, 2, 64);
      p1 |= __shfl_xor(p1, 1, 64);
      p1 |= __shfl_xor(p1, 2, 64);
      if ((ln & 3) == 0)
      {
        uint32_t *desc = (uint32_t *)(feats + (size_t)k * 164 + 36);
        desc[ln >> 2] = p0;
        desc[16 + (ln >> 2)] = p1;
      }
      if (CAN_POST && dr.post)
      {
        // the posted record: 9 header words from the section record (written by the extraction and orientation launches), the
        // descriptor's 32 words from the registers of the lanes that hold them
        const uint32_t j = (uint32_t)ln & 15u;
        const uint32_t d0 = __shfl(p0, (int)(4u * j), 64), d1 = __shfl(p1, (int)(4u * j), 64);
        uint32_t *out = (uint32_t *)(dr.post + (size_t)b * dr.post_img_stride) + (size_t)(s_row0 + k) * 41u;
        if (ln < 32)
          out[9 + ln] = ln < 16 ? d0 : d1;
        else if (ln < 41)
          out[ln - 32] = ((const uint32_t *)rec)[ln - 32];
      }
      if (dr.desc)
      {
        const uint32_t row0 = s_row0;
        uint32_t *dense = (uint32_t *)(dr.desc + (size_t)b * dr.desc_img_stride) + (size_t)row0 * 32;
        uint32_t *dense_norm = dr.norm + (size_t)b * dr.norm_img_stride + row0;
        // the same 128 bytes as a dense row + |d - 128|^2 = sum d^2 - 256 sum d + 128 * 128^2 (k_shifted_norms): every lane of a
        // group of four holds the group's two dwords, the sums run over the 16 groups
        uint32_t s2 = __builtin_amdgcn_udot4(p0, p0, 0u, false), s1 = __builtin_amdgcn_udot4(p0, 0x01010101u, 0u, false);
        s2 = __builtin_amdgcn_udot4(p1, p1, s2, false), s1 = __builtin_amdgcn_udot4(p1, 0x01010101u, s1, false);
#pragma unroll
        for (int dlt = 32; dlt >= 4; dlt >>= 1)
        {
          s2 += __shfl_xor(s2, dlt, 64);
          s1 += __shfl_xor(s1, dlt, 64);
        }
        if ((ln & 3) == 0)
        {
          uint32_t *row = dense + (size_t)k * 32;
          row[ln >> 2] = p0;
          row[16 + (ln >> 2)] = p1;
        }
        if (ln == 0)
          dense_norm[k] = s2 - 256u * s1 + 128u * 128u * 128u;
      }
    }
  }
}

FeatArgs make_args(const vksift_hip_OctaveJob *job)
{
  FeatArgs a;
  a.gauss = job->gauss;
  a.fp16 = job->fp16;
  a.w = (int)job->w, a.h = (int)job->h, a.pitch = (int)job->pitch;
  a.plane_stride = job->plane_stride, a.img_stride = job->img_stride;
  a.feats = job->feats, a.feat_img_stride = job->feat_img_stride, a.cap = job->cap;
  a.found = job->found, a.found_img_stride = job->found_img_stride;
  a.ori_ang = job->ori_ang, a.ori_cnt = job->ori_cnt, a.ori_img_stride = job->ori_img_stride;
  uint32_t mk = job->max_ori == 0 ? VKSIFT_HIP_MAX_ORI : job->max_ori;
  a.max_keep = mk > VKSIFT_HIP_MAX_ORI ? VKSIFT_HIP_MAX_ORI : mk;
  a.use_vlfeat = job->use_vlfeat;
  a.desc_fp_tab = job->desc_fp_tab, a.desc_fp_tab_len = job->desc_fp_tab_len;
  a.sec = job->sec_index;
  return a;
}

// Work -> workgroup mapping of the per-keypoint kernels. The number of busy workgroups of an image (its keypoint count)
// is only known on the device and usually a small part of the grid. With the keypoint index fastest the busy ones form
// a run at the start of every image's row, which the dispatcher's round-robin over XCDs and shader engines can map onto a
// fraction of the machine (measured on the refinement kernels: 3x); with the image index fastest the busy workgroups
// are contiguous in dispatch order, and for batches that are multiples of 8 every image stays on one XCD (its pyramid
// planes in one L2). grid.y is limited to 65535.
bool img_fast(uint32_t batch, bool descriptor)
{
  static int mode = -1;
  if (mode < 0)
  {
    /* bit 0: orientation kernel, bit 1: descriptor kernel. Measured (128 x 640x480): orientation -5 %, descriptor +3 %
     * (its grid is mostly busy and the keypoint-fastest order spreads the long coarse-scale windows better): default 1 */
    mode = 1;
  }
  return batch > 1 && ((mode >> (descriptor ? 1 : 0)) & 1);
}

// grid sizing only (the kernels stride over the real, device-side count): a generous estimate of the keypoints of an octave
uint32_t expected_keypoints(const vksift_hip_OctaveJob *job, uint32_t batch)
{
  static int div = -1;
  if (div < 0)
  {
    div = 512; /* pixels per expected keypoint */
  }
  if (div <= 0 || batch < 8u) /* a handful of images: idle workgroups cost nothing, keep the full parallelism */
    return 0xFFFFFFFFu;
  const uint64_t n = (uint64_t)job->w * job->h / (uint32_t)div;
  return n < 32u ? 32u : (n > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n);
}

} // namespace

// the launches of one run of at most MULTI_MAX octaves
static int orientation_run(const vksift_hip_OctaveJob *jobs, uint32_t n, uint32_t batch, hipStream_t hs)
{
  Multi<FeatArgs> mo, mf;
  mo.n = mf.n = 0;
  const bool f = img_fast(batch, false);
  for (uint32_t i = 0; i < n; i++)
  {
    const FeatArgs a = make_args(&jobs[i]);
    /* The keypoint count is only known on the device; an octave's share of the grid is sized for a dense octave (one keypoint per
     * 512 pixels, twice the usual yield) and strides over whatever is there. Sizing it by the section capacity alone gave the
     * coarse octaves 131 k workgroups per stage for a few hundred keypoints: 40-70 us of pure dispatch each. */
    const uint32_t dense = expected_keypoints(&jobs[i], batch);
    uint32_t blocks = ((jobs[i].cap < dense ? jobs[i].cap : dense) + 3) / 4;
    /* ... and capped so that an octave's share is ~16 k workgroups whatever the batch: the workgroups stride over the keypoints,
     * and beyond that more (mostly idle) workgroups only cost dispatch — 512 frames: 1024 per image 1.97 ms, 128: 1.84, 32: 1.71 */
    uint32_t cap_blocks = 16384u / batch;
    cap_blocks = cap_blocks < 32u ? 32u : (cap_blocks > 1024u ? 1024u : cap_blocks);
    if (blocks > cap_blocks)
      blocks = cap_blocks;
    if (blocks == 0)
      blocks = 1;
    if (!multi_add(mo, a, f ? batch : blocks, f ? blocks : batch, 1u) || !multi_add(mf, a, batch, 1u, 1u))
      return (int)hipErrorInvalidValue;
  }
  const bool f16 = mo.oct[0].fp16 != 0;
  const dim3 grid(mo.start[mo.n]);
  if (f)
  {
    if (f16)
      hipLaunchKernelGGL((k_orientation<true, true>), grid, dim3(256), 0, hs, mo);
    else
      hipLaunchKernelGGL((k_orientation<true, false>), grid, dim3(256), 0, hs, mo);
  }
  else
  {
    if (f16)
      hipLaunchKernelGGL((k_orientation<false, true>), grid, dim3(256), 0, hs, mo);
    else
      hipLaunchKernelGGL((k_orientation<false, false>), grid, dim3(256), 0, hs, mo);
  }
  hipLaunchKernelGGL(k_orientation_finalize, dim3(mf.start[mf.n]), dim3(1024), 0, hs, mf);
  return (int)hipGetLastError();
}

static int descriptor_run(const vksift_hip_OctaveJob *jobs, uint32_t n, uint32_t batch, const vksift_hip_DenseRows &dr, hipStream_t hs)
{
  Multi<FeatArgs> md;
  md.n = 0;
  const bool f = img_fast(batch, true);
  for (uint32_t i = 0; i < n; i++)
  {
    const FeatArgs a = make_args(&jobs[i]);
    const uint32_t dense = expected_keypoints(&jobs[i], batch);
    uint32_t blocks = jobs[i].cap < dense ? jobs[i].cap : dense;
    if (blocks > 2048)
      blocks = 2048;
    if (blocks == 0)
      blocks = 1;
    if (!multi_add(md, a, f ? batch : blocks, f ? blocks : batch, 1u))
      return (int)hipErrorInvalidValue;
  }
  /* waves per keypoint: 4 keep the critical path of a single image short; a batch has keypoints to spare and runs
   * 3-4 % faster with 2 (less redundant per-keypoint work, measured under the overlapped batch schedule) */
  /* waves per keypoint: 4 for single images (latency), 2 for batches (1 wave: 55 % slower, 8: no faster than 4; measured round 2) */
  const int nwv = batch >= 8u ? 2 : 4;
  if (dr.post && nwv != 4)
    return (int)hipErrorInvalidValue; /* posting is compiled into the four-wave form only */
  const bool f16 = md.oct[0].fp16 != 0;
  const dim3 grid(md.start[md.n]);
#define VKSIFT_DESC(N)                                                                  \
  do                                                                                    \
  {                                                                                     \
    if (f)                                                                              \
    {                                                                                   \
      if (f16)                                                                          \
        hipLaunchKernelGGL((k_descriptor<N, true, true>), grid, dim3(64 * N), 0, hs, md, dr);  \
      else                                                                              \
        hipLaunchKernelGGL((k_descriptor<N, true, false>), grid, dim3(64 * N), 0, hs, md, dr); \
    }                                                                                   \
    else                                                                                \
    {                                                                                   \
      if (f16)                                                                          \
        hipLaunchKernelGGL((k_descriptor<N, false, true>), grid, dim3(64 * N), 0, hs, md, dr); \
      else                                                                              \
        hipLaunchKernelGGL((k_descriptor<N, false, false>), grid, dim3(64 * N), 0, hs, md, dr); \
    }                                                                                   \
  } while (0)
  if (nwv == 2)
    VKSIFT_DESC(2);
  else
    VKSIFT_DESC(4);
#undef VKSIFT_DESC
  return (int)hipGetLastError();
}

template <typename F>
static int for_runs(const vksift_hip_OctaveJob *jobs, uint32_t n_jobs, F run)
{
  for (uint32_t i0 = 0; i0 < n_jobs;)
  {
    uint32_t i1 = i0 + 1;
    while (i1 < n_jobs && i1 - i0 < multi_run_max() && jobs[i1].fp16 == jobs[i0].fp16)
      i1++;
    const int e = run(jobs + i0, i1 - i0);
    if (e)
      return e;
    i0 = i1;
  }
  return 0;
}

namespace
{
// test only (vksift_hip_selftest_inrange): the short forms against the compiler's general ones on pseudo-random operands inside their ranges
__device__ __forceinline__ uint32_t st_hash(uint32_t x)
{
  x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
  return x;
}
// 2^e * (1 + m / 2^23) with e uniform in [elo, ehi]
__device__ __forceinline__ float st_float(uint32_t h, int elo, int ehi)
{
  const int e = elo + (int)((h >> 23) % (uint32_t)(ehi - elo + 1));
  return __uint_as_float(((uint32_t)(e + 127) << 23) | (h & 0x7FFFFFu));
}
__global__ void __launch_bounds__(256) k_inrange_selftest(uint32_t n, uint32_t seed, uint32_t *bad)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n)
    return;
  const uint32_t h0 = st_hash(i * 3u + seed), h1 = st_hash(i * 3u + 1u + seed), h2 = st_hash(i * 3u + 2u + seed);
  uint32_t miss = 0;
  // sqrt: 0 and [2^-96, 2^8), the ends of the range in the first lanes
  float x = i == 0 ? 0.f : (i == 1 ? 0x1p-96f : (i == 2 ? __uint_as_float(0x0F800001u) : st_float(h0, -96, 7)));
  asm volatile("" : "+v"(x));
  miss += __float_as_uint(sqrt_inrange(x)) != __float_as_uint(sqrtf(x));
  // a / b: b in [2^-49, 4), a = 0 or in [2^-64, b]
  float b = st_float(h1, -49, 1), a = st_float(h2, -64, 1);
  a = (i & 15u) == 3u ? 0.f : (a > b ? b * __uint_as_float(0x3F000000u | (h2 & 0x7FFFFFu)) : a); // b * [0.5, 1) where the draw exceeds b
  a = a < 0x1p-64f && a != 0.f ? 0x1p-64f : a;
  asm volatile("" : "+v"(a), "+v"(b));
  miss += __float_as_uint(div_inrange(a, b)) != __float_as_uint(a / b);
  // x / 2 pi: +-[2^-96, 2^8)
  float y = st_float(h0 ^ h1, -96, 7);
  y = (h2 & 1u) ? -y : y;
  asm volatile("" : "+v"(y));
  miss += __float_as_uint(div_2pi_inrange(y)) != __float_as_uint(y / (2.f * PI_F));
  // grad_polar's own guard: wherever it does not ask for the general forms, the short forms must give their bits — components from
  // denormals to 4 and exact zeros, incl. (0, denormal), whose squares underflow to 0
  {
    auto comp = [](uint32_t h, uint32_t i) {
      const uint32_t kind = (h >> 27) & 7u;
      float v = kind == 0u ? 0.f : (kind == 1u ? __uint_as_float(h & 0x7FFFFFu) /* denormal */ : __uint_as_float((((h >> 23) & 0xFFu) % 129u) << 23 | (h & 0x7FFFFFu)));
      if (i == 4u)
        v = __uint_as_float(1u);
      return (h & 0x80000000u) ? -v : v;
    };
    float gx = comp(st_hash(h0 ^ 0x9E3779B9u), 0u), gy = comp(st_hash(h1 ^ 0x85EBCA6Bu), i);
    if (i == 4u || i == 5u)
      gx = 0.f;
    asm volatile("" : "+v"(gx), "+v"(gy));
    float o1, l1, o2, l2;
    bool odd;
    grad_polar<true>(gx, gy, &o1, &l1, &odd);
    grad_polar<false>(gx, gy, &o2, &l2, nullptr);
    miss += !odd && (__float_as_uint(o1) != __float_as_uint(o2) || __float_as_uint(l1) != __float_as_uint(l2));
  }
  if (miss)
    atomicAdd(bad, miss);
}
} // namespace

extern "C"
{
  int vksift_hip_selftest_inrange(uint32_t n, uint32_t seed, uint32_t *d_mismatches, vksift_hip_stream s)
  {
    hipError_t e = hipMemsetAsync(d_mismatches, 0, sizeof(uint32_t), (hipStream_t)s);
    if (e != hipSuccess)
      return (int)e;
    hipLaunchKernelGGL(k_inrange_selftest, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)s, n, seed, d_mismatches);
    return (int)hipGetLastError();
  }

  int vksift_hip_orientations_multi(const vksift_hip_OctaveJob *jobs, uint32_t n_jobs, uint32_t batch, vksift_hip_stream s)
  {
    if (batch == 0)
      return 0;
    return for_runs(jobs, n_jobs, [&](const vksift_hip_OctaveJob *j, uint32_t n) { return orientation_run(j, n, batch, (hipStream_t)s); });
  }

  int vksift_hip_descriptors_multi_dense(const vksift_hip_OctaveJob *jobs, uint32_t n_jobs, uint32_t batch, const vksift_hip_DenseRows *dense,
                                         vksift_hip_stream s)
  {
    if (batch == 0)
      return 0;
    vksift_hip_DenseRows dr = {};
    if (dense)
    {
      const bool rows = dense->desc != NULL, post = dense->post != NULL;
      if ((!rows && !post) || (rows && (!dense->norm || !dense->n)) || (post && (!dense->found_post || dense->found_post_n > 64u)) || dense->nsec == 0 ||
          dense->nsec > 16u)
        return (int)hipErrorInvalidValue;
      for (uint32_t i = 0; i < n_jobs; i++)
        if (jobs[i].sec_index >= dense->nsec)
          return (int)hipErrorInvalidValue;
      dr = *dense;
    }
    return for_runs(jobs, n_jobs, [&](const vksift_hip_OctaveJob *j, uint32_t n) { return descriptor_run(j, n, batch, dr, (hipStream_t)s); });
  }

  int vksift_hip_descriptors_multi(const vksift_hip_OctaveJob *jobs, uint32_t n_jobs, uint32_t batch, vksift_hip_stream s)
  {
    return vksift_hip_descriptors_multi_dense(jobs, n_jobs, batch, nullptr, s);
  }

  int vksift_hip_orientations(const vksift_hip_OctaveJob *job, uint32_t batch, vksift_hip_stream s) { return vksift_hip_orientations_multi(job, 1, batch, s); }

  int vksift_hip_descriptors(const vksift_hip_OctaveJob *job, uint32_t batch, vksift_hip_stream s) { return vksift_hip_descriptors_multi(job, 1, batch, s); }
}
