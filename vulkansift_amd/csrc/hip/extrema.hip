// extrema.hip — DoG extrema detection, sub-pixel refinement and deterministic compaction (gfx950).
//
// The DoG planes of the reference (DifferenceOfGaussian.comp: D[z] = G[z+1] - G[z]) are not stored by this build: every
// DoG value below is formed where it is consumed as ONE fp32 subtraction of two stored Gaussian texels, which is the
// shader's own expression — bit-identical, and 20 B per octave pixel of HBM writes + 20 B of reads become 24 B of reads.
//
// Replaces ExtractKeypoints.comp (dispatch: sift_detector.c:1106-1189). The reference appends
// keypoints with a global atomicAdd (ExtractKeypoints.comp:208), which makes their order
// non-deterministic. Here the append is an atomic-free pipeline whose output order is raster (scale, y, x):
//   k_extrema_lean   : streaming 26-neighbour test; one u64 candidate ballot per 64-pixel row segment
//   k_segment_scan   : exclusive prefix sum of popcount(mask) over segments in (scale, y, x) order
//   k_cand_list      : one thread per segment writes its candidates' packed (x, y, s) at offset + rank
//   k_refine_flags   : one thread per CANDIDATE (dense waves: no lane idles while a neighbour refines) -> accept flag
//   k_cand_emit      : second level of the scan of the accept counts, accepted candidates recompute their record and store it
//                      at their rank (clamped to the section capacity); the un-clamped count goes to found[]
// The arithmetic of refine_texel() is kept operation-for-operation identical to
// oracle/sift_oracle.c:extract_one (fp32, no contraction) so results are bit-exact.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>

#include "../detmath.h"
#include "multi.h"
#include "vksift_hip.h"

namespace
{

struct DogView
{
  const float *base; // GAUSSIAN layer 0 of the octave: DoG layer s = Gaussian layer s+1 - Gaussian layer s
  int w, h, pitch;
  size_t plane; // texels between layers
  int S;
  int fp16;     // binary16 texels (VKSIFT_PYRAMID_PRECISION_FLOAT16): widened exactly; the DoG image of such a pyramid is binary16 too
};

// imageLoad of the DoG image with robust out-of-bounds behaviour on the layer axis (quirk Q1): layer S+2 reads 0.
// F16 (binary16 texels) is a template parameter of everything that refines: a run-time flag inside this function was
// miscompiled (fp32 results of images >= 1 of a batch changed with unrelated edits of the callers).
template <bool F16>
__device__ __forceinline__ float ld(const DogView &d, int s, int x, int y)
{
  if (s < 0 || s > d.S + 1)
    return 0.f;
  if (F16)
  {
    const _Float16 *p = (const _Float16 *)d.base + (size_t)s * d.plane + (size_t)y * d.pitch + x;
    return (float)(_Float16)((float)p[d.plane] - (float)p[0]);
  }
  const float *p = d.base + (size_t)s * d.plane + (size_t)y * d.pitch + x;
  return p[d.plane] - p[0];
}

struct KpRecord
{
  float x, y, scale_x, scale_y;
  uint32_t scale_idx;
  int32_t octave_idx;
  float sigma, orientation, intensity;
};

// Refinement + acceptance tests (ExtractKeypoints.comp:121-224).
template <bool F16>
__device__ bool refine_texel(const DogView &d, int x, int y, int s, float dog_threshold, float edge_limit, float seed_sigma, int octave_idx, KpRecord *kp)
{
  const int W = d.w, H = d.h, S = d.S;
  float oX = 0.f, oY = 0.f, oS = 0.f, gX = 0.f, gY = 0.f, gS = 0.f;
  int rx = x, ry = y, rs = s;
  for (int step = 0; step < 5; step++)
  {
    float vc = ld<F16>(d, rs, rx, ry);
    float sp = ld<F16>(d, rs + 1, rx, ry), sm = ld<F16>(d, rs - 1, rx, ry);
    float xp = ld<F16>(d, rs, rx + 1, ry), xm = ld<F16>(d, rs, rx - 1, ry);
    float yp = ld<F16>(d, rs, rx, ry + 1), ym = ld<F16>(d, rs, rx, ry - 1);
    gS = 0.5f * (sp - sm);
    gX = 0.5f * (xp - xm);
    gY = 0.5f * (yp - ym);
    float h11 = sp + sm - 2.f * vc;
    float h22 = xp + xm - 2.f * vc;
    float h33 = yp + ym - 2.f * vc;
    float h12 = 0.25f * (ld<F16>(d, rs + 1, rx + 1, ry) - ld<F16>(d, rs + 1, rx - 1, ry) - ld<F16>(d, rs - 1, rx + 1, ry) + ld<F16>(d, rs - 1, rx - 1, ry));
    float h13 = 0.25f * (ld<F16>(d, rs + 1, rx, ry + 1) - ld<F16>(d, rs + 1, rx, ry - 1) - ld<F16>(d, rs - 1, rx, ry + 1) + ld<F16>(d, rs - 1, rx, ry - 1));
    float h23 = 0.25f * (ld<F16>(d, rs, rx + 1, ry + 1) - ld<F16>(d, rs, rx + 1, ry - 1) - ld<F16>(d, rs, rx - 1, ry + 1) + ld<F16>(d, rs, rx - 1, ry - 1));

    float det = h11 * ((h22 * h33) - (h23 * h23)) - h12 * ((h12 * h33) - (h13 * h23)) + h13 * ((h12 * h23) - (h13 * h22));
    if (det == 0.0f)
      return false;
    float i11 = ((h22 * h33) - (h23 * h23)) / det;
    float i12 = -1.f * ((h12 * h33) - (h13 * h23)) / det;
    float i13 = ((h12 * h23) - (h13 * h22)) / det;
    float i22 = ((h11 * h33) - (h13 * h13)) / det;
    float i23 = -1.f * ((h11 * h23) - (h13 * h12)) / det;
    float i33 = ((h11 * h22) - (h12 * h12)) / det;
    oS = -i11 * gS - i12 * gX - i13 * gY;
    oX = -i12 * gS - i22 * gX - i23 * gY;
    oY = -i13 * gS - i23 * gX - i33 * gY;

    if (fabsf(oX) < 0.6f && fabsf(oY) < 0.6f && fabsf(oS) < 0.6f)
      break;
    else if (step < 4)
    {
      rx += ((oX >= 0.6f && rx < (W - 2)) ? 1 : 0) + ((oX <= -0.6f && rx > 1) ? -1 : 0);
      ry += ((oY >= 0.6f && ry < (H - 2)) ? 1 : 0) + ((oY <= -0.6f && ry > 1) ? -1 : 0);
      rs += ((oS >= 0.6f && rs < (S + 1)) ? 1 : 0) + ((oS <= -0.6f && rs > 1) ? -1 : 0);
    }
  }
  float sx = (float)rx + oX, sy = (float)ry + oY, ss = (float)rs + oS;
  float vc = ld<F16>(d, rs, rx, ry);
  float nv = vc + 0.5f * (gX * oX + gY * oY + gS * oS);
  if (!(fabsf(nv) > dog_threshold && fabsf(oX) < 1.5f && fabsf(oY) < 1.5f && fabsf(oS) < 1.5f && sx >= 0 && sx < (float)W && sy >= 0 && sy < (float)H &&
        ss >= 0 && ss <= (float)(S + 1)))
    return false;
  float e11 = ld<F16>(d, rs, rx + 1, ry) + ld<F16>(d, rs, rx - 1, ry) - 2.f * vc;
  float e22 = ld<F16>(d, rs, rx, ry + 1) + ld<F16>(d, rs, rx, ry - 1) - 2.f * vc;
  float e12 = 0.25f * (ld<F16>(d, rs, rx + 1, ry + 1) - ld<F16>(d, rs, rx + 1, ry - 1) - ld<F16>(d, rs, rx - 1, ry + 1) + ld<F16>(d, rs, rx - 1, ry - 1));
  float edgeness = ((e11 + e22) * (e11 + e22)) / ((e11 * e22) - (e12 * e12));
  if (!((edgeness < edge_limit) && (edgeness >= 0)))
    return false;

  float scale_factor = octave_idx >= 0 ? dm_pow2i(octave_idx) : 1.f / dm_pow2i(-octave_idx);
  kp->scale_x = sx;
  kp->scale_y = sy;
  kp->scale_idx = (uint32_t)roundf(ss);
  kp->octave_idx = octave_idx;
  kp->sigma = seed_sigma * dm_exp2f(ss / (float)S) * scale_factor;
  kp->orientation = 0.f;
  kp->intensity = nv;
  kp->x = sx * scale_factor;
  kp->y = sy * scale_factor;
  return true;
}

// The same refinement through a BUFFER RESOURCE over the octave of one image (32-bit byte offsets, the rows as scalar offsets, the
// columns as immediates): refine_texel() above addresses every DoG value with two 64-bit multiply-adds and an exec-mask branch for
// the layer test — 300 of its 700 VALU instructions and 38 branches per iteration are addressing. Here an iteration loads the 28
// GAUSSIAN texels of its neighbourhood once (5 + 9 + 9 + 5 over the four layers; the pointer form loads 38 + 38) and forms the 19
// DoG values from them with the same subtraction; the acceptance tests reuse the last iteration's values (the position does not
// move after the last loads). Same operations on the same operands in the same order: bit-identical. The caller guarantees
// (S + 3) * plane * texel bytes < 2^31.
template <bool F16>
__device__ bool refine_texel_buf(const __amdgpu_buffer_rsrc_t rsrc, int W, int H, int pitch, unsigned plane, int S, int x, int y, int s, float dog_threshold,
                                 float edge_limit, float seed_sigma, int octave_idx, KpRecord *kp)
{
  constexpr unsigned EB = F16 ? 2u : 4u;
  const int pitch_b = pitch * (int)EB;
  const unsigned plane_b = plane * EB;
  auto tex = [&](unsigned base, int row_off) -> float {
    if (F16)
      return (float)__builtin_bit_cast(_Float16, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc, base, row_off, 0));
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, base, row_off, 0));
  };
  auto dog = [&](float hi, float lo) -> float { return F16 ? (float)(_Float16)(hi - lo) : hi - lo; };
  float oX = 0.f, oY = 0.f, oS = 0.f, gX = 0.f, gY = 0.f, gS = 0.f;
  float vc = 0.f, xp = 0.f, xm = 0.f, yp = 0.f, ym = 0.f, h23 = 0.f;
  int rx = x, ry = y, rs = s;
  for (int step = 0; step < 5; step++)
  {
    // byte offset of Gaussian texel (rx - 1, ry - 1) of layer rs; rs >= 1, rx >= 1, ry >= 1 always
    const unsigned b0 = ((unsigned)rs * plane + (unsigned)(ry - 1) * (unsigned)pitch + (unsigned)(rx - 1)) * EB;
    const unsigned bm = b0 - plane_b, b1 = b0 + plane_b, b2 = b1 + plane_b; // layers rs - 1, rs + 1, rs + 2 (past the last layer: reads 0)
    // Gaussian texels: the full 3x3 of layers rs and rs + 1, the cross of layers rs - 1 and rs + 2
    float g0[3][3], g1[3][3];
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int i = 0; i < 3; i++)
      {
        g0[j][i] = tex(b0 + (unsigned)i * EB, j * pitch_b);
        g1[j][i] = tex(b1 + (unsigned)i * EB, j * pitch_b);
      }
    const float gm_c = tex(bm + EB, pitch_b), gm_xm = tex(bm, pitch_b), gm_xp = tex(bm + 2u * EB, pitch_b), gm_ym = tex(bm + EB, 0), gm_yp = tex(bm + EB, 2 * pitch_b);
    const float g2_c = tex(b2 + EB, pitch_b), g2_xm = tex(b2, pitch_b), g2_xp = tex(b2 + 2u * EB, pitch_b), g2_ym = tex(b2 + EB, 0), g2_yp = tex(b2 + EB, 2 * pitch_b);
    // DoG layer rs + 1 exists up to S + 1 (quirk Q1: beyond it the reference's image load returns 0)
    const bool up = rs + 1 <= S + 1;
    const float sp = up ? dog(g2_c, g1[1][1]) : 0.f, sm = dog(g0[1][1], gm_c);
    const float p_xp = up ? dog(g2_xp, g1[1][2]) : 0.f, p_xm = up ? dog(g2_xm, g1[1][0]) : 0.f;
    const float p_yp = up ? dog(g2_yp, g1[2][1]) : 0.f, p_ym = up ? dog(g2_ym, g1[0][1]) : 0.f;
    const float m_xp = dog(g0[1][2], gm_xp), m_xm = dog(g0[1][0], gm_xm), m_yp = dog(g0[2][1], gm_yp), m_ym = dog(g0[0][1], gm_ym);
    float d0[3][3];
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int i = 0; i < 3; i++)
        d0[j][i] = dog(g1[j][i], g0[j][i]);
    vc = d0[1][1];
    xp = d0[1][2], xm = d0[1][0], yp = d0[2][1], ym = d0[0][1];
    gS = 0.5f * (sp - sm);
    gX = 0.5f * (xp - xm);
    gY = 0.5f * (yp - ym);
    float h11 = sp + sm - 2.f * vc;
    float h22 = xp + xm - 2.f * vc;
    float h33 = yp + ym - 2.f * vc;
    float h12 = 0.25f * (p_xp - p_xm - m_xp + m_xm);
    float h13 = 0.25f * (p_yp - p_ym - m_yp + m_ym);
    h23 = 0.25f * (d0[2][2] - d0[0][2] - d0[2][0] + d0[0][0]);

    float det = h11 * ((h22 * h33) - (h23 * h23)) - h12 * ((h12 * h33) - (h13 * h23)) + h13 * ((h12 * h23) - (h13 * h22));
    if (det == 0.0f)
      return false;
    float i11 = ((h22 * h33) - (h23 * h23)) / det;
    float i12 = -1.f * ((h12 * h33) - (h13 * h23)) / det;
    float i13 = ((h12 * h23) - (h13 * h22)) / det;
    float i22 = ((h11 * h33) - (h13 * h13)) / det;
    float i23 = -1.f * ((h11 * h23) - (h13 * h12)) / det;
    float i33 = ((h11 * h22) - (h12 * h12)) / det;
    oS = -i11 * gS - i12 * gX - i13 * gY;
    oX = -i12 * gS - i22 * gX - i23 * gY;
    oY = -i13 * gS - i23 * gX - i33 * gY;

    if (fabsf(oX) < 0.6f && fabsf(oY) < 0.6f && fabsf(oS) < 0.6f)
      break;
    else if (step < 4)
    {
      rx += ((oX >= 0.6f && rx < (W - 2)) ? 1 : 0) + ((oX <= -0.6f && rx > 1) ? -1 : 0);
      ry += ((oY >= 0.6f && ry < (H - 2)) ? 1 : 0) + ((oY <= -0.6f && ry > 1) ? -1 : 0);
      rs += ((oS >= 0.6f && rs < (S + 1)) ? 1 : 0) + ((oS <= -0.6f && rs > 1) ? -1 : 0);
    }
  }
  // (rx, ry, rs) is where the last neighbourhood was loaded: vc, the axis neighbours and h23's diagonal differences are current
  float sx = (float)rx + oX, sy = (float)ry + oY, ss = (float)rs + oS;
  float nv = vc + 0.5f * (gX * oX + gY * oY + gS * oS);
  if (!(fabsf(nv) > dog_threshold && fabsf(oX) < 1.5f && fabsf(oY) < 1.5f && fabsf(oS) < 1.5f && sx >= 0 && sx < (float)W && sy >= 0 && sy < (float)H &&
        ss >= 0 && ss <= (float)(S + 1)))
    return false;
  float e11 = xp + xm - 2.f * vc;
  float e22 = yp + ym - 2.f * vc;
  float e12 = h23;
  float edgeness = ((e11 + e22) * (e11 + e22)) / ((e11 * e22) - (e12 * e12));
  if (!((edgeness < edge_limit) && (edgeness >= 0)))
    return false;

  float scale_factor = octave_idx >= 0 ? dm_pow2i(octave_idx) : 1.f / dm_pow2i(-octave_idx);
  kp->scale_x = sx;
  kp->scale_y = sy;
  kp->scale_idx = (uint32_t)roundf(ss);
  kp->octave_idx = octave_idx;
  kp->sigma = seed_sigma * dm_exp2f(ss / (float)S) * scale_factor;
  kp->orientation = 0.f;
  kp->intensity = nv;
  kp->x = sx * scale_factor;
  kp->y = sy * scale_factor;
  return true;
}

struct ExtremaArgs
{
  const float *gauss; // Gaussian layer 0 of image 0 of the octave (S+3 layers, plane_stride apart)
  int fp16;           // binary16 texels (strides stay in texels)
  int w, h, pitch;
  uint64_t plane_stride, img_stride;
  int S, octave_idx;
  float seed_sigma, dog_threshold, edge_limit;
  uint64_t *seg_mask;
  uint32_t *seg_off;
  uint64_t seg_img_stride;
  int nseg;
  uint8_t *feats;
  uint64_t feat_img_stride;
  uint32_t cap;
  uint32_t *found;
  uint32_t found_img_stride;
  uint32_t *cand_xy;   // packed x | y << 14 | scale << 28, raster order
  uint32_t *cand_flag; // 1 = accepted by the refinement
  uint32_t *cand_n;    // per image: number of candidates (clamped to cand_cap)
  uint64_t cand_img_stride;
  uint32_t cand_cap;
  int band;        // rows per wave of the streaming scan
  uint32_t nsegs;  // S * h * nseg: mask segments of one image
  uint32_t nchunks; // ceil(nsegs / SEG_CHUNK)
  int scan_rev; // the streaming scan walks every XCD's share of the work space back to front (vksift_hip_OctaveJob::scan_reverse)
};

// Streaming detection pass: one wave owns a 64-column segment and marches down a band of
// rows. For every DoG layer it keeps, for the last three rows, the horizontal 3-max / 3-min of its column
// (neighbours come from lane shuffles, the two halo columns from one extra 2-lane load), so the 26-neighbour test of
// a texel of scale s is  c > max over {layers s-1,s,s+1} x {rows y-1,y,y+1} of the 3-max  with the centre's own
// entry replaced by max(left, right) — identical to 26 strict comparisons for finite values. Each DoG plane is read
// exactly once (20 B per octave pixel at S = 3, coalesced) instead of 27 scattered loads per candidate. The
// per-segment candidate ballots feed k_segment_scan / k_cand_list; refinement happens later on dense waves.
// lane i <- lane i-1 (lane 0 keeps `edge`), lane i <- lane i+1 (lane 63 keeps `edge`): gfx9 DPP wave shifts, one VALU op
__device__ __forceinline__ float wave_shr1(float v, float edge)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1(float v, float edge)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// v_max_f32 / v_min_f32 / v_max3_f32 / v_min3_f32 on values known to be finite. fmaxf()/fminf() in IEEE mode first
// canonicalise both operands (one extra v_max_f32 x, x each) to quieten signalling NaNs; DoG planes never hold NaNs.
__device__ __forceinline__ float fmax2(float a, float b)
{
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float fmin2(float a, float b)
{
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float fmax3(float a, float b, float c)
{
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float fmin3(float a, float b, float c)
{
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// spread the low 32 bits of x to the even bit positions of a 64-bit word
__device__ __forceinline__ unsigned long long spread32(unsigned long long x)
{
  x &= 0xFFFFFFFFull;
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x << 2)) & 0x3333333333333333ull;
  x = (x | (x << 1)) & 0x5555555555555555ull;
  return x;
}

// ---------------------------------------------------------------------------------------------
// k_extrema_lean — the streaming 26-neighbour test (a generic form of it, k_extrema_stream, was the fallback until round 5: planes
// stay below 2 GiB — sides of 16384 and more are refused — so nothing ever took it), built around what bounds it. The pass is a pure stream (20 B per octave pixel at S = 3, no reuse beyond a 3-row window), so
// its speed is the number of bytes a CU keeps in flight: the generic kernel holds the horizontal 3-max AND 3-min of three
// rows per layer and column (~100 live registers of window state, one row of loads in flight, and ~700 issued
// instructions per row — 64-bit per-lane address arithmetic, exec-mask branches around every load, window moves).
//   * the window holds the RAW texels (3 rows x (2 columns + 1 halo) per layer — they serve max and min alike); the
//     vertical 3-max / 3-min of a column is formed first, its left / right neighbours then come from ONE lane shift per
//     side, layer and sign. 60-75 registers of state: 4 waves per SIMD with two rows of loads in flight each.
//   * raw buffer loads straight into the window slots: lane-constant byte offset (out-of-range for lanes that must not
//     load: the hardware returns 0), the row offset is one SGPR, one resource per layer — no VALU address arithmetic,
//     no branches. The row loop is unrolled over the NSLOT window slots, so the window rotates by renaming.
//   * rows are clamped and columns are not masked: values outside the image only ever reach the test of non-interior
//     texels, which the (lane-constant) column masks and the scalar row range remove
//   * the comparisons are ballots combined on the scalar unit; the (rare) store of a non-empty ballot interleaves the
//     two column masks with two more ballots instead of 100 scalar bit operations
// 26 strict comparisons == centre above the max / below the min of: the own layer's 8 neighbours
// (max3(left column's vertical max, right column's vertical max, max(up, down))) and the full 3x3 of both adjacent layers.
// ---------------------------------------------------------------------------------------------
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
constexpr unsigned EXT_OOB = 0x80000000u;

template <int S, int NSLOT, bool F16>
__global__ void __launch_bounds__(256, 4) k_extrema_lean(Multi<ExtremaArgs> m, int strip_major)
{
  const VBlock vb = vblock(m);
  const ExtremaArgs &a = m.oct[vb.o];
  const int band = a.band;
  constexpr int NL = S + 2;
  constexpr int EB = F16 ? 2 : 4; // bytes per texel
  constexpr int AHEAD = NSLOT - 3; // rows of loads in flight behind the 3-row window
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // 4 independent waves per block, one row band each
  int b = vb.z;
  int bx = vb.x, by = vb.y * 4 + wv;
  if (strip_major)
  {
    // the 4 waves of a block take 4 adjacent strips of one band (strips fastest over the flattened wave index)
    unsigned blk = vb.y * vb.gx + vb.x;
    if (strip_major & 2)
    {
      // XCD-contiguous order (workgroup n runs on XCD n % 8, each with its own L2): every XCD takes a contiguous range of the
      // (image, band, strip) space, so the bands that share halo rows are read through the same L2
      const unsigned per_img = vb.gx * vb.gy, total = per_img * vb.gz;
      if ((total & 7u) == 0)
      {
        const unsigned n = blk + per_img * vb.z, per = total >> 3, k = n >> 3;
        const unsigned wi = (n & 7u) * per + (a.scan_rev ? per - 1u - k : k);
        b = (int)(wi / per_img);
        blk = wi - (unsigned)b * per_img;
      }
    }
    const unsigned id = blk * 4u + (unsigned)wv, ns = (unsigned)(a.nseg + 1) / 2u;
    by = (int)(id / ns), bx = (int)(id - (unsigned)by * ns);
  }
  const int y0 = by * band;
  if (y0 >= a.h)
    return;
  const int y1 = min(y0 + band, a.h);
  const int x0 = bx * 128, x = x0 + 2 * lane;
  const uint8_t *img = (const uint8_t *)a.gauss + (size_t)b * a.img_stride * EB;
  const int pitch4 = a.pitch * EB; // row pitch in bytes
  __amdgpu_buffer_rsrc_t rs[NL + 1]; // one resource per GAUSSIAN layer
#pragma unroll
  for (int l = 0; l <= NL; l++)
    rs[l] = __builtin_amdgcn_make_buffer_rsrc((void *)(img + (size_t)l * a.plane_stride * EB), 0, a.pitch * a.h * EB, 0x00020000);
  // the pitch is a multiple of 64 texels, x is even: the pair (x, x+1) is inside the row or entirely outside
  const unsigned off2 = x < a.pitch ? (unsigned)x * (unsigned)EB : EXT_OOB;
  const int hx = lane == 0 ? x0 - 1 : x0 + 128;
  const unsigned offh = ((lane == 0 || lane == 63) && hx >= 0 && hx < a.pitch) ? (unsigned)hx * (unsigned)EB : EXT_OOB;
  const unsigned long long colA = __ballot(x >= 1 && x < a.w - 1), colB = __ballot(x + 1 < a.w - 1);
  const float pre = a.dog_threshold * 0.8f;
  const int seg0 = bx * 2;
  const bool has_seg1 = seg0 + 1 < a.nseg;
  const int ylo = max(y0, 1), yhi = min(y1, a.h - 1); // rows tested by this wave
  const bool odd = lane & 1;
  const unsigned half = (unsigned)lane >> 1;

  // a slot receives the NL+1 Gaussian texels of a row; when the row becomes the newest row of the window they are turned
  // into the NL DoG texels in place (entry NL is dead from then on)
  u32x2_t wv2[NSLOT][NL + 1]; // columns x, x+1
  unsigned wh[NSLOT][NL + 1]; // halo column (lanes 0 and 63)
  auto fetch_row = [&](int r, auto SLOT) {
    constexpr int slot = decltype(SLOT)::value;
    const int rr = min(max(r, 0), a.h - 1);
    const int so = rr * pitch4;
#pragma unroll
    for (int l = 0; l <= NL; l++)
    {
      if (F16)
      {
        // two binary16 texels in one dword, the halo texel in the low half of another: half the registers of the fp32 form in flight
        wv2[slot][l].x = __builtin_amdgcn_raw_buffer_load_b32(rs[l], off2, so, 0);
        wh[slot][l] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs[l], offh, so, 0);
      }
      else
      {
        wv2[slot][l] = __builtin_amdgcn_raw_buffer_load_b64(rs[l], off2, so, 0);
        wh[slot][l] = __builtin_amdgcn_raw_buffer_load_b32(rs[l], offh, so, 0);
      }
    }
  };
  typedef _Float16 h2x __attribute__((ext_vector_type(2)));
  auto to_dog = [&](auto SLOT) {
    constexpr int slot = decltype(SLOT)::value;
    if (F16)
    {
      float lo0, lo1, loh;
      {
        const h2x t = __builtin_bit_cast(h2x, wv2[slot][0].x);
        lo0 = (float)t.x, lo1 = (float)t.y;
        loh = (float)__builtin_bit_cast(h2x, wh[slot][0]).x;
      }
#pragma unroll
      for (int l = 0; l < NL; l++)
      {
        const h2x t = __builtin_bit_cast(h2x, wv2[slot][l + 1].x);
        const float hi0 = (float)t.x, hi1 = (float)t.y, hih = (float)__builtin_bit_cast(h2x, wh[slot][l + 1]).x;
        // the DoG image of a binary16 pyramid is a binary16 image: round the difference, keep it widened
        wv2[slot][l].x = __float_as_uint((float)(_Float16)(hi0 - lo0));
        wv2[slot][l].y = __float_as_uint((float)(_Float16)(hi1 - lo1));
        wh[slot][l] = __float_as_uint((float)(_Float16)(hih - loh));
        lo0 = hi0, lo1 = hi1, loh = hih;
      }
      return;
    }
#pragma unroll
    for (int l = 0; l < NL; l++)
    {
      wv2[slot][l].x = __float_as_uint(__uint_as_float(wv2[slot][l + 1].x) - __uint_as_float(wv2[slot][l].x));
      wv2[slot][l].y = __float_as_uint(__uint_as_float(wv2[slot][l + 1].y) - __uint_as_float(wv2[slot][l].y));
      wh[slot][l] = __float_as_uint(__uint_as_float(wh[slot][l + 1]) - __uint_as_float(wh[slot][l]));
    }
  };

  // Row r has just become the newest row of the window (slot P); the centre row is y = r - 1. Slot (P + 1 + AHEAD - 1) ...
  // receives row r + AHEAD: it held row r - 3, which no later test reads.
  auto phase = [&](auto PC, int r) {
    constexpr int P = decltype(PC)::value;
    constexpr int sn = P, sm = (P + NSLOT - 1) % NSLOT, so_ = (P + NSLOT - 2) % NSLOT, sf = (P + AHEAD) % NSLOT;
    if (r + AHEAD <= y1)
      fetch_row(r + AHEAD, std::integral_constant<int, sf>{});
    to_dog(std::integral_constant<int, sn>{}); // row r: Gaussian texels -> DoG texels (every row passes here exactly once)
    const int y = r - 1;
    if (y < ylo || y >= yhi)
      return;
    float ownxA[S], ownnA[S], ownxB[S], ownnB[S], fullxA[NL], fullnA[NL], fullxB[NL], fullnB[NL];
#pragma unroll
    for (int l = 0; l < NL; l++)
    {
      const float ao = __uint_as_float(wv2[so_][l].x), am = __uint_as_float(wv2[sm][l].x), an = __uint_as_float(wv2[sn][l].x);
      const float bo = __uint_as_float(wv2[so_][l].y), bm = __uint_as_float(wv2[sm][l].y), bn = __uint_as_float(wv2[sn][l].y);
      const float ho = __uint_as_float(wh[so_][l]), hm = __uint_as_float(wh[sm][l]), hn = __uint_as_float(wh[sn][l]);
      const float vxa = fmax3(ao, am, an), vna = fmin3(ao, am, an);
      const float vxb = fmax3(bo, bm, bn), vnb = fmin3(bo, bm, bn);
      const float vxh = fmax3(ho, hm, hn), vnh = fmin3(ho, hm, hn);
      const float Lx = wave_shr1(vxb, vxh), Ln = wave_shr1(vnb, vnh); // column x-1 = previous lane's column x+1
      const float Rx = wave_shl1(vxa, vxh), Rn = wave_shl1(vna, vnh); // column x+2 = next lane's column x
      if (l >= 1 && l <= S)
      {
        ownxA[l - 1] = fmax3(Lx, vxb, fmax2(ao, an)), ownnA[l - 1] = fmin3(Ln, vnb, fmin2(ao, an));
        ownxB[l - 1] = fmax3(vxa, Rx, fmax2(bo, bn)), ownnB[l - 1] = fmin3(vna, Rn, fmin2(bo, bn));
      }
      fullxA[l] = fmax3(Lx, vxa, vxb), fullnA[l] = fmin3(Ln, vna, vnb);
      fullxB[l] = fmax3(vxa, vxb, Rx), fullnB[l] = fmin3(vna, vnb, Rn);
    }
#pragma unroll
    for (int sz = 0; sz < S; sz++)
    {
      const int l = sz + 1;
      const float ca = __uint_as_float(wv2[sm][l].x), cb = __uint_as_float(wv2[sm][l].y);
      const float nxa = fmax3(ownxA[sz], fullxA[l - 1], fullxA[l + 1]), nna = fmin3(ownnA[sz], fullnA[l - 1], fullnA[l + 1]);
      const float nxb = fmax3(ownxB[sz], fullxB[l - 1], fullxB[l + 1]), nnb = fmin3(ownnB[sz], fullnB[l - 1], fullnB[l + 1]);
      const unsigned long long ma = (__ballot(ca > nxa) | __ballot(ca < nna)) & __ballot(fabsf(ca) > pre) & colA;
      const unsigned long long mb = (__ballot(cb > nxb) | __ballot(cb < nnb)) & __ballot(fabsf(cb) > pre) & colB;
      if ((ma | mb) != 0ull) // the mask array was zeroed by a memset: only non-empty ballots are written
      {
        // pixel 2i of the wave comes from ma bit i, pixel 2i+1 from mb bit i: lane j of a segment looks up its own bit
        const unsigned lo = odd ? (unsigned)mb : (unsigned)ma, hi = odd ? (unsigned)(mb >> 32) : (unsigned)(ma >> 32);
        const unsigned long long m0 = __ballot((lo >> half) & 1u), m1 = __ballot((hi >> half) & 1u);
        if (lane == 0)
        {
          const size_t base = ((size_t)sz * a.h + y) * a.nseg + (size_t)b * a.seg_img_stride;
          a.seg_mask[base + seg0] = m0;
          if (has_seg1)
            a.seg_mask[base + seg0 + 1] = m1;
        }
      }
    }
  };

  // rows y0-1 .. y0-2+AHEAD are in flight before the loop; phase k handles row r = y0 - 1 + k
  if (AHEAD >= 1)
    fetch_row(y0 - 1, std::integral_constant<int, 0>{});
  if (AHEAD >= 2)
    fetch_row(y0, std::integral_constant<int, 1 % NSLOT>{});
  int r = y0 - 1;
  for (;;)
  {
    phase(std::integral_constant<int, 0>{}, r);
    if (++r > y1)
      break;
    phase(std::integral_constant<int, 1>{}, r);
    if (++r > y1)
      break;
    phase(std::integral_constant<int, 2>{}, r);
    if (++r > y1)
      break;
    phase(std::integral_constant<int, 3>{}, r);
    if (++r > y1)
      break;
    if (NSLOT > 4)
    {
      phase(std::integral_constant<int, 4 % NSLOT>{}, r);
      if (++r > y1)
        break;
    }
  }
}

// Exclusive scan of popcount(mask) over the n segments of an image, in two parallel levels (a single workgroup per image
// made this the longest kernel of a 1080p detection): every 1024-thread workgroup scans one chunk of SEG_CHUNK segments
// locally and publishes the chunk total; the consumer (k_cand_list) adds up the totals in front of its chunk (in place)
// and adds base + local offset.
constexpr uint32_t SEG_CHUNK = 4096;

__global__ void __launch_bounds__(1024) k_segment_scan(Multi<ExtremaArgs> m)
{
  __shared__ uint32_t wave_tot[16];
  const VBlock vb = vblock(m); // virtual grid (chunks, images)
  const ExtremaArgs &a = m.oct[vb.o];
  const int b = (int)vb.y;
  const uint64_t *__restrict__ mask = a.seg_mask + (size_t)b * a.seg_img_stride;
  uint32_t *__restrict__ off = a.seg_off + (size_t)b * a.seg_img_stride;
  uint32_t *__restrict__ chunk_tot = a.cand_flag; // the chunk totals / bases live at the start of the flag array until the refinement overwrites it
  const uint64_t chunk_img_stride = a.cand_img_stride;
  const uint32_t n = a.nsegs;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t i0 = vb.x * SEG_CHUNK + 4u * threadIdx.x;
  uint32_t p[4];
#pragma unroll
  for (int k = 0; k < 4; k++)
    p[k] = i0 + k < n ? (uint32_t)__popcll(mask[i0 + k]) : 0u;
  const uint32_t tsum = p[0] + p[1] + p[2] + p[3];
  uint32_t incl = tsum;
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
  uint32_t wave_base = 0, total = 0;
  for (int wv = 0; wv < 16; wv++)
  {
    if (wv < wave)
      wave_base += wave_tot[wv];
    total += wave_tot[wv];
  }
  uint32_t run = wave_base + incl - tsum;
#pragma unroll
  for (int k = 0; k < 4; k++)
  {
    if (i0 + k < n)
      off[i0 + k] = run;
    run += p[k];
  }
  if (threadIdx.x == 0)
    chunk_tot[(size_t)b * chunk_img_stride + vb.x] = total;
}

// The second level of the two-level scans lives in the consumers since round 6: the lists are short (an image has nsegs / 4096 segment
// chunks and candidates / 256 refinement chunks), and adding up the entries in front of one's own chunk costs less than the launch of a
// scan kernel did (k_chunk_offsets: two launches of ~5 us each on the critical path of a single-image detection).

// One thread per 64-pixel segment: expand its candidate ballot into packed coordinates at offset + rank.
__global__ void __launch_bounds__(256) k_cand_list(Multi<ExtremaArgs> mu)
{
  const VBlock vb = vblock(mu); // virtual grid (segment blocks, images)
  const ExtremaArgs &a = mu.oct[vb.o];
  const uint32_t seg = vb.x * 256 + threadIdx.x;
  const int b = (int)vb.y;
  // (k_segment_scan left the chunk totals at the start of the still unused flag array)
  const uint32_t *chunk_tot = a.cand_flag + (size_t)b * a.cand_img_stride;
  if (vb.x == 0 && threadIdx.x == 0)
  {
    // the first workgroup of an image posts the candidate count: the totals of all its chunks (15 for a 1280x960 octave)
    uint32_t total = 0;
    for (uint32_t c = 0; c < a.nchunks; c++)
      total += chunk_tot[c];
    a.cand_n[b] = total;
    if (total == 0)
      a.found[(size_t)b * a.found_img_stride] = 0; // no candidate: k_cand_emit visits no chunk of this image
  }
  unsigned long long m = seg < a.nsegs ? a.seg_mask[seg + (size_t)b * a.seg_img_stride] : 0ull;
  if (__ballot(m != 0ull) == 0ull)
    return; // (wave-uniform) no candidate in these 64 segments: most waves of the launch
  // offset inside the chunk (k_segment_scan) + base of the chunk = the totals of the chunks in front of it: a wave's 64 segments lie in one
  // chunk, so the wave adds the totals up together — lane c loads chunk c (one load for up to 64 chunks), a butterfly sums them. No barrier,
  // no dependent chain of loads (a per-lane loop over the 15 totals of a 1280x960 octave was one).
  uint32_t base = 0;
  for (uint32_t c = threadIdx.x & 63u, nc = (vb.x * 256u) / SEG_CHUNK; c < nc; c += 64u)
    base += chunk_tot[c];
#pragma unroll
  for (int dlt = 32; dlt >= 1; dlt >>= 1)
    base += __shfl_xor(base, dlt, 64);
  if (m == 0ull)
    return;
  uint32_t pos = a.seg_off[seg + (size_t)b * a.seg_img_stride] + base;
  const uint32_t segx = seg % (uint32_t)a.nseg;
  const uint32_t yy = (seg / (uint32_t)a.nseg) % (uint32_t)a.h;
  const uint32_t sz = seg / ((uint32_t)a.nseg * (uint32_t)a.h);
  uint32_t *out = a.cand_xy + (size_t)b * a.cand_img_stride;
  while (m)
  {
    const int bit = __ffsll((long long)m) - 1;
    m &= m - 1;
    if (pos < a.cand_cap)
      out[pos] = (segx * 64u + (uint32_t)bit) | (yy << 14) | ((sz + 1u) << 28);
    pos++;
  }
}

// Dense refinement: thread t of a 256-candidate chunk refines candidate chunk*256 + t (count read from HBM, workgroups
// stride over the chunks). Besides the accept flags every chunk publishes its number of accepted candidates (into the
// segment-offset array, free again after k_cand_list) for the second scan level inside k_cand_emit.
template <bool F16, bool BUF>
__global__ void __launch_bounds__(256) k_refine_flags(Multi<ExtremaArgs> m)
{
  __shared__ uint32_t s_cnt[4];
  const VBlock vb = vblock(m); // virtual grid (images, chunks)
  const ExtremaArgs &a = m.oct[vb.o];
  const int b = (int)vb.x; // image index fastest: see the launch
  uint32_t n = a.cand_n[b];
  n = n < a.cand_cap ? n : a.cand_cap;
  const uint32_t nch = (n + 255u) / 256u;
  DogView d{(const float *)((const uint8_t *)a.gauss + (size_t)b * a.img_stride * (a.fp16 ? 2u : 4u)), a.w, a.h, a.pitch, (size_t)a.plane_stride, a.S, a.fp16};
  // the octave of this image as one buffer (BUF: the launcher has checked that it stays below 2 GiB)
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)d.base, 0, (int)((unsigned)(a.S + 3) * (unsigned)a.plane_stride * (F16 ? 2u : 4u)), 0x00020000);
  const uint32_t *xy = a.cand_xy + (size_t)b * a.cand_img_stride;
  uint32_t *flag = a.cand_flag + (size_t)b * a.cand_img_stride;
  uint32_t *chunk_sum = a.seg_off + (size_t)b * a.seg_img_stride;
  for (uint32_t chunk = vb.y; chunk < nch; chunk += vb.gy)
  {
    const uint32_t i = chunk * 256u + threadIdx.x;
    bool ok = false;
    if (i < n)
    {
      const uint32_t c = xy[i];
      KpRecord kp;
      const int cx = (int)(c & 0x3fffu), cy = (int)((c >> 14) & 0x3fffu), cs = (int)(c >> 28);
      ok = BUF ? refine_texel_buf<F16>(rsrc, a.w, a.h, a.pitch, (unsigned)a.plane_stride, a.S, cx, cy, cs, a.dog_threshold, a.edge_limit, a.seed_sigma, a.octave_idx, &kp)
               : refine_texel<F16>(d, cx, cy, cs, a.dog_threshold, a.edge_limit, a.seed_sigma, a.octave_idx, &kp);
      flag[i] = ok ? 1u : 0u;
    }
    const unsigned long long bal = __ballot(ok);
    if ((threadIdx.x & 63) == 0)
      s_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0)
      chunk_sum[chunk] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    __syncthreads();
  }
}

// Accepted candidates recompute their record (bit-identical) and store it at chunk base + rank inside the chunk (raster
// order is preserved) if it fits the section. About one candidate in six is accepted: the accepted ones of a chunk are
// first compacted through LDS, so the recomputation runs on dense lanes (one wave per chunk instead of four sparse ones).
template <bool F16, bool BUF>
__global__ void __launch_bounds__(256) k_cand_emit(Multi<ExtremaArgs> m)
{
  __shared__ uint32_t s_cnt[4];
  __shared__ uint32_t s_red[4];
  __shared__ uint32_t s_list[256];
  const VBlock vb = vblock(m); // virtual grid (images, chunks)
  const ExtremaArgs &a = m.oct[vb.o];
  const int b = (int)vb.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t n = a.cand_n[b];
  n = n < a.cand_cap ? n : a.cand_cap;
  const uint32_t nch = (n + 255u) / 256u;
  DogView d{(const float *)((const uint8_t *)a.gauss + (size_t)b * a.img_stride * (a.fp16 ? 2u : 4u)), a.w, a.h, a.pitch, (size_t)a.plane_stride, a.S, a.fp16};
  // the octave of this image as one buffer (BUF: the launcher has checked that it stays below 2 GiB)
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)d.base, 0, (int)((unsigned)(a.S + 3) * (unsigned)a.plane_stride * (F16 ? 2u : 4u)), 0x00020000);
  const uint32_t *xy = a.cand_xy + (size_t)b * a.cand_img_stride;
  const uint32_t *flag = a.cand_flag + (size_t)b * a.cand_img_stride;
  const uint32_t *chunk_sum = a.seg_off + (size_t)b * a.seg_img_stride; // accepted candidates per chunk (k_refine_flags)
  for (uint32_t chunk = vb.y; chunk < nch; chunk += vb.gy)
  {
    // records of the chunks in front of this one (raster order is preserved): partial sums of their accept counts ride on the barrier
    // the rank computation needs anyway; the image's last chunk posts the keypoint count
    uint32_t part = 0;
    for (uint32_t c = threadIdx.x; c < chunk; c += 256u)
      part += chunk_sum[c];
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1)
      part += __shfl_xor(part, dlt, 64);
    const uint32_t i = chunk * 256u + threadIdx.x;
    const bool v = i < n && flag[i] != 0u;
    const unsigned long long bal = __ballot(v);
    if (lane == 0)
      s_cnt[wave] = (uint32_t)__popcll(bal), s_red[wave] = part;
    __syncthreads();
    const uint32_t cbase = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    uint32_t rank = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    for (int wv = 0; wv < wave; wv++)
      rank += s_cnt[wv];
    const uint32_t total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (v)
      s_list[rank] = xy[i];
    __syncthreads();
    const uint32_t idx = cbase + threadIdx.x;
    if (chunk + 1u == nch && threadIdx.x == 0)
      a.found[(size_t)b * a.found_img_stride] = cbase + total; // un-clamped, like nb_elem (0 candidates: the counter reset's 0 stays)
    if (threadIdx.x < total && idx < a.cap)
    {
      const uint32_t c = s_list[threadIdx.x];
      KpRecord kp;
      const int cx = (int)(c & 0x3fffu), cy = (int)((c >> 14) & 0x3fffu), cs = (int)(c >> 28);
      if (BUF)
        refine_texel_buf<F16>(rsrc, a.w, a.h, a.pitch, (unsigned)a.plane_stride, a.S, cx, cy, cs, a.dog_threshold, a.edge_limit, a.seed_sigma, a.octave_idx, &kp);
      else
        refine_texel<F16>(d, cx, cy, cs, a.dog_threshold, a.edge_limit, a.seed_sigma, a.octave_idx, &kp);
      uint32_t *rec = (uint32_t *)(a.feats + (size_t)b * a.feat_img_stride + (size_t)idx * 164);
      rec[0] = __float_as_uint(kp.x);
      rec[1] = __float_as_uint(kp.y);
      rec[2] = __float_as_uint(kp.scale_x);
      rec[3] = __float_as_uint(kp.scale_y);
      rec[4] = kp.scale_idx;
      rec[5] = (uint32_t)kp.octave_idx;
      rec[6] = __float_as_uint(kp.sigma);
      rec[7] = __float_as_uint(kp.orientation);
      rec[8] = __float_as_uint(kp.intensity);
    }
    __syncthreads();
  }
}

} // namespace

static int make_extrema_args(const vksift_hip_OctaveJob *job, ExtremaArgs *out)
{
  if (job->w >= 16384u || job->h >= 16384u || job->S > 14u)
    return (int)hipErrorInvalidValue; /* candidate coordinates are packed 14 + 14 + 4 bits */
  ExtremaArgs a;
  a.gauss = job->gauss;
  a.fp16 = (int)job->fp16;
  a.w = (int)job->w, a.h = (int)job->h, a.pitch = (int)job->pitch;
  a.plane_stride = job->plane_stride, a.img_stride = job->img_stride;
  a.S = (int)job->S, a.octave_idx = job->octave_idx;
  a.seed_sigma = job->seed_sigma, a.dog_threshold = job->dog_threshold, a.edge_limit = job->edge_limit;
  a.seg_mask = job->seg_mask, a.seg_off = job->seg_off, a.seg_img_stride = job->seg_img_stride;
  a.nseg = (int)((job->w + 63) / 64);
  a.feats = job->feats, a.feat_img_stride = job->feat_img_stride, a.cap = job->cap;
  a.found = job->found, a.found_img_stride = job->found_img_stride;
  a.cand_xy = job->cand_xy, a.cand_flag = job->cand_flag, a.cand_n = job->cand_n;
  a.cand_img_stride = job->cand_img_stride, a.cand_cap = job->cand_cap;
  a.scan_rev = (int)job->scan_reverse;
  /* 48 rows per wave on the large octaves (2 halo rows per band: 4 %; swept in the pipeline, 512 frames: 32 rows 3.91 ms, 40 3.85, 48 3.83,
   * 56 3.84, 64 3.89); 16 on the small ones, whose share of a launch is latency bound and gains more from twice the waves */
  a.band = job->h > 256u ? 48 : 16;
  if (vksift_hip_tune_get(VKSIFT_TUNE_SCAN_BAND) > 0 && job->h > 256u)
    a.band = vksift_hip_tune_get(VKSIFT_TUNE_SCAN_BAND);
  a.nsegs = job->S * job->h * (uint32_t)a.nseg;
  a.nchunks = (a.nsegs + SEG_CHUNK - 1u) / SEG_CHUNK;
  /* the per-image mask regions are contiguous (the clear below is one fill per octave), the chunk bases fit the flag array */
  if (job->seg_img_stride != a.nsegs || a.nchunks > a.cand_cap)
    return (int)hipErrorInvalidValue;
  *out = a;
  return 0;
}

/* one fill per run of octaves whose mask regions follow each other (the instance's layout) */
static int clear_masks(const ExtremaArgs *args, uint32_t n, uint32_t batch, hipStream_t hs)
{
  for (uint32_t i = 0; i < n;)
  {
    uint32_t j = i + 1;
    size_t bytes = sizeof(uint64_t) * (size_t)args[i].nsegs * batch;
    while (j < n && (const uint8_t *)args[j].seg_mask == (const uint8_t *)args[i].seg_mask + bytes)
      bytes += sizeof(uint64_t) * (size_t)args[j].nsegs * batch, j++;
    const hipError_t me = hipMemsetAsync(args[i].seg_mask, 0, bytes, hs);
    if (me != hipSuccess)
      return (int)me;
    i = j;
  }
  return 0;
}

/* the launches of one run of at most MULTI_MAX octaves (same S, same texel type) */
static int extract_run(const vksift_hip_OctaveJob *jobs, uint32_t n, uint32_t batch, hipStream_t hs, hipEvent_t scan_done)
{
  ExtremaArgs args[MULTI_MAX];
  for (uint32_t i = 0; i < n; i++)
  {
    const int e = make_extrema_args(&jobs[i], &args[i]);
    if (e)
      return e;
  }
  const bool f16 = args[0].fp16 != 0;
  const int S = args[0].S;
  {
    /* A launch that cannot fill the chip (one image, a handful) is bound by the length of a wave's march, not by bandwidth: 16-row bands
     * everywhere give it twice the waves at half the length (one 640x480 image: the scan 35 -> ~20 us, detection 0.367 -> 0.347 ms) */
    uint64_t waves = 0;
    for (uint32_t i = 0; i < n; i++)
      waves += (uint64_t)batch * ((args[i].w + 127) / 128) * ((args[i].h + args[i].band - 1) / args[i].band);
    if (waves < 2048u && vksift_hip_tune_get(VKSIFT_TUNE_SCAN_BAND) <= 0)
      for (uint32_t i = 0; i < n; i++)
        args[i].band = 16;
  }

  /* 1. candidate ballots: only non-empty 64-pixel segments are stored (scattered 8-byte stores were the bottleneck of this
   * pass), so the mask arrays are cleared first: one fill when the octaves' regions follow each other (the instance's layout) */
  bool cleared = true;
  for (uint32_t i = 0; i < n; i++)
    cleared = cleared && jobs[i].masks_cleared != 0;
  if (!cleared)
  {
    const int me = clear_masks(args, n, batch, hs);
    if (me)
      return me;
  }
  /* bit 0: the 4 waves of a block take adjacent strips (-5 % against adjacent bands), bit 1: XCD-contiguous block order (-3 %) */
  const int sm = 3;
  /* window slots: fp32 texels 4 (one row of loads in flight; two rows spill registers since the slots receive S+3 Gaussian
   * texels), binary16 texels 5 (a row in flight costs half the registers) */
  /* the kernel addresses a plane with 32-bit byte offsets: planes stay below 2 GiB (sides of 16384 and more are refused above) */
  for (uint32_t i = 0; i < n; i++)
    if ((uint64_t)jobs[i].pitch * jobs[i].h * (f16 ? 2u : 4u) >= 0x80000000ull)
      return (int)hipErrorInvalidValue;

#define VKSIFT_MULTI(M, GX, GY, GZ)                              \
  Multi<ExtremaArgs> M;                                           \
  M.n = 0;                                                        \
  for (uint32_t i = 0; i < n; i++)                                \
  {                                                               \
    const ExtremaArgs &a = args[i];                               \
    (void)a;                                                      \
    if (!multi_add(M, args[i], (GX), (GY), (GZ)))                 \
      return (int)hipErrorInvalidValue;                           \
  }
  {
    VKSIFT_MULTI(ms, (uint32_t)(a.nseg + 1) / 2u, ((a.h + a.band - 1) / a.band + 3u) / 4u, batch)
    const dim3 sgrid(ms.start[ms.n]);
    switch (S)
    {
#define VKSIFT_CASE(N)                                                                \
  case N:                                                                             \
    if (f16)                                                                          \
      hipLaunchKernelGGL((k_extrema_lean<N, 5, true>), sgrid, dim3(256), 0, hs, ms, sm);  \
    else                                                                              \
      hipLaunchKernelGGL((k_extrema_lean<N, 4, false>), sgrid, dim3(256), 0, hs, ms, sm); \
    break;
      VKSIFT_CASE(1) VKSIFT_CASE(2) VKSIFT_CASE(3) VKSIFT_CASE(4) VKSIFT_CASE(5) VKSIFT_CASE(6) VKSIFT_CASE(7) VKSIFT_CASE(8) VKSIFT_CASE(9)
      VKSIFT_CASE(10) VKSIFT_CASE(11) VKSIFT_CASE(12) VKSIFT_CASE(13)
#undef VKSIFT_CASE
    default:
      return (int)hipErrorInvalidValue;
    }
  }
  if (scan_done)
    (void)hipEventRecord(scan_done, hs);
  /* 2. offsets + candidate count: chunk-local scan, then the (short) scan of the chunk totals; the totals/bases live at
   * the start of the flag array until the refinement overwrites it */
  {
    VKSIFT_MULTI(m2, a.nchunks, batch, 1u)
    hipLaunchKernelGGL(k_segment_scan, dim3(m2.start[m2.n]), dim3(1024), 0, hs, m2);
  }
  /* 3. compact list, 4. dense refinement (+ per-chunk accept counts), 5. scan of those counts, 6. accepted -> records */
  {
    VKSIFT_MULTI(m3, (a.nsegs + 255u) / 256u, batch, 1u)
    hipLaunchKernelGGL(k_cand_list, dim3(m3.start[m3.n]), dim3(256), 0, hs, m3);
  }
  /* virtual grid = (image, chunk): the busy workgroups (chunk < candidates / 256, a small and unknown part of the grid) are then
   * contiguous in dispatch order. With the chunk index fastest they formed a short run at the start of every image's row of
   * 512 workgroups, which the dispatcher's round-robin maps onto the same half of the shader engines of every XCD:
   * measured 221 us instead of 70 us for this launch (and 137 instead of 51 us for k_cand_emit). */
  /* at most 512 chunk workgroups per image, and ~64 k per octave whatever the batch (they stride over the chunks; idle ones only
   * cost dispatch: 512 frames with 512 per image 1.25 ms for the stage without the scan, with 128 per image 0.86 ms) */
  const uint32_t rcap = 65536u / batch < 16u ? 16u : (65536u / batch > 512u ? 512u : 65536u / batch);
  VKSIFT_MULTI(mr, batch, ((a.cand_cap + 255u) / 256u) > rcap ? rcap : ((a.cand_cap + 255u) / 256u), 1u)
  const dim3 rgrid(mr.start[mr.n]);
  /* the refinement addresses an image's octave through one buffer resource with 32-bit offsets where it fits (always, short of
   * 4096 x 4096 octaves with many scales); the pointer form serves the rest */
  bool buf = vksift_hip_tune_get(VKSIFT_TUNE_REFINE_PTR) == 0;
  for (uint32_t i = 0; i < n; i++)
    buf = buf && (uint64_t)(args[i].S + 3) * args[i].plane_stride * (f16 ? 2u : 4u) < 0x7FFF0000ull;
  if (f16 && buf)
    hipLaunchKernelGGL((k_refine_flags<true, true>), rgrid, dim3(256), 0, hs, mr);
  else if (f16)
    hipLaunchKernelGGL((k_refine_flags<true, false>), rgrid, dim3(256), 0, hs, mr);
  else if (buf)
    hipLaunchKernelGGL((k_refine_flags<false, true>), rgrid, dim3(256), 0, hs, mr);
  else
    hipLaunchKernelGGL((k_refine_flags<false, false>), rgrid, dim3(256), 0, hs, mr);
  if (f16 && buf)
    hipLaunchKernelGGL((k_cand_emit<true, true>), rgrid, dim3(256), 0, hs, mr);
  else if (f16)
    hipLaunchKernelGGL((k_cand_emit<true, false>), rgrid, dim3(256), 0, hs, mr);
  else if (buf)
    hipLaunchKernelGGL((k_cand_emit<false, true>), rgrid, dim3(256), 0, hs, mr);
  else
    hipLaunchKernelGGL((k_cand_emit<false, false>), rgrid, dim3(256), 0, hs, mr);
#undef VKSIFT_MULTI
  return (int)hipGetLastError();
}

extern "C" int vksift_hip_extract_keypoints_multi(const vksift_hip_OctaveJob *jobs, uint32_t n_jobs, uint32_t batch, vksift_hip_stream s,
                                                  vksift_hip_event scan_done)
{
  if (n_jobs == 0 || batch == 0)
    return 0;
  /* runs of octaves that one launch can serve: same S and texel type (always the case inside one detection), at most MULTI_MAX */
  for (uint32_t i0 = 0; i0 < n_jobs;)
  {
    uint32_t i1 = i0 + 1;
    while (i1 < n_jobs && i1 - i0 < multi_run_max() && jobs[i1].S == jobs[i0].S && jobs[i1].fp16 == jobs[i0].fp16)
      i1++;
    const int e = extract_run(jobs + i0, i1 - i0, batch, (hipStream_t)s, i1 == n_jobs ? (hipEvent_t)scan_done : nullptr);
    if (e)
      return e;
    i0 = i1;
  }
  return 0;
}

extern "C" int vksift_hip_clear_segment_masks(const vksift_hip_OctaveJob *jobs, uint32_t n_jobs, uint32_t batch, vksift_hip_stream s)
{
  if (n_jobs == 0 || batch == 0)
    return 0;
  if (n_jobs > 16)
    return (int)hipErrorInvalidValue;
  ExtremaArgs args[16];
  for (uint32_t i = 0; i < n_jobs; i++)
  {
    const int e = make_extrema_args(&jobs[i], &args[i]);
    if (e)
      return e;
  }
  return clear_masks(args, n_jobs, batch, (hipStream_t)s);
}

extern "C" int vksift_hip_extract_keypoints(const vksift_hip_OctaveJob *job, uint32_t batch, vksift_hip_stream s, vksift_hip_event scan_done)
{
  return vksift_hip_extract_keypoints_multi(job, 1, batch, s, scan_done);
}
