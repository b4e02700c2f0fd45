// extrema.hip — DoG extrema detection, sub-pixel refinement and deterministic compaction (gfx950).
//
// Replaces ExtractKeypoints.comp (dispatch: sift_detector.c:1106-1189). The reference appends
// keypoints with a global atomicAdd (ExtractKeypoints.comp:208), which makes their order
// non-deterministic. Here the append is a three-step, atomic-free compaction:
//   k_extrema_detect : one wave per 64-pixel row segment; every lane tests + refines its texel,
//                      the wave's acceptance ballot (one u64) is stored per segment
//   k_segment_scan   : exclusive prefix sum of popcount(mask) over segments in (scale, y, x) order
//   k_extrema_emit   : lanes whose bit is set recompute their record and store it at
//                      offset[segment] + (number of lower set bits)            -> raster order
// The arithmetic of refine_texel() is kept operation-for-operation identical to
// oracle/sift_oracle.c:extract_one (fp32, no contraction) so results are bit-exact.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../detmath.h"
#include "vksift_hip.h"

namespace
{

struct DogView
{
  const float *base; // layer 0
  int w, h, pitch;
  size_t plane; // floats between layers
  int S;
};

// imageLoad with robust out-of-bounds behaviour on the layer axis (quirk Q1): layer S+2 reads 0.
__device__ __forceinline__ float ld(const DogView &d, int s, int x, int y)
{
  if (s < 0 || s > d.S + 1)
    return 0.f;
  return d.base[(size_t)s * d.plane + (size_t)y * d.pitch + x];
}

struct KpRecord
{
  float x, y, scale_x, scale_y;
  uint32_t scale_idx;
  int32_t octave_idx;
  float sigma, orientation, intensity;
};

// 26-neighbour strict extremum test (ExtractKeypoints.comp:56-116).
__device__ __forceinline__ bool is_extremum(const DogView &d, int s, int x, int y, float c)
{
  bool is_max = true, is_min = true;
#pragma unroll
  for (int ds = -1; ds <= 1; ds++)
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
      for (int dx = -1; dx <= 1; dx++)
      {
        if (!ds && !dy && !dx)
          continue;
        float v = d.base[(size_t)(s + ds) * d.plane + (size_t)(y + dy) * d.pitch + (x + dx)];
        is_max = is_max && (c > v);
        is_min = is_min && (c < v);
      }
  return is_max || is_min;
}

// Refinement + acceptance tests (ExtractKeypoints.comp:121-224).
__device__ bool refine_texel(const DogView &d, int x, int y, int s, float dog_threshold, float edge_limit, float seed_sigma, int octave_idx, KpRecord *kp)
{
  const int W = d.w, H = d.h, S = d.S;
  float oX = 0.f, oY = 0.f, oS = 0.f, gX = 0.f, gY = 0.f, gS = 0.f;
  int rx = x, ry = y, rs = s;
  for (int step = 0; step < 5; step++)
  {
    float vc = ld(d, rs, rx, ry);
    float sp = ld(d, rs + 1, rx, ry), sm = ld(d, rs - 1, rx, ry);
    float xp = ld(d, rs, rx + 1, ry), xm = ld(d, rs, rx - 1, ry);
    float yp = ld(d, rs, rx, ry + 1), ym = ld(d, rs, rx, ry - 1);
    gS = 0.5f * (sp - sm);
    gX = 0.5f * (xp - xm);
    gY = 0.5f * (yp - ym);
    float h11 = sp + sm - 2.f * vc;
    float h22 = xp + xm - 2.f * vc;
    float h33 = yp + ym - 2.f * vc;
    float h12 = 0.25f * (ld(d, rs + 1, rx + 1, ry) - ld(d, rs + 1, rx - 1, ry) - ld(d, rs - 1, rx + 1, ry) + ld(d, rs - 1, rx - 1, ry));
    float h13 = 0.25f * (ld(d, rs + 1, rx, ry + 1) - ld(d, rs + 1, rx, ry - 1) - ld(d, rs - 1, rx, ry + 1) + ld(d, rs - 1, rx, ry - 1));
    float h23 = 0.25f * (ld(d, rs, rx + 1, ry + 1) - ld(d, rs, rx + 1, ry - 1) - ld(d, rs, rx - 1, ry + 1) + ld(d, rs, rx - 1, ry - 1));

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
  float vc = ld(d, rs, rx, ry);
  float nv = vc + 0.5f * (gX * oX + gY * oY + gS * oS);
  if (!(fabsf(nv) > dog_threshold && fabsf(oX) < 1.5f && fabsf(oY) < 1.5f && fabsf(oS) < 1.5f && sx >= 0 && sx < (float)W && sy >= 0 && sy < (float)H &&
        ss >= 0 && ss <= (float)(S + 1)))
    return false;
  float e11 = ld(d, rs, rx + 1, ry) + ld(d, rs, rx - 1, ry) - 2.f * vc;
  float e22 = ld(d, rs, rx, ry + 1) + ld(d, rs, rx, ry - 1) - 2.f * vc;
  float e12 = 0.25f * (ld(d, rs, rx + 1, ry + 1) - ld(d, rs, rx + 1, ry - 1) - ld(d, rs, rx - 1, ry + 1) + ld(d, rs, rx - 1, ry - 1));
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

__device__ __forceinline__ bool test_texel(const DogView &d, int x, int y, int s, float dog_threshold, float edge_limit, float seed_sigma, int octave_idx,
                                           KpRecord *kp)
{
  if (!(x >= 1 && x < d.w - 1 && y >= 1 && y < d.h - 1))
    return false;
  float c = d.base[(size_t)s * d.plane + (size_t)y * d.pitch + x];
  if (!(fabsf(c) > dog_threshold * 0.8f))
    return false;
  if (!is_extremum(d, s, x, y, c))
    return false;
  return refine_texel(d, x, y, s, dog_threshold, edge_limit, seed_sigma, octave_idx, kp);
}

struct ExtremaArgs
{
  const float *dog;
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
};

template <bool EMIT>
__global__ void __launch_bounds__(256) k_extrema(ExtremaArgs a)
{
  const int lane = threadIdx.x & 63;
  const int segx = blockIdx.x;
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int sz = blockIdx.z % a.S; // scale - 1
  const int b = blockIdx.z / a.S;
  if (y >= a.h)
    return;
  DogView d{a.dog + (size_t)b * a.img_stride, a.w, a.h, a.pitch, (size_t)a.plane_stride, a.S};
  const size_t seg = ((size_t)sz * a.h + y) * a.nseg + segx + (size_t)b * a.seg_img_stride;
  const int x = segx * 64 + lane;
  KpRecord kp;
  if (!EMIT)
  {
    bool ok = test_texel(d, x, y, sz + 1, a.dog_threshold, a.edge_limit, a.seed_sigma, a.octave_idx, &kp);
    unsigned long long m = __ballot(ok);
    if (lane == 0)
      a.seg_mask[seg] = m;
  }
  else
  {
    unsigned long long m = a.seg_mask[seg];
    if (m == 0ull)
      return;
    if ((m >> lane) & 1ull)
    {
      uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      uint32_t idx = a.seg_off[seg] + rank;
      if (idx < a.cap)
      {
        // recompute (bit-identical) instead of round-tripping candidate records through HBM
        refine_texel(d, x, y, sz + 1, a.dog_threshold, a.edge_limit, a.seed_sigma, a.octave_idx, &kp);
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
    }
  }
}

// Exclusive scan of popcount(mask) over n segments; one 1024-thread block per image.
__global__ void __launch_bounds__(1024) k_segment_scan(const uint64_t *__restrict__ mask, uint32_t *__restrict__ off, uint64_t seg_img_stride, uint32_t n,
                                                       uint32_t *found, uint32_t found_img_stride)
{
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  const int b = blockIdx.x;
  mask += (size_t)b * seg_img_stride;
  off += (size_t)b * seg_img_stride;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0)
    carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n; base += 1024)
  {
    uint32_t i = base + threadIdx.x;
    uint32_t v = i < n ? (uint32_t)__popcll(mask[i]) : 0u;
    // inclusive wave scan
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
    if (i < n)
      off[i] = carry + wave_base + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023)
      carry_s = carry + wave_base + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0)
    found[(size_t)b * found_img_stride] = carry_s;
}

} // namespace

extern "C" int vksift_hip_extract_keypoints(const vksift_hip_OctaveJob *job, uint32_t batch, vksift_hip_stream s)
{
  ExtremaArgs a;
  a.dog = job->dog;
  a.w = (int)job->w, a.h = (int)job->h, a.pitch = (int)job->pitch;
  a.plane_stride = job->plane_stride, a.img_stride = job->img_stride;
  a.S = (int)job->S, a.octave_idx = job->octave_idx;
  a.seed_sigma = job->seed_sigma, a.dog_threshold = job->dog_threshold, a.edge_limit = job->edge_limit;
  a.seg_mask = job->seg_mask, a.seg_off = job->seg_off, a.seg_img_stride = job->seg_img_stride;
  a.nseg = (int)((job->w + 63) / 64);
  a.feats = job->feats, a.feat_img_stride = job->feat_img_stride, a.cap = job->cap;
  a.found = job->found, a.found_img_stride = job->found_img_stride;
  dim3 grid(a.nseg, (job->h + 3) / 4, job->S * batch);
  uint32_t nsegs = job->S * job->h * (uint32_t)a.nseg;
  hipLaunchKernelGGL(k_extrema<false>, grid, dim3(256), 0, (hipStream_t)s, a);
  hipLaunchKernelGGL(k_segment_scan, dim3(batch), dim3(1024), 0, (hipStream_t)s, (const uint64_t *)a.seg_mask, a.seg_off, a.seg_img_stride, nsegs, a.found,
                     a.found_img_stride);
  hipLaunchKernelGGL(k_extrema<true>, grid, dim3(256), 0, (hipStream_t)s, a);
  return (int)hipGetLastError();
}
