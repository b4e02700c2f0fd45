// pyramid.hip — Gaussian scale-space construction kernels for gfx950 (wave64, LDS-tiled).
//
// Replaces, in the reference (paths relative to src/vulkansift/):
//   vkCmdCopyBufferToImage + vkCmdBlitImage(LINEAR)   sift_detector.c:881, 909-916   -> k_input_blit
//   GaussianBlur*.comp H + V dispatches               sift_detector.c:927-1001       -> k_blur_tile (fused)
//   DifferenceOfGaussian.comp                         sift_detector.c:1039-1079      -> fused into k_blur_tile
//   vkCmdBlitImage(NEAREST) down-sample               sift_detector.c:1003-1034      -> k_downsample
//
// Arithmetic contract (must stay bit-identical to oracle/sift_oracle.c blur_plane/blit_*):
//   pass:  acc = centre*k0;  acc = fmaf(t(+i) + t(-i), k[i], acc)  for i = 1..n-1 ascending
//   compiled with -ffp-contract=off so nothing else is fused.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vksift_hip.h"

namespace
{

struct Taps
{
  float k[VKSIFT_HIP_MAX_TAPS];
};

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
__global__ void __launch_bounds__(256) k_input_blit(const uint8_t *__restrict__ src, int sw, int sh, uint64_t src_img_stride, float *__restrict__ dst,
                                                    int dw, int dh, int dpitch, uint64_t dst_img_stride)
{
  int x = blockIdx.x * 64 + (threadIdx.x & 63);
  int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh)
    return;
  const uint8_t *img = src + (size_t)blockIdx.z * src_img_stride;
  float *out = dst + (size_t)blockIdx.z * dst_img_stride;
  if (dw == sw && dh == sh)
  {
    out[(size_t)y * dpitch + x] = (float)img[(size_t)y * sw + x] / 255.f;
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
  out[(size_t)y * dpitch + x] = fmaf(b, r1, (1.f - b) * r0);
}

// ---------------------------------------------------------------------------------------------
// Fused separable blur (+ optional DoG) over a 64x64 output tile staged through LDS.
//   LDS: s_src[(64+2R)][SS]  source tile with halo (mirrored at the image border)
//        s_mid[(64+2R)][64]  horizontally blurred rows
// 4 waves; lane = column so every LDS access is stride-1 (conflict-free) and every global row
// access is one 256-byte coalesced segment.
// ---------------------------------------------------------------------------------------------
constexpr int TILE = 64;

__global__ void __launch_bounds__(256) k_blur_tile(const float *__restrict__ src, uint64_t src_img_stride, int spitch, float *__restrict__ dst,
                                                   uint64_t dst_img_stride, int dpitch, float *__restrict__ dog, uint64_t dog_img_stride, int gpitch,
                                                   int w, int h, Taps taps, int ntaps)
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
  const float *in = src + (size_t)blockIdx.z * src_img_stride;

  // stage source tile
  for (int r = wave; r < SH; r += 4)
  {
    int gy = mirror_idx(y0 - R + r, h);
    const float *row = in + (size_t)gy * spitch;
    for (int c = lane; c < SW; c += 64)
      s_src[r * SS + c] = row[mirror_idx(x0 - R + c, w)];
  }
  __syncthreads();

  // horizontal pass over all staged rows
  const float k0 = taps.k[0];
  for (int r = wave; r < SH; r += 4)
  {
    const float *p = s_src + r * SS + lane + R;
    float acc = p[0] * k0;
    for (int i = 1; i < ntaps; i++)
      acc = fmaf(p[i] + p[-i], taps.k[i], acc);
    s_mid[r * TILE + lane] = acc;
  }
  __syncthreads();

  // vertical pass, 16 output rows per wave
  const int gx = x0 + lane;
  float *out = dst + (size_t)blockIdx.z * dst_img_stride;
  float *gout = dog ? dog + (size_t)blockIdx.z * dog_img_stride : nullptr;
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
    {
      out[(size_t)gy * dpitch + gx] = acc;
      if (gout)
        gout[(size_t)gy * gpitch + gx] = acc - s_src[(rr + R) * SS + lane + R];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Nearest-neighbour resample (2:1 -> odd source texels), one thread per destination pixel.
// ---------------------------------------------------------------------------------------------
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
  const float *in = src + (size_t)blockIdx.z * src_img_stride;
  float *out = dst + (size_t)blockIdx.z * dst_img_stride;
  out[(size_t)y * dpitch + x] = in[(size_t)yy * spitch + xx];
}

} // namespace

extern "C"
{

  int vksift_hip_input_blit(const uint8_t *src, uint32_t sw, uint32_t sh, uint64_t src_img_stride, vksift_hip_Plane dst, uint32_t batch, vksift_hip_stream s)
  {
    dim3 grid((dst.w + 63) / 64, (dst.h + 3) / 4, batch);
    hipLaunchKernelGGL(k_input_blit, grid, dim3(256), 0, (hipStream_t)s, src, (int)sw, (int)sh, src_img_stride, dst.base, (int)dst.w, (int)dst.h,
                       (int)dst.pitch, dst.img_stride);
    return (int)hipGetLastError();
  }

  int vksift_hip_blur(vksift_hip_Plane src, vksift_hip_Plane dst, vksift_hip_Plane dog, const float *taps, uint32_t ntaps, uint32_t batch,
                      vksift_hip_stream s)
  {
    if (ntaps < 1 || ntaps > VKSIFT_HIP_MAX_TAPS || src.base == dst.base)
      return (int)hipErrorInvalidValue;
    Taps t;
    for (uint32_t i = 0; i < VKSIFT_HIP_MAX_TAPS; i++)
      t.k[i] = i < ntaps ? taps[i] : 0.f;
    int R = (int)ntaps - 1;
    int SH = TILE + 2 * R, SS = TILE + 2 * R + 1;
    size_t lds_bytes = sizeof(float) * ((size_t)SH * SS + (size_t)SH * TILE);
    static bool lds_attr_set = false;
    if (!lds_attr_set)
    {
      /* largest tile (R = 19) needs 68 KiB of the CU's 160 KiB LDS: above the 64 KiB default opt-in limit */
      const int Rm = VKSIFT_HIP_MAX_TAPS - 1;
      const int max_bytes = (int)(sizeof(float) * ((TILE + 2 * Rm) * (TILE + 2 * Rm + 1) + (TILE + 2 * Rm) * TILE));
      hipError_t ae = hipFuncSetAttribute((const void *)k_blur_tile, hipFuncAttributeMaxDynamicSharedMemorySize, max_bytes);
      if (ae != hipSuccess)
        return (int)ae;
      lds_attr_set = true;
    }
    dim3 grid((src.w + TILE - 1) / TILE, (src.h + TILE - 1) / TILE, batch);
    hipLaunchKernelGGL(k_blur_tile, grid, dim3(256), lds_bytes, (hipStream_t)s, src.base, src.img_stride, (int)src.pitch, dst.base, dst.img_stride,
                       (int)dst.pitch, dog.base, dog.img_stride, (int)dog.pitch, (int)src.w, (int)src.h, t, (int)ntaps);
    return (int)hipGetLastError();
  }

  int vksift_hip_downsample(vksift_hip_Plane src, vksift_hip_Plane dst, uint32_t batch, vksift_hip_stream s)
  {
    dim3 grid((dst.w + 63) / 64, (dst.h + 3) / 4, batch);
    hipLaunchKernelGGL(k_downsample, grid, dim3(256), 0, (hipStream_t)s, src.base, src.img_stride, (int)src.w, (int)src.h, (int)src.pitch, dst.base,
                       dst.img_stride, (int)dst.w, (int)dst.h, (int)dst.pitch);
    return (int)hipGetLastError();
  }
}
