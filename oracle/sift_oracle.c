/*
 * sift_oracle.c — CPU oracle for the vksift detect/match hot path.  TEST INFRASTRUCTURE ONLY
 * (see sift_oracle.h: "PARITY UNPINNED"). Plain C99, fp32 throughout, no vectorisation tricks:
 * readability against the reference source is the point. Every function names the reference
 * lines it restates (paths relative to /root/reference/src/vulkansift/).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile). Fused multiply-adds are
 * written as fmaf() where this restatement chooses to fuse; nothing else may be contracted.
 *
 * Fixed-function semantics that are not in the reference tree (Vulkan 1.x specification):
 *   - R8_UNORM -> float conversion: v / 255
 *   - vkCmdBlitImage coordinate mapping u = (i + 0.5) * src/dst, LINEAR: bilinear about u - 0.5 with
 *     clamp-to-edge; NEAREST: floor(u)
 *   - sampler MIRRORED_REPEAT addressing; image loads outside the image return 0 (robust access)
 */
#include "sift_oracle.h"

#include "../vulkansift_amd/csrc/detmath.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PI_F 3.14159265358979323846f /* GLSL "#define PI 3.14159265358979323846" is a float literal */

/* ------------------------------------------------------------------------------------------- */
/* math back-ends                                                                              */
/* ------------------------------------------------------------------------------------------- */
typedef struct
{
  float (*exp_)(float);
  float (*exp2_)(float);
  float (*atan2_)(float, float);
  void (*sincos_)(float, float *, float *);
  int (*ceil_log2_)(float);
  int det;
} MathOps;

static float libm_exp(float x) { return expf(x); }
static float libm_exp2(float x) { return powf(2.f, x); }
static float libm_atan2(float y, float x) { return atan2f(y, x); }
static void libm_sincos(float t, float *s, float *c)
{
  *s = sinf(t);
  *c = cosf(t);
}
static int libm_ceil_log2(float m) { return (int)ceilf(log2f(m)); }

static float det_exp(float x) { return dm_expf(x); }
static float det_exp2(float x) { return dm_exp2f(x); }
static float det_atan2(float y, float x) { return dm_atan2f(y, x); }
static void det_sincos(float t, float *s, float *c) { dm_sincosf(t, s, c); }
static int det_ceil_log2(float m) { return dm_ceil_log2f(m); }

static MathOps get_math(const orc_Config *cfg)
{
  MathOps m;
  if (cfg->math_mode == 1)
  {
    m.exp_ = det_exp, m.exp2_ = det_exp2, m.atan2_ = det_atan2, m.sincos_ = det_sincos, m.ceil_log2_ = det_ceil_log2, m.det = 1;
  }
  else
  {
    m.exp_ = libm_exp, m.exp2_ = libm_exp2, m.atan2_ = libm_atan2, m.sincos_ = libm_sincos, m.ceil_log2_ = libm_ceil_log2, m.det = 0;
  }
  return m;
}

/* ------------------------------------------------------------------------------------------- */
/* configuration + host maths                                                                  */
/* ------------------------------------------------------------------------------------------- */
/* vulkansift.c:47-64 */
void orc_default_config(orc_Config *cfg)
{
  cfg->input_image_max_size = 1920u * 1080u;
  cfg->max_nb_sift_per_buffer = 100000u;
  cfg->use_input_upsampling = 1;
  cfg->nb_octaves = 0;
  cfg->nb_scales_per_octave = 3;
  cfg->input_image_blur_level = 0.5f;
  cfg->seed_scale_sigma = 1.6f;
  cfg->intensity_threshold = 0.04f;
  cfg->edge_threshold = 10.f;
  cfg->max_nb_orientation_per_keypoint = 4;
  cfg->use_vlfeat_format = 0;
  cfg->use_hardware_interpolated_blur = 1;
  cfg->math_mode = 0;
  cfg->pyramid_fp16 = 0;
  cfg->sampler_model = 0;
}

/* sift_memory.c:644-660 */
uint32_t orc_max_nb_octaves(const orc_Config *cfg, uint32_t *rounded_max_image_size)
{
  uint32_t side = (uint32_t)ceilf(sqrtf((float)cfg->input_image_max_size));
  if (rounded_max_image_size)
    *rounded_max_image_size = side * side;
  uint32_t max_oct = (uint32_t)(log2f((float)side) - 4 + (cfg->use_input_upsampling ? 1 : 0));
  if (cfg->nb_octaves > 0 && (uint32_t)cfg->nb_octaves < max_oct)
    max_oct = (uint32_t)cfg->nb_octaves;
  return max_oct;
}

/* sift_memory.c:15-38 */
uint32_t orc_scale_space_info(const orc_Config *cfg, uint32_t w, uint32_t h, uint32_t *ow, uint32_t *oh)
{
  uint32_t lowest = w > h ? h : w;
  uint32_t n = (uint32_t)(log2f((float)lowest) - 4 + (cfg->use_input_upsampling ? 1 : 0));
  uint32_t max_oct = orc_max_nb_octaves(cfg, NULL);
  if (max_oct < n)
    n = max_oct;
  float sf = cfg->use_input_upsampling ? 0.5f : 1.f;
  for (uint32_t o = 0; o < n && o < ORC_MAX_OCTAVES; o++)
  {
    ow[o] = (uint32_t)((1.f / (powf(2.f, (float)o) * sf)) * (float)w);
    oh[o] = (uint32_t)((1.f / (powf(2.f, (float)o) * sf)) * (float)h);
  }
  return n;
}

/* sift_memory.c:61-87 */
void orc_section_caps(uint32_t max_nb_sift, uint32_t nb_octaves, uint32_t *caps)
{
  float mx = (float)max_nb_sift;
  float halves_sum = mx - powf(0.5f, (float)nb_octaves) * mx;
  float corrector = mx / halves_sum;
  for (uint32_t i = 0; i < nb_octaves; i++)
    caps[i] = (uint32_t)floorf((powf(0.5f, (float)(i + 1)) * mx) * corrector);
}

/* sift_detector.c:52-145 */
void orc_gaussian_kernels(const orc_Config *cfg, float *kernels, uint32_t *sizes, float *sigmas)
{
  uint32_t S = (uint32_t)cfg->nb_scales_per_octave;
  for (uint32_t i = 0; i < ORC_MAX_KERNEL * (S + 3); i++)
    kernels[i] = 0.f;
  for (uint32_t scale_i = 0; scale_i < S + 3; scale_i++)
  {
    float sigma;
    if (scale_i == 0)
    {
      float init = cfg->use_input_upsampling ? cfg->input_image_blur_level * 2.f : cfg->input_image_blur_level;
      sigma = sqrtf((cfg->seed_scale_sigma * cfg->seed_scale_sigma) - (init * init));
    }
    else
    {
      float sig_prev = powf(powf(2.f, 1.f / S), (float)(scale_i - 1)) * cfg->seed_scale_sigma;
      float sig_total = sig_prev * powf(2.f, 1.f / S);
      sigma = sqrtf(sig_total * sig_total - sig_prev * sig_prev);
    }
    if (sigmas)
      sigmas[scale_i] = sigma;
    uint32_t ksize = (uint32_t)(int)(ceilf(sigma * 4.f) + 1.f);
    if (ksize > ORC_MAX_KERNEL)
      ksize = ORC_MAX_KERNEL;
    sizes[scale_i] = ksize;

    float tmp[ORC_MAX_KERNEL];
    tmp[0] = 1.f;
    float sum = tmp[0];
    for (uint32_t i = 1; i < ksize; i++)
    {
      /* the reference calls double exp() on float-typed powf results (sift_detector.c:108) */
      tmp[i] = (float)exp(-0.5 * powf((float)i, 2.f) / powf(sigma, 2.f));
      sum += 2 * tmp[i];
    }
    for (uint32_t i = 0; i < ksize; i++)
      tmp[i] /= sum;

    float *k = &kernels[scale_i * ORC_MAX_KERNEL];
    if (cfg->use_hardware_interpolated_blur)
    {
      k[0] = tmp[0];
      k[1] = 0.f;
      for (uint32_t d = 1, ki = 1; (d + 1) < ksize; d += 2, ki++)
      {
        k[ki * 2] = tmp[d] + tmp[d + 1];
        k[ki * 2 + 1] = (((float)d * tmp[d]) + ((float)(d + 1) * tmp[d + 1])) / (tmp[d] + tmp[d + 1]);
      }
    }
    else
    {
      for (uint32_t i = 0; i < ksize; i++)
        k[i] = tmp[i];
    }
  }
}

/* What GaussianBlur.comp:32-44 / GaussianBlurInterpolated.comp:32-44 apply, as direct taps. */
void orc_effective_taps(const orc_Config *cfg, float *taps, uint32_t *ntaps)
{
  uint32_t S = (uint32_t)cfg->nb_scales_per_octave;
  float kernels[ORC_MAX_KERNEL * 16];
  uint32_t sizes[16];
  orc_gaussian_kernels(cfg, kernels, sizes, NULL);
  for (uint32_t s = 0; s < S + 3; s++)
  {
    const float *k = &kernels[s * ORC_MAX_KERNEL];
    float *t = &taps[s * ORC_MAX_KERNEL];
    for (uint32_t i = 0; i < ORC_MAX_KERNEL; i++)
      t[i] = 0.f;
    if (!cfg->use_hardware_interpolated_blur)
    {
      for (uint32_t i = 0; i < sizes[s]; i++)
        t[i] = k[i];
      ntaps[s] = sizes[s];
    }
    else
    {
      /* shader loop: for (i = 2; i < kernel_size; i += 2) sample at +-kernel[i+1] with weight kernel[i].
       * A bilinear fetch at texel offset off = d + f (d integer, 0 <= f < 1) returns
       * (1-f)*texel[d] + f*texel[d+1]. */
      t[0] = k[0];
      uint32_t n = 1;
      for (uint32_t i = 2; i < sizes[s]; i += 2)
      {
        float c = k[i], off = k[i + 1];
        float d = floorf(off);
        float f = off - d;
        uint32_t di = (uint32_t)d;
        t[di] += c * (1.f - f);
        t[di + 1] += c * f;
        if (di + 2 > n)
          n = di + 2;
      }
      ntaps[s] = n;
    }
  }
}

/* ------------------------------------------------------------------------------------------- */
/* pyramid                                                                                     */
/* ------------------------------------------------------------------------------------------- */
struct orc_Pyramid
{
  uint32_t nb_octaves, S;
  uint32_t w[ORC_MAX_OCTAVES], h[ORC_MAX_OCTAVES];
  float *gauss[ORC_MAX_OCTAVES]; /* (S+3) planes of w*h */
  float *dog[ORC_MAX_OCTAVES];   /* (S+2) planes */
  int ups;
};

/* VK_SAMPLER_ADDRESS_MODE_MIRRORED_REPEAT (sift_detector.c:214-216) */
static inline int mirror_idx(int i, int n)
{
  int period = 2 * n;
  int j = i % period;
  if (j < 0)
    j += period;
  return j < n ? j : period - 1 - j;
}

/* One separable blur: H pass src -> tmp, V pass tmp -> dst (sift_detector.c:918-1001).
 * Per pass (GaussianBlur*.comp:32-44): acc = c*k0; acc += (t(+i) + t(-i)) * k[i], i ascending. */
/* fp32 -> IEEE binary16 (round to nearest even, subnormals kept) -> fp32: what storing a texel in an R16_SFLOAT image and
 * loading it back does. Plain integer code: no dependence on F16C or on the compiler's _Float16 support. */
static float round_trip_f16(float x)
{
  uint32_t f;
  memcpy(&f, &x, 4);
  const uint32_t sign = f & 0x80000000u, e = (f >> 23) & 0xffu;
  uint32_t m = f & 0x7fffffu, h;
  if (e == 255u)
    return x; /* inf / nan */
  const int E = (int)e - 127 + 15;
  if (E >= 31)
    h = 0x7c00u; /* overflow -> inf */
  else if (E <= 0)
  {
    if (E < -10)
      h = 0; /* below half of the smallest subnormal */
    else
    {
      m |= 0x800000u;
      const uint32_t shift = (uint32_t)(14 - E), rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1u);
      h = m >> shift;
      if (rem > half || (rem == half && (h & 1u)))
        h++;
    }
  }
  else
  {
    h = ((uint32_t)E << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u)))
      h++; /* a mantissa carry moves into the exponent field: still the right encoding */
  }
  /* widen */
  const uint32_t he = (h >> 10) & 0x1fu, hm = h & 0x3ffu;
  uint32_t o;
  if (he == 0)
  {
    if (hm == 0)
      o = sign;
    else
    {
      int sh = 0;
      uint32_t mm = hm;
      while (!(mm & 0x400u))
        mm <<= 1, sh++;
      o = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((mm & 0x3ffu) << 13);
    }
  }
  else if (he == 31u)
    o = sign | 0x7f800000u | (hm << 13);
  else
    o = sign | ((he + 127u - 15u) << 23) | (hm << 13);
  float r;
  memcpy(&r, &o, 4);
  return r;
}

static void store_as_f16(float *plane, size_t n)
{
  for (size_t i = 0; i < n; i++)
    plane[i] = round_trip_f16(plane[i]);
}

static void blur_plane(const float *src, float *dst, float *tmp, int w, int h, const float *taps, int n, int fp16)
{
  for (int y = 0; y < h; y++)
  {
    const float *row = src + (size_t)y * w;
    float *out = tmp + (size_t)y * w;
    for (int x = 0; x < w; x++)
    {
      float acc = row[x] * taps[0];
      for (int i = 1; i < n; i++)
        acc = fmaf(row[mirror_idx(x + i, w)] + row[mirror_idx(x - i, w)], taps[i], acc);
      out[x] = fp16 ? round_trip_f16(acc) : acc; /* the horizontal pass writes the octave's temporary image: R16 in the fp16 mode */
    }
  }
  for (int y = 0; y < h; y++)
  {
    float *out = dst + (size_t)y * w;
    for (int x = 0; x < w; x++)
    {
      float acc = tmp[(size_t)y * w + x] * taps[0];
      for (int i = 1; i < n; i++)
        acc = fmaf(tmp[(size_t)mirror_idx(y + i, h) * w + x] + tmp[(size_t)mirror_idx(y - i, h) * w + x], taps[i], acc);
      out[x] = fp16 ? round_trip_f16(acc) : acc;
    }
  }
}

/* The same blur as a texture unit would run it (orc_Config.sampler_model): k is the reference's interpolated kernel — k[0] the
 * centre weight, then (weight, offset) pairs (sift_detector.c:119-135). A fetch at texel-centre distance `off` = d + f interpolates
 * texels d and d + 1 (on the minus side: -d and -d - 1) with weight alpha = round(f * 256) / 256 on the farther one, as
 * t0 + alpha * (t1 - t0); the two fetches are added, then multiplied by the pair's weight (GaussianBlurInterpolated.comp:34-37). */
static void blur_plane_sampler(const float *src, float *dst, float *tmp, int w, int h, const float *k, int ksize)
{
  for (int pass = 0; pass < 2; pass++)
  {
    const float *in = pass == 0 ? src : tmp;
    float *out = pass == 0 ? tmp : dst;
    const int n = pass == 0 ? w : h;
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++)
      {
        const int c = pass == 0 ? x : y;
        float acc = in[(size_t)y * w + x] * k[0];
        for (int i = 2; i < ksize; i += 2)
        {
          const float off = k[i + 1];
          const int d = (int)floorf(off);
          const float alpha = floorf((off - (float)d) * 256.f + 0.5f) / 256.f;
          float t[4];
          const int idx[4] = {c + d, c + d + 1, c - d, c - d - 1};
          for (int q = 0; q < 4; q++)
          {
            const int m = mirror_idx(idx[q], n);
            t[q] = pass == 0 ? in[(size_t)y * w + m] : in[(size_t)m * w + x];
          }
          const float sp = t[0] + alpha * (t[1] - t[0]), sm = t[2] + alpha * (t[3] - t[2]);
          acc += (sp + sm) * k[i];
        }
        out[(size_t)y * w + x] = acc;
      }
  }
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* vkCmdBlitImage R8_UNORM -> R32F, VK_FILTER_LINEAR (sift_detector.c:909-916). */
static void blit_input(const uint8_t *img, int sw, int sh, float *dst, int dw, int dh)
{
  if (dw == sw && dh == sh)
  {
    for (size_t i = 0; i < (size_t)sw * sh; i++)
      dst[i] = (float)img[i] / 255.f;
    return;
  }
  float sx = (float)sw / (float)dw, sy = (float)sh / (float)dh;
  for (int y = 0; y < dh; y++)
  {
    float v = ((float)y + 0.5f) * sy - 0.5f;
    float fy = floorf(v);
    float b = v - fy;
    int y0 = clampi((int)fy, 0, sh - 1), y1 = clampi((int)fy + 1, 0, sh - 1);
    for (int x = 0; x < dw; x++)
    {
      float u = ((float)x + 0.5f) * sx - 0.5f;
      float fx = floorf(u);
      float a = u - fx;
      int x0 = clampi((int)fx, 0, sw - 1), x1 = clampi((int)fx + 1, 0, sw - 1);
      float t00 = (float)img[(size_t)y0 * sw + x0] / 255.f, t10 = (float)img[(size_t)y0 * sw + x1] / 255.f;
      float t01 = (float)img[(size_t)y1 * sw + x0] / 255.f, t11 = (float)img[(size_t)y1 * sw + x1] / 255.f;
      float r0 = fmaf(a, t10, (1.f - a) * t00);
      float r1 = fmaf(a, t11, (1.f - a) * t01);
      dst[(size_t)y * dw + x] = fmaf(b, r1, (1.f - b) * r0);
    }
  }
}

/* vkCmdBlitImage VK_FILTER_NEAREST, octave o layer S -> octave o+1 layer 0 (sift_detector.c:1003-1034). */
static void blit_nearest(const float *src, int sw, int sh, float *dst, int dw, int dh)
{
  float sx = (float)sw / (float)dw, sy = (float)sh / (float)dh;
  for (int y = 0; y < dh; y++)
  {
    int yy = clampi((int)floorf(((float)y + 0.5f) * sy), 0, sh - 1);
    for (int x = 0; x < dw; x++)
    {
      int xx = clampi((int)floorf(((float)x + 0.5f) * sx), 0, sw - 1);
      dst[(size_t)y * dw + x] = src[(size_t)yy * sw + xx];
    }
  }
}

orc_Pyramid *orc_pyramid_build(const orc_Config *cfg, const uint8_t *img, uint32_t w, uint32_t h)
{
  orc_Pyramid *p = (orc_Pyramid *)calloc(1, sizeof(orc_Pyramid));
  uint32_t S = (uint32_t)cfg->nb_scales_per_octave;
  p->S = S;
  p->ups = cfg->use_input_upsampling;
  p->nb_octaves = orc_scale_space_info(cfg, w, h, p->w, p->h);
  float taps[ORC_MAX_KERNEL * 16];
  uint32_t ntaps[16];
  orc_effective_taps(cfg, taps, ntaps);
  float kern[ORC_MAX_KERNEL * 16];
  uint32_t ksz[16];
  orc_gaussian_kernels(cfg, kern, ksz, NULL);
  const int sampler = cfg->sampler_model && cfg->use_hardware_interpolated_blur && !cfg->pyramid_fp16;

  float *tmp = (float *)malloc(sizeof(float) * (size_t)p->w[0] * p->h[0]);
  for (uint32_t o = 0; o < p->nb_octaves; o++)
  {
    size_t px = (size_t)p->w[o] * p->h[o];
    p->gauss[o] = (float *)malloc(sizeof(float) * px * (S + 3));
    p->dog[o] = (float *)malloc(sizeof(float) * px * (S + 2));
    float *g = p->gauss[o];
    if (o == 0)
    {
      blit_input(img, (int)w, (int)h, g, (int)p->w[0], (int)p->h[0]);
      /* fp16 mode: every image the reference allocates in the pyramid format holds binary16 texels (sift_memory.c:139,185-186:
       * the octave images AND the blur temporaries): the blit target, the horizontal pass's output, every layer, every DoG layer */
      if (cfg->pyramid_fp16)
        store_as_f16(g, px);
      /* seed blur in place on layer 0: H layer0 -> tmp, V tmp -> layer0 (sift_detector.c:927-952) */
      if (sampler)
        blur_plane_sampler(g, g, tmp, (int)p->w[0], (int)p->h[0], &kern[0], (int)ksz[0]);
      else
        blur_plane(g, g, tmp, (int)p->w[0], (int)p->h[0], &taps[0], (int)ntaps[0], cfg->pyramid_fp16);
    }
    else
    {
      blit_nearest(p->gauss[o - 1] + (size_t)S * p->w[o - 1] * p->h[o - 1], (int)p->w[o - 1], (int)p->h[o - 1], g, (int)p->w[o], (int)p->h[o]);
    }
    for (uint32_t s = 1; s < S + 3; s++)
    {
      if (sampler)
        blur_plane_sampler(g + (s - 1) * px, g + s * px, tmp, (int)p->w[o], (int)p->h[o], &kern[s * ORC_MAX_KERNEL], (int)ksz[s]);
      else
        blur_plane(g + (s - 1) * px, g + s * px, tmp, (int)p->w[o], (int)p->h[o], &taps[s * ORC_MAX_KERNEL], (int)ntaps[s], cfg->pyramid_fp16);
    }
    /* DifferenceOfGaussian.comp:13-17 */
    for (uint32_t s = 0; s < S + 2; s++)
      for (size_t i = 0; i < px; i++)
        p->dog[o][s * px + i] = g[(s + 1) * px + i] - g[s * px + i];
    if (cfg->pyramid_fp16)
      store_as_f16(p->dog[o], px * (S + 2));
  }
  free(tmp);
  return p;
}

void orc_pyramid_free(orc_Pyramid *p)
{
  if (!p)
    return;
  for (uint32_t o = 0; o < p->nb_octaves; o++)
  {
    free(p->gauss[o]);
    free(p->dog[o]);
  }
  free(p);
}
uint32_t orc_pyramid_nb_octaves(const orc_Pyramid *p) { return p->nb_octaves; }
void orc_pyramid_resolution(const orc_Pyramid *p, uint32_t o, uint32_t *w, uint32_t *h)
{
  *w = p->w[o];
  *h = p->h[o];
}
const float *orc_pyramid_gauss(const orc_Pyramid *p, uint32_t o, uint32_t s) { return p->gauss[o] + (size_t)s * p->w[o] * p->h[o]; }
const float *orc_pyramid_dog(const orc_Pyramid *p, uint32_t o, uint32_t s) { return p->dog[o] + (size_t)s * p->w[o] * p->h[o]; }

/* ------------------------------------------------------------------------------------------- */
/* K4: ExtractKeypoints.comp                                                                   */
/* ------------------------------------------------------------------------------------------- */
typedef struct
{
  const float *base;
  int w, h, layers;
} Img;

/* imageLoad with robust out-of-bounds behaviour (returns 0), quirk Q1/Q2 */
static inline float ld(const Img *im, int s, int x, int y)
{
  if (s < 0 || s >= im->layers || x < 0 || x >= im->w || y < 0 || y >= im->h)
    return 0.f;
  return im->base[((size_t)s * im->h + y) * im->w + x];
}

/* ExtractKeypoints.comp:46-229; returns 1 and fills *kp when the texel yields a keypoint. */
static int extract_one(const orc_Config *cfg, const MathOps *m, const Img *dog, int S, int octave_idx, int x, int y, int s, orc_Feature *kp)
{
  const float dog_threshold = cfg->intensity_threshold / (float)S; /* sift_detector.c:1136 */
  const float edge_threshold = cfg->edge_threshold;
  const int W = dog->w, H = dog->h;

  float c = ld(dog, s, x, y);
  if (!(fabsf(c) > dog_threshold * 0.8f))
    return 0;
  int is_max = 1, is_min = 1;
  for (int ds = -1; ds <= 1; ds++)
    for (int dy = -1; dy <= 1; dy++)
      for (int dx = -1; dx <= 1; dx++)
      {
        if (!ds && !dy && !dx)
          continue;
        float v = ld(dog, s + ds, x + dx, y + dy);
        if (!(c > v))
          is_max = 0;
        if (!(c < v))
          is_min = 0;
      }
  if (!is_max && !is_min)
    return 0;

  /* refinement, :121-188 */
  float oX = 0.f, oY = 0.f, oS = 0.f, gX = 0.f, gY = 0.f, gS = 0.f;
  int rx = x, ry = y, rs = s;
  for (int step = 0; step < 5; step++)
  {
    float vc = ld(dog, rs, rx, ry);
    gS = 0.5f * (ld(dog, rs + 1, rx, ry) - ld(dog, rs - 1, rx, ry));
    gX = 0.5f * (ld(dog, rs, rx + 1, ry) - ld(dog, rs, rx - 1, ry));
    gY = 0.5f * (ld(dog, rs, rx, ry + 1) - ld(dog, rs, rx, ry - 1));
    float h11 = ld(dog, rs + 1, rx, ry) + ld(dog, rs - 1, rx, ry) - 2.f * vc;
    float h22 = ld(dog, rs, rx + 1, ry) + ld(dog, rs, rx - 1, ry) - 2.f * vc;
    float h33 = ld(dog, rs, rx, ry + 1) + ld(dog, rs, rx, ry - 1) - 2.f * vc;
    float h12 = 0.25f * (ld(dog, rs + 1, rx + 1, ry) - ld(dog, rs + 1, rx - 1, ry) - ld(dog, rs - 1, rx + 1, ry) + ld(dog, rs - 1, rx - 1, ry));
    float h13 = 0.25f * (ld(dog, rs + 1, rx, ry + 1) - ld(dog, rs + 1, rx, ry - 1) - ld(dog, rs - 1, rx, ry + 1) + ld(dog, rs - 1, rx, ry - 1));
    float h23 = 0.25f * (ld(dog, rs, rx + 1, ry + 1) - ld(dog, rs, rx + 1, ry - 1) - ld(dog, rs, rx - 1, ry + 1) + ld(dog, rs, rx - 1, ry - 1));

    float det = h11 * ((h22 * h33) - (h23 * h23)) - h12 * ((h12 * h33) - (h13 * h23)) + h13 * ((h12 * h23) - (h13 * h22));
    if (det == 0.0f)
      return 0;
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
  /* acceptance tests, :189-206 */
  float sx = (float)rx + oX, sy = (float)ry + oY, ss = (float)rs + oS;
  float vc = ld(dog, rs, rx, ry);
  float nv = vc + 0.5f * (gX * oX + gY * oY + gS * oS);
  if (!(fabsf(nv) > dog_threshold && fabsf(oX) < 1.5f && fabsf(oY) < 1.5f && fabsf(oS) < 1.5f && sx >= 0 && sx < (float)W && sy >= 0 && sy < (float)H &&
        ss >= 0 && ss <= (float)(S + 1)))
    return 0;
  float e11 = ld(dog, rs, rx + 1, ry) + ld(dog, rs, rx - 1, ry) - 2.f * vc;
  float e22 = ld(dog, rs, rx, ry + 1) + ld(dog, rs, rx, ry - 1) - 2.f * vc;
  float e12 = 0.25f * (ld(dog, rs, rx + 1, ry + 1) - ld(dog, rs, rx + 1, ry - 1) - ld(dog, rs, rx - 1, ry + 1) + ld(dog, rs, rx - 1, ry - 1));
  float edgeness = ((e11 + e22) * (e11 + e22)) / ((e11 * e22) - (e12 * e12));
  float edge_limit = ((edge_threshold + 1.f) * (edge_threshold + 1.f)) / edge_threshold; /* pow(e+1,2)/e */
  if (!((edgeness < edge_limit) && (edgeness >= 0)))
    return 0;

  /* emission, :208-224 */
  float scale_factor = ldexpf(1.f, octave_idx); /* pow(2, octave_idx), exact */
  memset(kp, 0, sizeof(*kp));
  kp->scale_x = sx;
  kp->scale_y = sy;
  kp->scale_idx = (uint32_t)roundf(ss);
  kp->octave_idx = octave_idx;
  kp->sigma = cfg->seed_scale_sigma * m->exp2_(ss / (float)S) * scale_factor;
  kp->orientation = 0.f;
  kp->intensity = nv;
  kp->x = sx * scale_factor;
  kp->y = sy * scale_factor;
  return 1;
}

uint32_t orc_extract_keypoints(const orc_Config *cfg, const orc_Pyramid *p, uint32_t o, orc_Feature *out, uint32_t cap)
{
  MathOps m = get_math(cfg);
  int S = (int)p->S;
  Img dog = {p->dog[o], (int)p->w[o], (int)p->h[o], S + 2};
  int octave_idx = (int)o - (p->ups ? 1 : 0); /* sift_detector.c:1134 */
  uint32_t n = 0;
  orc_Feature kp;
  for (int s = 1; s <= S; s++)
    for (int y = 1; y < dog.h - 1; y++)
      for (int x = 1; x < dog.w - 1; x++)
        if (extract_one(cfg, &m, &dog, S, octave_idx, x, y, s, &kp))
        {
          if (n < cap)
            out[n] = kp;
          n++;
        }
  return n; /* un-clamped, like nb_elem */
}

/* ------------------------------------------------------------------------------------------- */
/* K5: ComputeOrientation.comp                                                                 */
/* ------------------------------------------------------------------------------------------- */
uint32_t orc_orientations(const orc_Config *cfg, const orc_Pyramid *p, uint32_t o, const orc_Feature *kp, float *angles, uint32_t *hist_out)
{
  MathOps m = get_math(cfg);
  Img im = {p->gauss[o], (int)p->w[o], (int)p->h[o], (int)p->S + 3};
  const int W = im.w, H = im.h;
  uint32_t hist[36], tmp[36];
  memset(hist, 0, sizeof(hist));

  float scale_factor = ldexpf(1.f, kp->octave_idx);
  float lambda = 1.5f * (kp->sigma / scale_factor);
  int r = (int)floorf(3 * lambda);
  float es = -1.f / (2.f * lambda * lambda);

  /* fixed-point scale, :73-81 */
  float max_elem_val = 0.f;
  if (!m.det)
  {
    for (int i = -r; i <= r; i++)
      for (int j = -r; j <= r; j++)
        max_elem_val += m.exp_(es * (float)((i * i) + (j * j))) * sqrtf(2.f);
  }
  else
  {
    /* det mode: same quantity through the separable identity sum_ij e^{es(i^2+j^2)} = (sum_i e^{es i^2})^2,
     * O(r) instead of O(r^2); only ceil(log2()) of it is used. The HIP kernel uses this exact order. */
    float g = 1.f;
    for (int i = 1; i <= r; i++)
      g += 2.f * m.exp_(es * (float)(i * i));
    max_elem_val = (g * g) * sqrtf(2.f);
  }
  float fp = (float)(1u << (uint32_t)(30 - m.ceil_log2_(max_elem_val)));

  float rsx = roundf(kp->scale_x), rsy = roundf(kp->scale_y);
  int box = 2 * r + 1;
  for (int pix = 0; pix < box * box; pix++)
  {
    int dy = (pix / box) - r, dx = (pix % box) - r;
    int gx = (int)rsx + dx, gy = (int)rsy + dy;
    float sdx = (rsx + (float)dx) - kp->scale_x;
    float sdy = (rsy + (float)dy) - kp->scale_y;
    float d2 = (sdx * sdx) + (sdy * sdy);
    /* quirk Q2: '&&' — a pixel is skipped only if it is outside the interior AND outside the circle */
    if ((gx < 1 || gx >= (W - 1) || gy < 1 || gy >= (H - 1)) && (d2 > (float)(r * r)))
      continue;
    int L = (int)kp->scale_idx;
    float gradX = 0.5f * (ld(&im, L, gx + 1, gy) - ld(&im, L, gx - 1, gy));
    float gradY = 0.5f * (ld(&im, L, gx, gy + 1) - ld(&im, L, gx, gy - 1));
    float mag = m.exp_(d2 * es) * sqrtf((gradX * gradX) + (gradY * gradY));
    float ori = m.atan2_(gradY, gradX);
    if (ori < 0)
      ori += 2.f * PI_F;
    else if (ori > (2.f * PI_F))
      ori -= 2.f * PI_F;
    int bin = (int)((ori * 36.f / (2.f * PI_F)));
    if (bin < 0)
      bin += 36;
    else if (bin >= 36)
      bin -= 36;
    hist[bin] += (uint32_t)(mag * fp);
  }

  /* smoothing, :130-147 */
  for (int it = 0; it < 3; it++)
  {
    for (int i = 0; i < 36; i++)
      tmp[i] = (uint32_t)((float)(hist[(i + 35) % 36] + hist[i] + hist[(i + 1) % 36]) / 3.f);
    for (int i = 0; i < 36; i++)
      hist[i] = (uint32_t)((float)(tmp[(i + 35) % 36] + tmp[i] + tmp[(i + 1) % 36]) / 3.f);
  }
  if (hist_out)
    memcpy(hist_out, hist, sizeof(hist));
  uint32_t mx = 0;
  for (int i = 0; i < 36; i++)
    if (hist[i] > mx)
      mx = hist[i];

  /* peaks, :156-183; bins visited in ascending order (arrival order is unspecified in the reference) */
  uint32_t n = 0;
  for (int i = 0; i < 36; i++)
  {
    int pi_ = (i + 35) % 36, ni = (i + 1) % 36;
    if (((float)hist[i] >= (0.8f * (float)mx)) && (hist[i] > hist[pi_]) && (hist[i] > hist[ni]))
    {
      /* quirk Q3: uint32 wrap-around arithmetic before the float conversion */
      uint32_t num = hist[pi_] - hist[ni];
      uint32_t den = hist[pi_] - (2u * hist[i]) + hist[ni];
      float idx = (float)i + 0.5f * ((float)num / (float)den);
      angles[n++] = (idx + 0.5f) * (2.f * PI_F) / 36.f;
    }
  }
  return n;
}

/* ------------------------------------------------------------------------------------------- */
/* K6: ComputeDescriptors.comp                                                                 */
/* ------------------------------------------------------------------------------------------- */
static inline int smod8(int v) { return ((v % 8) + 8) % 8; } /* OpSMod: sign follows the divisor (quirk Q5) */

void orc_descriptor(const orc_Config *cfg, const orc_Pyramid *p, uint32_t o, orc_Feature *kp, uint32_t *raw_out)
{
  MathOps m = get_math(cfg);
  Img im = {p->gauss[o], (int)p->w[o], (int)p->h[o], (int)p->S + 3};
  const int W = im.w, H = im.h;
  uint32_t work[128];
  memset(work, 0, sizeof(work));

  float scale_factor = ldexpf(1.f, kp->octave_idx);
  float lambda = 3.0f * (kp->sigma / scale_factor);
  float radius = sqrtf(2.f) * lambda * 5.f * 0.5f; /* sqrt(2)*lambda*(NB_HIST+1)*0.5, left to right */
  int R = (int)floorf(radius + 0.5f);
  float sn, cs;
  m.sincos_(kp->orientation, &sn, &cs);
  float kcos = cs / lambda, ksin = sn / lambda;
  const float es = -1.f / (2.f * 2 * 2);

  /* fixed-point scale, :116-124 (depends on R/2 only) */
  float max_elem_val = 0.f;
  for (int i = 0; i < R / 2; i++)
  {
    max_elem_val += m.exp_(es * (float)((i * i) + (i * i))) * sqrtf(2.f);
    for (int j = i + 1; j < R / 2; j++)
      max_elem_val += m.exp_(es * (float)((i * i) + (j * j))) * sqrtf(2.f) * 2;
  }
  float fp = (float)(1u << (uint32_t)(16 - m.ceil_log2_(max_elem_val)));

  float rsx = roundf(kp->scale_x), rsy = roundf(kp->scale_y);
  int box = 2 * R + 1;
  for (int pix = 0; pix < box * box; pix++)
  {
    int dy = (pix / box) - R, dx = (pix % box) - R;
    int ix = (int)rsx + dx, iy = (int)rsy + dy;
    float sdx = (rsx + (float)dx) - kp->scale_x;
    float sdy = (rsy + (float)dy) - kp->scale_y;
    if (ix < 1 || ix >= (W - 1) || iy < 1 || iy >= (H - 1))
      continue;
    float ox = kcos * sdx + ksin * sdy;
    float oy = kcos * sdy - ksin * sdx;
    int L = (int)kp->scale_idx;
    float gradX = 0.5f * (ld(&im, L, ix + 1, iy) - ld(&im, L, ix - 1, iy));
    float gradY = 0.5f * (ld(&im, L, ix, iy + 1) - ld(&im, L, ix, iy - 1));
    float ori = m.atan2_(gradY, gradX);
    if (ori < 0)
      ori += 2.f * PI_F;
    else if (ori > (2.f * PI_F))
      ori -= 2.f * PI_F;
    ori = ori - kp->orientation;
    if (ori < 0)
      ori += 2.f * PI_F;
    else if (ori > (2.f * PI_F))
      ori -= 2.f * PI_F;
    float mag = m.exp_(es * ((ox * ox) + (oy * oy))) * sqrtf((gradX * gradX) + (gradY * gradY));

    float fhx = ox + 2.f, fhy = oy + 2.f;
    float fbin = cfg->use_vlfeat_format ? (ori * 8.f / (2.f * PI_F)) : (-ori * 8.f / (2.f * PI_F));
    int hx = (int)floorf(fhx - 0.5f), hy = (int)floorf(fhy - 0.5f), hb = (int)floorf(fbin);
    float rhx = fhx - ((float)hx + 0.5f), rhy = fhy - ((float)hy + 0.5f), rb = fbin - (float)hb;
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++)
        for (int k = 0; k < 2; k++)
          if ((i + hx) >= 0 && (i + hx) < 4 && (j + hy) >= 0 && (j + hy) < 4)
          {
            int idx = (j + hy) * 32 + (i + hx) * 8 + smod8(k + hb);
            float val = fabsf(1.f - (float)i - rhx) * fabsf(1.f - (float)j - rhy) * fabsf(1.f - (float)k - rb) * mag;
            work[idx] += (uint32_t)(val * fp);
          }
  }
  if (raw_out)
    memcpy(raw_out, work, sizeof(work));

  /* post-processing, :200-265 */
  uint32_t acc = 0;
  for (int i = 0; i < 128; i++)
    acc += work[i] * work[i];
  float norm = sqrtf((float)acc);
  uint32_t lim = (uint32_t)(norm * 0.2f);
  for (int i = 0; i < 128; i++)
    if (work[i] > lim)
      work[i] = lim;
  acc = 0;
  for (int i = 0; i < 128; i++)
    acc += work[i] * work[i];
  norm = sqrtf((float)acc);
  for (int i = 0; i < 128; i++)
  {
    float v = (float)work[i] * (512.f / norm);
    uint8_t b;
    if (v != v)
      b = 0; /* norm == 0 -> 0*inf = NaN; uint(NaN) is undefined in GLSL, v_cvt_u32_f32 gives 0 */
    else if (v < 0.f)
      b = 0;
    else if (v > 255.f)
      b = 255;
    else
      b = (uint8_t)(uint32_t)v;
    kp->descriptor[i] = b;
  }
}

/* ------------------------------------------------------------------------------------------- */
/* full detection                                                                              */
/* ------------------------------------------------------------------------------------------- */
uint32_t orc_detect_from_pyramid(const orc_Config *cfg, const orc_Pyramid *p, orc_Feature *out, uint32_t out_cap, uint32_t *counts_found)
{
  uint32_t caps[ORC_MAX_OCTAVES];
  orc_section_caps(cfg->max_nb_sift_per_buffer, p->nb_octaves, caps);
  uint32_t written = 0;
  for (uint32_t o = 0; o < p->nb_octaves; o++)
  {
    uint32_t cap = caps[o];
    orc_Feature *sec = (orc_Feature *)malloc(sizeof(orc_Feature) * (cap ? cap : 1));
    uint32_t found = orc_extract_keypoints(cfg, p, o, sec, cap);
    uint32_t n0 = found < cap ? found : cap; /* orientation dispatch count = features actually stored */
    uint32_t total = found;
    uint32_t stored = n0;
    for (uint32_t k = 0; k < n0; k++)
    {
      float angles[36];
      uint32_t na = orc_orientations(cfg, p, o, &sec[k], angles, NULL);
      for (uint32_t j = 0; j < na; j++)
      {
        if (j == 0)
          sec[k].orientation = angles[0];
        else if (cfg->max_nb_orientation_per_keypoint == 0 || j < cfg->max_nb_orientation_per_keypoint)
        {
          uint32_t idx = total++;
          if (idx < cap)
          {
            sec[idx] = sec[k];
            sec[idx].orientation = angles[j];
            stored = idx + 1;
          }
        }
      }
    }
    for (uint32_t k = 0; k < stored; k++)
      orc_descriptor(cfg, p, o, &sec[k], NULL);
    if (counts_found)
      counts_found[o] = total;
    for (uint32_t k = 0; k < stored && written < out_cap; k++)
      out[written++] = sec[k];
    free(sec);
  }
  return written;
}

uint32_t orc_detect(const orc_Config *cfg, const uint8_t *img, uint32_t w, uint32_t h, orc_Feature *out, uint32_t out_cap, uint32_t *counts_found)
{
  orc_Pyramid *p = orc_pyramid_build(cfg, img, w, h);
  uint32_t n = orc_detect_from_pyramid(cfg, p, out, out_cap, counts_found);
  orc_pyramid_free(p);
  return n;
}

/* ------------------------------------------------------------------------------------------- */
/* K7: Get2NearestNeighbors.comp:43-103                                                        */
/* ------------------------------------------------------------------------------------------- */
static float desc_dist(const uint8_t *a, const uint8_t *b)
{
  float dist = 0.f;
  for (int i = 0; i < 128; i++)
  {
    uint32_t ae = a[i], be = b[i];
    dist += (float)((ae - be) * (ae - be)); /* uint32 wrap then square: exact (a-b)^2 */
  }
  return sqrtf(dist);
}

static void match_rows(const uint8_t *a, size_t sa, uint32_t na, const uint8_t *b, size_t sb, uint32_t nb, orc_Match *out)
{
  for (uint32_t ai = 0; ai < na; ai++)
  {
    const uint8_t *da = a + (size_t)ai * sa;
    /* quirk Q6: b[0] and b[1] are read unconditionally by the shader; callers guarantee nb >= 2 */
    float d0 = desc_dist(da, b), d1 = desc_dist(da, b + sb);
    float best_d, second_d;
    uint32_t best_i, second_i;
    if (d0 < d1)
      best_d = d0, best_i = 0, second_d = d1, second_i = 1;
    else /* quirk Q7: a tie makes index 1 the best */
      best_d = d1, best_i = 1, second_d = d0, second_i = 0;
    for (uint32_t bi = 2; bi < nb; bi++)
    {
      float d = desc_dist(da, b + (size_t)bi * sb);
      if (d < best_d)
      {
        second_d = best_d, second_i = best_i;
        best_d = d, best_i = bi;
      }
      else if (d < second_d)
        second_d = d, second_i = bi;
    }
    out[ai].idx_a = ai;
    out[ai].idx_b1 = best_i;
    out[ai].idx_b2 = second_i;
    out[ai].dist_a_b1 = best_d;
    out[ai].dist_a_b2 = second_d;
  }
}

void orc_match_2nn(const orc_Feature *a, uint32_t na, const orc_Feature *b, uint32_t nb, orc_Match *out)
{
  match_rows(a->descriptor, sizeof(orc_Feature), na, b->descriptor, sizeof(orc_Feature), nb, out);
}
void orc_match_2nn_desc(const uint8_t *a, uint32_t na, const uint8_t *b, uint32_t nb, orc_Match *out) { match_rows(a, 128, na, b, 128, nb, out); }

/* Match filtering as every caller of the reference does it on the CPU after vksift_downloadMatches:
 * src/examples/test_sift_match.cpp:90-107 and src/perf/perf_common.cpp:123-169 (cross-check: the nearest neighbour of
 * a's nearest neighbour must be a; Lowe ratio d1/d2 < ratio in both directions), perf_common.cpp:151-163 without
 * cross-check (ratio test of the forward match only). Output pairs in increasing idx_a order; returns their number. */
uint32_t orc_filter_matches(const orc_Match *m12, uint32_t n12, const orc_Match *m21, uint32_t n21, float ratio, int cross_check, uint32_t *out_a,
                            uint32_t *out_b)
{
  uint32_t n = 0;
  for (uint32_t i = 0; i < n12; i++)
  {
    const uint32_t j = m12[i].idx_b1;
    if (cross_check)
    {
      if (j >= n21 || m21[j].idx_b1 != i)
        continue;
    }
    if (!((m12[i].dist_a_b1 / m12[i].dist_a_b2) < ratio))
      continue;
    if (cross_check && !((m21[j].dist_a_b1 / m21[j].dist_a_b2) < ratio))
      continue;
    out_a[n] = m12[i].idx_a;
    out_b[n] = j;
    n++;
  }
  return n;
}

/* ------------------------------------------------------------------------------------------- */
/* detmath.h entry points exported for tests/test_detmath.py                                   */
/* ------------------------------------------------------------------------------------------- */
float orc_dm_expf(float x) { return dm_expf(x); }
float orc_dm_exp2f(float x) { return dm_exp2f(x); }
float orc_dm_atan2f(float y, float x) { return dm_atan2f(y, x); }
float orc_dm_div_2pi(float x) { return dm_div_2pi(x); }
/* dm_div_3 against x / 3.f over every integer-valued float in [0, 2^32]: the number of results that differ in any bit (0 expected) */
uint32_t orc_check_div_3(void)
{
  uint32_t bad = 0;
  for (uint32_t i = 0; i < (1u << 23); i++)
  {
    const float x = (float)i, a = dm_div_3(x), b = x / 3.f;
    bad += memcmp(&a, &b, 4) != 0;
  }
  for (uint32_t u = dm_f2u(0x1p23f); u <= dm_f2u(0x1p32f); u++) /* from 2^23 on every float is an integer */
  {
    const float x = dm_u2f(u), a = dm_div_3(x), b = x / 3.f;
    bad += memcmp(&a, &b, 4) != 0;
  }
  return bad;
}
float orc_dm_expf_nb(float x) { return dm_expf_nb(x); }
float orc_dm_expf_nb_nonpos(float x) { return dm_expf_nb_nonpos(x); } /* kernel-side helper, exported for tests/test_detmath.py only */ /* kernel-side helper, exported for tests/test_detmath.py only */
float orc_dm_sinf(float t)
{
  float s, c;
  dm_sincosf(t, &s, &c);
  return s;
}
float orc_dm_cosf(float t)
{
  float s, c;
  dm_sincosf(t, &s, &c);
  return c;
}
int orc_dm_ceil_log2f(float m) { return dm_ceil_log2f(m); }
