/* vksift_hostmath.c — see vksift_hostmath.h. Compiled with -ffp-contract=off. */
#include "vksift_hostmath.h"

#include <math.h>
#include <string.h>

#include "../detmath.h"

uint32_t vksift_hm_max_octaves(const vksift_Config *cfg, uint32_t *rounded_max_image_size)
{
  /* The reference sizes everything for a square image of the configured area (sift_memory.c:644-647)
   * and derives the octave limit from its side (:655). */
  uint32_t side = (uint32_t)ceilf(sqrtf((float)cfg->input_image_max_size));
  if (rounded_max_image_size)
    *rounded_max_image_size = side * side;
  uint32_t lim = (uint32_t)(log2f((float)side) - 4 + (cfg->use_input_upsampling ? 1 : 0));
  if (cfg->nb_octaves > 0 && cfg->nb_octaves < lim)
    lim = cfg->nb_octaves;
  if (lim > VKSIFT_MAX_OCTAVES)
    lim = VKSIFT_MAX_OCTAVES;
  return lim;
}

uint32_t vksift_hm_octaves_for(const vksift_Config *cfg, uint32_t max_octaves, uint32_t w, uint32_t h, uint32_t *ow, uint32_t *oh)
{
  uint32_t shortest = w < h ? w : h;
  /* sift_memory.c:22; an image too small for a single octave has none (the float -> unsigned cast of a negative value is undefined) */
  const float fn = log2f((float)shortest) - 4 + (cfg->use_input_upsampling ? 1 : 0);
  uint32_t n = fn >= 1.f ? (uint32_t)fn : 0u;
  if (n > max_octaves)
    n = max_octaves;
  const float first_octave_scale = cfg->use_input_upsampling ? 0.5f : 1.f;
  for (uint32_t o = 0; o < n; o++)
  {
    float inv = 1.f / (powf(2.f, (float)o) * first_octave_scale);
    ow[o] = (uint32_t)(inv * (float)w);
    oh[o] = (uint32_t)(inv * (float)h);
  }
  return n;
}

void vksift_hm_section_caps(uint32_t max_nb_sift, uint32_t nb_octaves, uint32_t *caps)
{
  /* section o gets the o-th "half" of the buffer, rescaled so the halves fill it (sift_memory.c:47-58) */
  const float total = (float)max_nb_sift;
  const float covered = total - powf(0.5f, (float)nb_octaves) * total;
  const float rescale = total / covered;
  for (uint32_t o = 0; o < nb_octaves; o++)
    caps[o] = (uint32_t)floorf((powf(0.5f, (float)(o + 1)) * total) * rescale);
}

void vksift_hm_blur_taps(const vksift_Config *cfg, float *taps, uint32_t *ntaps)
{
  const uint32_t S = cfg->nb_scales_per_octave;
  const float k = powf(2.f, 1.f / S);
  for (uint32_t s = 0; s < S + 3; s++)
  {
    /* blur increment that takes scale s-1 to scale s (sift_detector.c:76-89) */
    float sigma;
    if (s == 0)
    {
      float assumed = cfg->use_input_upsampling ? cfg->input_image_blur_level * 2.f : cfg->input_image_blur_level;
      sigma = sqrtf((cfg->seed_scale_sigma * cfg->seed_scale_sigma) - (assumed * assumed));
    }
    else
    {
      float from = powf(k, (float)(s - 1)) * cfg->seed_scale_sigma;
      float to = from * k;
      sigma = sqrtf(to * to - from * from);
    }
    uint32_t n = (uint32_t)(int)(ceilf(sigma * 4.f) + 1.f); /* :92 */
    if (n > VKSIFT_MAX_TAPS)
      n = VKSIFT_MAX_TAPS;

    float g[VKSIFT_MAX_TAPS];
    float norm = 1.f;
    g[0] = 1.f;
    for (uint32_t i = 1; i < n; i++)
    {
      g[i] = (float)exp(-0.5 * powf((float)i, 2.f) / powf(sigma, 2.f)); /* double exp of float operands, :108 */
      norm += 2 * g[i];
    }
    for (uint32_t i = 0; i < n; i++)
      g[i] /= norm;

    float *t = taps + (size_t)s * VKSIFT_MAX_TAPS;
    memset(t, 0, sizeof(float) * VKSIFT_MAX_TAPS);
    if (!cfg->use_hardware_interpolated_blur)
    {
      memcpy(t, g, sizeof(float) * n);
      ntaps[s] = n;
      continue;
    }
    /* Sampler-interpolated variant (sift_detector.c:122-136, GaussianBlurInterpolated.comp:32-44): taps
     * (d, d+1) are fetched as one bilinear sample of weight c at offset off; that sample equals
     * c*(1-f)*texel[d] + c*f*texel[d+1], f = off - d. The loop only forms pairs while d+1 < n, so an
     * unpaired last tap is never sampled although it was counted in the normalisation. */
    t[0] = g[0];
    uint32_t used = 1;
    for (uint32_t d = 1; d + 1 < n; d += 2)
    {
      float c = g[d] + g[d + 1];
      float off = (((float)d * g[d]) + ((float)(d + 1) * g[d + 1])) / (g[d] + g[d + 1]);
      float fl = floorf(off);
      float f = off - fl;
      uint32_t di = (uint32_t)fl;
      t[di] += c * (1.f - f);
      t[di + 1] += c * f;
      if (di + 2 > used)
        used = di + 2;
    }
    ntaps[s] = used;
  }
}

uint32_t vksift_hm_desc_fp_table(const vksift_Config *cfg, float *tab, uint32_t cap)
{
  /* Largest descriptor window: sub-pixel scale <= S+1 (ExtractKeypoints.comp:198) so
   * sigma_oct <= seed * 2^((S+1)/S); radius = sqrt(2)*3*sigma_oct*2.5 (ComputeDescriptors.comp:107-109). */
  const uint32_t S = cfg->nb_scales_per_octave;
  float sigma_max = cfg->seed_scale_sigma * powf(2.f, (float)(S + 1) / (float)S) * 1.01f;
  uint32_t n_max = (uint32_t)(floorf(sqrtf(2.f) * 3.f * sigma_max * 2.5f + 0.5f)) / 2 + 2;
  if (n_max > cap)
    n_max = cap;
  const float es = -1.f / (2.f * 2 * 2);
  for (uint32_t n = 0; n < n_max; n++)
  {
    float m = 0.f;
    for (uint32_t i = 0; i < n; i++)
    {
      m += dm_expf(es * (float)((i * i) + (i * i))) * sqrtf(2.f);
      for (uint32_t j = i + 1; j < n; j++)
        m += dm_expf(es * (float)((i * i) + (j * j))) * sqrtf(2.f) * 2;
    }
    /* n == 0 cannot occur for a real keypoint (radius >= 1); keep the table total */
    int e = m > 0.f ? dm_ceil_log2f(m) : 0;
    int sh = 16 - e;
    if (sh < 0)
      sh = 0;
    if (sh > 31)
      sh = 31;
    tab[n] = (float)(1u << (uint32_t)sh);
  }
  return n_max;
}
