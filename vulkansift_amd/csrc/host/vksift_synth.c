/* vksift_synth.c — deterministic synthetic inputs for tests and benchmarks.
 * Not part of the reference API; exported through include/vksift_ext.h. */
#include "vksift_ext.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static uint64_t splitmix64(uint64_t *state)
{
  uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static double uniform01(uint64_t *state) { return (double)(splitmix64(state) >> 11) * (1.0 / 9007199254740992.0); }

/* uint8 image: 128 + sum_k A_k exp(-r^2 / 2 s_k^2) + U(-4,4), clamped. Blob centres uniform, s_k
 * log-uniform in [1.5, 12] px, |A_k| uniform in [20, 100] with random sign. */
void vksift_ext_genSyntheticImage(uint64_t seed, uint32_t width, uint32_t height, uint32_t nb_blobs, uint8_t *out)
{
  uint64_t st = seed;
  const size_t npx = (size_t)width * height;
  float *acc = (float *)malloc(sizeof(float) * npx);
  if (!acc)
  {
    memset(out, 128, npx);
    return;
  }
  for (size_t i = 0; i < npx; i++)
    acc[i] = 128.f;
  if (nb_blobs == 0)
    nb_blobs = (uint32_t)(npx / 60u); /* 5120 blobs on 640x480 -> ~2k SIFT features with the default config */
  for (uint32_t k = 0; k < nb_blobs; k++)
  {
    double cx = uniform01(&st) * width, cy = uniform01(&st) * height;
    double sg = exp(log(1.5) + uniform01(&st) * (log(12.0) - log(1.5)));
    double amp = 20.0 + uniform01(&st) * 80.0;
    if (splitmix64(&st) & 1ull)
      amp = -amp;
    int rad = (int)ceil(4.0 * sg);
    int x0 = (int)floor(cx) - rad, x1 = (int)floor(cx) + rad, y0 = (int)floor(cy) - rad, y1 = (int)floor(cy) + rad;
    if (x0 < 0)
      x0 = 0;
    if (y0 < 0)
      y0 = 0;
    if (x1 >= (int)width)
      x1 = (int)width - 1;
    if (y1 >= (int)height)
      y1 = (int)height - 1;
    const double inv = 1.0 / (2.0 * sg * sg);
    for (int y = y0; y <= y1; y++)
    {
      double dy = (y + 0.5) - cy;
      for (int x = x0; x <= x1; x++)
      {
        double dx = (x + 0.5) - cx;
        acc[(size_t)y * width + x] += (float)(amp * exp(-(dx * dx + dy * dy) * inv));
      }
    }
  }
  for (size_t i = 0; i < npx; i++)
  {
    double v = acc[i] + (uniform01(&st) * 8.0 - 4.0);
    v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
    out[i] = (uint8_t)v;
  }
  free(acc);
}

/* Two more image families for the parity tests (the blob images above hardly ever exercise the edge-response rejection of
 * ExtractKeypoints.comp:193-206 or the out-of-image orientation taps of ComputeOrientation.comp:97-100 on purpose):
 *   VKSIFT_EXT_SYNTH_EDGES   flat-shaded rotated rectangles and checker patches over a ramp, lightly box-smoothed: long step
 *                            edges (DoG extrema with a large principal-curvature ratio: rejected), corners and junctions (kept),
 *                            structures cut by the image border, plus U(-2,2) noise
 *   VKSIFT_EXT_SYNTH_FRACTAL 1/f value noise: six octaves (64 px cells down to 2 px) of smoothly interpolated random lattices,
 *                            amplitude x 0.8 per octave — texture at every scale, no preferred orientation, extrema in every octave of the pyramid
 * Everything integer / double arithmetic on splitmix64 draws: identical bytes on every machine. */
static void gen_edges(uint64_t st, uint32_t width, uint32_t height, uint8_t *out)
{
  const size_t npx = (size_t)width * height;
  float *img = (float *)malloc(sizeof(float) * npx * 2);
  if (!img)
  {
    memset(out, 128, npx);
    return;
  }
  float *tmp = img + npx;
  const double gx = uniform01(&st) * 60.0 - 30.0, gy = uniform01(&st) * 60.0 - 30.0;
  for (uint32_t y = 0; y < height; y++)
    for (uint32_t x = 0; x < width; x++)
      img[(size_t)y * width + x] = (float)(128.0 + gx * ((double)x / width - 0.5) + gy * ((double)y / height - 0.5));
  const uint32_t nshapes = (uint32_t)(npx / 2500u) + 8u;
  for (uint32_t k = 0; k < nshapes; k++)
  {
    /* centres may lie up to 10 % outside the image: shapes cut by the border */
    const double cx = (uniform01(&st) * 1.2 - 0.1) * width, cy = (uniform01(&st) * 1.2 - 0.1) * height;
    const double hw = 4.0 + uniform01(&st) * 0.12 * width, hh = 4.0 + uniform01(&st) * 0.12 * height;
    const double ang = uniform01(&st) * 3.141592653589793;
    const double ca = cos(ang), sa = sin(ang);
    const double level = 30.0 + uniform01(&st) * 195.0;
    const int checker = (splitmix64(&st) & 3ull) == 0ull; /* a quarter of the shapes are checker patches */
    const double cell = 5.0 + uniform01(&st) * 14.0;
    const double level2 = 30.0 + uniform01(&st) * 195.0;
    const double rad = sqrt(hw * hw + hh * hh);
    int x0 = (int)floor(cx - rad), x1 = (int)ceil(cx + rad), y0 = (int)floor(cy - rad), y1 = (int)ceil(cy + rad);
    x0 = x0 < 0 ? 0 : x0, y0 = y0 < 0 ? 0 : y0;
    x1 = x1 >= (int)width ? (int)width - 1 : x1, y1 = y1 >= (int)height ? (int)height - 1 : y1;
    for (int y = y0; y <= y1; y++)
      for (int x = x0; x <= x1; x++)
      {
        const double dx = (x + 0.5) - cx, dy = (y + 0.5) - cy;
        const double u = dx * ca + dy * sa, v = -dx * sa + dy * ca;
        if (fabs(u) > hw || fabs(v) > hh)
          continue;
        double val = level;
        if (checker && ((((long)floor((u + hw) / cell)) + ((long)floor((v + hh) / cell))) & 1L))
          val = level2;
        img[(size_t)y * width + x] = (float)val;
      }
  }
  /* 3x3 box filter (clamped): one-pixel-wide transitions instead of aliased staircases */
  for (uint32_t y = 0; y < height; y++)
    for (uint32_t x = 0; x < width; x++)
    {
      double acc = 0.0;
      for (int j = -1; j <= 1; j++)
        for (int i = -1; i <= 1; i++)
        {
          int yy = (int)y + j, xx = (int)x + i;
          yy = yy < 0 ? 0 : (yy >= (int)height ? (int)height - 1 : yy);
          xx = xx < 0 ? 0 : (xx >= (int)width ? (int)width - 1 : xx);
          acc += img[(size_t)yy * width + xx];
        }
      tmp[(size_t)y * width + x] = (float)(acc / 9.0);
    }
  for (size_t i = 0; i < npx; i++)
  {
    double v = tmp[i] + (uniform01(&st) * 4.0 - 2.0);
    v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
    out[i] = (uint8_t)v;
  }
  free(img);
}

static void gen_fractal(uint64_t st, uint32_t width, uint32_t height, uint8_t *out)
{
  const size_t npx = (size_t)width * height;
  double *acc = (double *)calloc(npx, sizeof(double));
  if (!acc)
  {
    memset(out, 128, npx);
    return;
  }
  double amp = 40.0, cell = 64.0;
  for (int oct = 0; oct < 7 && cell >= 1.5; oct++, amp *= 0.8, cell *= 0.5)
  {
    const uint32_t lw = (uint32_t)ceil(width / cell) + 2u, lh = (uint32_t)ceil(height / cell) + 2u;
    double *lat = (double *)malloc(sizeof(double) * (size_t)lw * lh);
    if (!lat)
      break;
    for (size_t i = 0; i < (size_t)lw * lh; i++)
      lat[i] = uniform01(&st) * 2.0 - 1.0;
    for (uint32_t y = 0; y < height; y++)
    {
      const double fy = (y + 0.5) / cell;
      const uint32_t iy = (uint32_t)fy;
      const double ty = fy - iy, sy = ty * ty * (3.0 - 2.0 * ty);
      for (uint32_t x = 0; x < width; x++)
      {
        const double fx = (x + 0.5) / cell;
        const uint32_t ix = (uint32_t)fx;
        const double tx = fx - ix, sx = tx * tx * (3.0 - 2.0 * tx);
        const double *l0 = lat + (size_t)iy * lw + ix, *l1 = l0 + lw;
        acc[(size_t)y * width + x] += amp * ((l0[0] * (1.0 - sx) + l0[1] * sx) * (1.0 - sy) + (l1[0] * (1.0 - sx) + l1[1] * sx) * sy);
      }
    }
    free(lat);
  }
  for (size_t i = 0; i < npx; i++)
  {
    double v = 128.0 + acc[i];
    v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
    out[i] = (uint8_t)v;
  }
  free(acc);
}

void vksift_ext_genSyntheticImageFamily(uint64_t seed, uint32_t width, uint32_t height, uint32_t family, uint8_t *out)
{
  if (family == VKSIFT_EXT_SYNTH_EDGES)
    gen_edges(seed ^ 0xED6E5ull, width, height, out);
  else if (family == VKSIFT_EXT_SYNTH_FRACTAL)
    gen_fractal(seed ^ 0xF2AC7A1ull, width, height, out);
  else
    vksift_ext_genSyntheticImage(seed, width, height, 0, out);
}

/* rows of min(255, trunc(512*|g|/||g||)) with g ~ N(0,1)^128 (Box-Muller) */
void vksift_ext_genSyntheticDescriptors(uint64_t seed, uint32_t rows, uint8_t *out)
{
  uint64_t st = seed;
  for (uint32_t r = 0; r < rows; r++)
  {
    double g[128], n2 = 0.0;
    for (int i = 0; i < 128; i += 2)
    {
      double u1 = uniform01(&st), u2 = uniform01(&st);
      if (u1 < 1e-300)
        u1 = 1e-300;
      double m = sqrt(-2.0 * log(u1));
      g[i] = fabs(m * cos(6.283185307179586 * u2));
      g[i + 1] = fabs(m * sin(6.283185307179586 * u2));
      n2 += g[i] * g[i] + g[i + 1] * g[i + 1];
    }
    double inv = 512.0 / sqrt(n2);
    for (int i = 0; i < 128; i++)
    {
      double v = floor(g[i] * inv);
      out[(size_t)r * 128 + i] = (uint8_t)(v > 255.0 ? 255.0 : v);
    }
  }
}
