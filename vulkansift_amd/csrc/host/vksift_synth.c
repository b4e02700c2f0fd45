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
