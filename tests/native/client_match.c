/* A plain-C client of the public API only (include/vulkansift/vulkansift.h + the synthetic-image helper of vksift_ext.h),
 * following the call sequence of the reference's src/examples/test_sift_match.cpp:19-78: load, create, detect two images
 * into two buffers, match both ways, download, destroy, unload. Prints a digest that the test compares with the Python
 * mirror's results for the same inputs. Built with gcc against libvulkansift.so: this is the "drop-in" at the C level. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vulkansift/vulkansift.h>

#include "vksift_ext.h"

static uint64_t fnv(const void *p, size_t n, uint64_t h)
{
  const unsigned char *b = (const unsigned char *)p;
  for (size_t i = 0; i < n; i++)
    h = (h ^ b[i]) * 1099511628211ull;
  return h;
}

int main(void)
{
  const uint32_t w = 320, h = 240;
  uint8_t *img1 = malloc(w * h), *img2 = malloc(w * h);
  vksift_ext_genSyntheticImage(101, w, h, 0, img1);
  vksift_ext_genSyntheticImage(102, w, h, 0, img2);

  vksift_setLogLevel(VKSIFT_LOG_ERROR);
  if (vksift_loadVulkan() != VKSIFT_SUCCESS)
    return 2;
  vksift_Config cfg = vksift_getDefaultConfig();
  cfg.input_image_max_size = w * h;
  vksift_Instance inst = NULL;
  if (vksift_createInstance(&inst, &cfg) != VKSIFT_SUCCESS)
    return 3;

  vksift_detectFeatures(inst, img1, w, h, 0u);
  vksift_detectFeatures(inst, img2, w, h, 1u);
  uint32_t n1 = vksift_getFeaturesNumber(inst, 0u), n2 = vksift_getFeaturesNumber(inst, 1u);
  vksift_Feature *f1 = malloc(sizeof(vksift_Feature) * (n1 + 1)), *f2 = malloc(sizeof(vksift_Feature) * (n2 + 1));
  vksift_downloadFeatures(inst, f1, 0u);
  vksift_downloadFeatures(inst, f2, 1u);

  vksift_matchFeatures(inst, 0u, 1u);
  uint32_t m12n = vksift_getMatchesNumber(inst);
  vksift_Match_2NN *m12 = malloc(sizeof(vksift_Match_2NN) * (m12n + 1));
  vksift_downloadMatches(inst, m12);
  vksift_matchFeatures(inst, 1u, 0u);
  uint32_t m21n = vksift_getMatchesNumber(inst);
  vksift_Match_2NN *m21 = malloc(sizeof(vksift_Match_2NN) * (m21n + 1));
  vksift_downloadMatches(inst, m21);

  /* the CPU filter of the reference example (cross-check + Lowe ratio 0.75) */
  uint32_t kept = 0;
  for (uint32_t i = 0; i < m12n; i++)
  {
    uint32_t j = m12[i].idx_b1;
    if (m21[j].idx_b1 == i && (m12[i].dist_a_b1 / m12[i].dist_a_b2) < 0.75 && (m21[j].dist_a_b1 / m21[j].dist_a_b2) < 0.75)
      kept++;
  }

  uint32_t ow = 0, oh = 0;
  vksift_getScaleSpaceOctaveResolution(inst, 0, &ow, &oh);
  printf("features %u %u matches %u %u kept %u octaves %u oct0 %ux%u available %d\n", n1, n2, m12n, m21n, kept, (unsigned)vksift_getScaleSpaceNbOctaves(inst), ow,
         oh, (int)vksift_isBufferAvailable(inst, 0u));
  printf("digest %016llx %016llx %016llx %016llx\n", (unsigned long long)fnv(f1, sizeof(vksift_Feature) * n1, 1469598103934665603ull),
         (unsigned long long)fnv(f2, sizeof(vksift_Feature) * n2, 1469598103934665603ull),
         (unsigned long long)fnv(m12, sizeof(vksift_Match_2NN) * m12n, 1469598103934665603ull),
         (unsigned long long)fnv(m21, sizeof(vksift_Match_2NN) * m21n, 1469598103934665603ull));

  vksift_destroyInstance(&inst);
  vksift_unloadVulkan();
  return inst == NULL ? 0 : 4;
}
