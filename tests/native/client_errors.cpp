// A C++ client in the spirit of the reference's src/examples/test_sift_error_handling.cpp: the user's error callback throws,
// the exception has to travel through the library's C frames (host C is compiled with -fexceptions) and the instance must
// stay usable after an invalid-input error.
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <vector>

#include <vulkansift/vulkansift.h>

#include "vksift_ext.h"

static void on_error(vksift_Result r)
{
  if (r == VKSIFT_INVALID_INPUT_ERROR)
    throw std::invalid_argument("invalid input");
  throw std::runtime_error("gpu error");
}

int main()
{
  vksift_setLogLevel(VKSIFT_NO_LOG);
  if (vksift_loadVulkan() != VKSIFT_SUCCESS)
    return 2;
  vksift_Config cfg = vksift_getDefaultConfig();
  cfg.on_error_callback_function = on_error;
  cfg.sift_buffer_count = 3;
  vksift_Instance inst = NULL;
  if (vksift_createInstance(&inst, &cfg) != VKSIFT_SUCCESS)
    return 3;
  int caught = 0, ok = 0;
  for (uint32_t i = 0; i < 6; i++)
  {
    try
    {
      vksift_getFeaturesNumber(inst, i); // i >= 3 is invalid
      ok++;
    }
    catch (const std::invalid_argument &)
    {
      caught++;
    }
  }
  // too small an image, then a NULL feature pointer: both invalid input
  std::vector<uint8_t> tiny(8 * 8, 0);
  try
  {
    vksift_detectFeatures(inst, tiny.data(), 8, 8, 0);
  }
  catch (const std::invalid_argument &)
  {
    caught++;
  }
  // the instance is still usable
  const uint32_t w = 128, h = 96;
  std::vector<uint8_t> img(w * h);
  vksift_ext_genSyntheticImage(7, w, h, 0, img.data());
  vksift_detectFeatures(inst, img.data(), w, h, 1);
  const uint32_t n = vksift_getFeaturesNumber(inst, 1);
  std::printf("ok %d caught %d features %u\n", ok, caught, n);
  vksift_destroyInstance(&inst);
  vksift_unloadVulkan();
  return (ok == 3 && caught == 4 && n > 0) ? 0 : 5;
}
