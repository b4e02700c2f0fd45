/*
 * vksift_api.c — the vksift_* C API over the HIP kernel layer (include/vksift_hip.h).
 *
 * Mirrors the behaviour of the reference façade src/vulkansift/vulkansift.c (validation, error
 * callback contract, blocking/asynchronous split) and the bookkeeping part of
 * src/vulkansift/sift_memory.c (octave geometry, SIFT-buffer sections, packed/sectioned state,
 * count read-back), with Vulkan objects replaced by plain HBM allocations:
 *
 *   pyramid   one allocation (two for batch instances); per image: for each octave (S+3) Gaussian planes, fp32 (binary16 in the
 *             FLOAT16 mode), row pitch padded to 64 texels. No DoG planes: D[s] = G[s+1] - G[s] is formed where it is consumed
 *   buffers   sift_buffer_count x max_nb_sift_per_buffer records of 164 B; after a detection a buffer
 *             is "sectioned" (one section per octave, capacities from sift_memory.c:40-87), after an
 *             upload it is one packed section
 *   counters  found[buffer][octave] u32 in HBM + a pinned host mirror filled by an async copy at
 *             the end of every detection (replaces the HOST_VISIBLE count staging buffers)
 *   streams   one HIP stream per instance = the reference's single general queue; two events play
 *             the role of end_of_detection_fence / end_of_matching_fence
 *
 * Built as C with -fexceptions: user error callbacks may throw through these frames
 * (vulkansift_types.h:148-152 of the reference).
 *
 * This file: defaults, validation, runtime life-cycle. vksift_instance.c: instances, layout, synchronisation helpers.
 * vksift_detect.c: the detection pipeline. vksift_buffers.c: feature accessors + scale-space inspection.
 * vksift_match.c: matching. vksift_ext.c: extensions. Shared private definitions: vksift_internal.h.
 */
#include "vksift_internal.h"

bool vksift_g_loaded = false;

/* ------------------------------------------------------------------------------------------------ */
/* defaults + validation (vulkansift.c:31-66, 550-661)                                              */
/* ------------------------------------------------------------------------------------------------ */
void default_error_callback(vksift_Result err)
{
  if (err == VKSIFT_INVALID_INPUT_ERROR)
    logDebug(LOG_TAG, "Aborting after invalid input error...");
  else if (err == VKSIFT_VULKAN_ERROR)
    logDebug(LOG_TAG, "Aborting after GPU runtime error...");
  abort();
}

vksift_Config vksift_getDefaultConfig()
{
  vksift_Config c;
  memset(&c, 0, sizeof(c));
  c.input_image_max_size = 1920u * 1080u;
  c.sift_buffer_count = 2u;
  c.max_nb_sift_per_buffer = 100000u;
  c.use_input_upsampling = true;
  c.nb_octaves = 0;
  c.nb_scales_per_octave = 3u;
  c.input_image_blur_level = 0.5f;
  c.seed_scale_sigma = 1.6f;
  c.intensity_threshold = 0.04f;
  c.edge_threshold = 10.f;
  c.max_nb_orientation_per_keypoint = 4; /* the code default of the reference (its header comment says 0) */
  c.descriptor_format = VKSIFT_DESCRIPTOR_FORMAT_UBC;
  c.gpu_device_index = -1;
  c.use_hardware_interpolated_blur = true;
  c.pyramid_precision_mode = VKSIFT_PYRAMID_PRECISION_FLOAT32;
  c.on_error_callback_function = default_error_callback;
  c.use_gpu_debug_functions = false;
  c.gpu_debug_external_window_info.context = NULL;
  c.gpu_debug_external_window_info.window = NULL;
  return c;
}

static bool cfg_check(bool cond, const char *msg)
{
  if (!cond)
    logError(LOG_TAG, "%s", msg);
  return cond;
}

bool config_is_valid(const vksift_Config *c)
{
  bool ok = true;
  ok &= cfg_check(c->input_image_max_size >= 1024, "vksift_Config rejected: input_image_max_size below the 1024-pixel minimum");
  ok &= cfg_check(c->sift_buffer_count > 0, "vksift_Config rejected: sift_buffer_count is 0, at least one buffer is needed");
  ok &= cfg_check(c->max_nb_sift_per_buffer > 0, "vksift_Config rejected: max_nb_sift_per_buffer is 0, a buffer must hold at least one feature");
  ok &= cfg_check(c->nb_scales_per_octave > 0, "vksift_Config rejected: nb_scales_per_octave is 0, an octave needs at least one scale");
  ok &= cfg_check(c->nb_scales_per_octave <= VKSIFT_MAX_SCALES, "vksift_Config rejected: nb_scales_per_octave above 13, the limit of this build");
  ok &= cfg_check(c->input_image_blur_level >= 0.f, "vksift_Config rejected: negative input_image_blur_level");
  ok &= cfg_check(c->seed_scale_sigma >= 0, "vksift_Config rejected: negative seed_scale_sigma");
  ok &= cfg_check(((c->use_input_upsampling ? 2.f : 1.f) * c->input_image_blur_level) <= c->seed_scale_sigma,
                  "vksift_Config rejected: seed_scale_sigma is smaller than the blur already in the input (input_image_blur_level, doubled by the up-sampling)");
  ok &= cfg_check(c->intensity_threshold >= 0.f, "vksift_Config rejected: negative intensity_threshold");
  ok &= cfg_check(c->edge_threshold >= 0.f, "vksift_Config rejected: negative edge_threshold");
  ok &= cfg_check(c->on_error_callback_function != NULL, "vksift_Config rejected: on_error_callback_function is NULL");
  switch (c->pyramid_precision_mode)
  {
  case VKSIFT_PYRAMID_PRECISION_FLOAT16:
  case VKSIFT_PYRAMID_PRECISION_FLOAT32:
    break;
  default:
    logError(LOG_TAG, "vksift_Config rejected: pyramid_precision_mode is neither FLOAT32 nor FLOAT16");
    ok = false;
  }
  return ok;
}

/* Note: the reference tests `idx > count` (vulkansift.c:588), letting idx == count through to an
 * out-of-bounds access; its header and error-handling demo document `idx >= count` as invalid, which
 * is what this build enforces. */
bool buffer_idx_valid(vksift_Instance inst, uint32_t idx)
{
  if (idx >= inst->cfg.sift_buffer_count)
  {
    logError(LOG_TAG, "SIFT buffer index %d out of range: the instance has %d buffer(s).", idx, inst->cfg.sift_buffer_count);
    return false;
  }
  return true;
}

bool resolution_valid(vksift_Instance inst, uint32_t w, uint32_t h)
{
  uint64_t size = (uint64_t)w * h;
  if (size > inst->max_image_size)
  {
    logError(LOG_TAG, "Image of %d x %d = %llu pixels exceeds the %d pixels the instance was configured for.", w, h, (unsigned long long)size,
             inst->max_image_size);
    return false;
  }
  if (size < 1024u)
  {
    logError(LOG_TAG, "Image of %d x %d = %llu pixels is below the 1024-pixel minimum.", w, h, (unsigned long long)size);
    return false;
  }
  return true;
}

/* ------------------------------------------------------------------------------------------------ */
/* runtime life-cycle                                                                               */
/* ------------------------------------------------------------------------------------------------ */
vksift_Result vksift_loadVulkan()
{
  if (vksift_g_loaded)
  {
    logError(LOG_TAG, "vksift_loadVulkan() failure: the GPU runtime is already loaded.");
    return VKSIFT_VULKAN_ERROR;
  }
  int e = vksift_hip_init();
  if (e != 0)
  {
    logError(LOG_TAG, "vksift_loadVulkan() failure when setting up the HIP runtime: %s", vksift_hip_error_string(e));
    return VKSIFT_VULKAN_ERROR;
  }
  vksift_g_loaded = true;
  logInfo(LOG_TAG, "vksift_loadVulkan() success");
  return VKSIFT_SUCCESS;
}

void vksift_unloadVulkan() { vksift_g_loaded = false; }

void vksift_getAvailableGPUs(uint32_t *gpu_count, VKSIFT_GPU_NAME *gpu_names)
{
  uint32_t n = (uint32_t)vksift_hip_device_count();
  if (gpu_names == NULL)
  {
    *gpu_count = n;
    return;
  }
  if (*gpu_count > n)
    *gpu_count = n;
  for (uint32_t i = 0; i < *gpu_count; i++)
    vksift_hip_device_name((int)i, gpu_names[i]);
}

void vksift_setLogLevel(vksift_LogLevel level)
{
  switch (level)
  {
  case VKSIFT_NO_LOG:
    vksift_log_set_level(VKSIFT_LOGLVL_NONE);
    break;
  case VKSIFT_LOG_ERROR:
    vksift_log_set_level(VKSIFT_LOGLVL_ERROR);
    break;
  case VKSIFT_LOG_WARNING:
    vksift_log_set_level(VKSIFT_LOGLVL_WARNING);
    break;
  case VKSIFT_LOG_INFO:
    vksift_log_set_level(VKSIFT_LOGLVL_INFO);
    break;
  case VKSIFT_LOG_DEBUG:
    vksift_log_set_level(VKSIFT_LOGLVL_DEBUG);
    break;
  default:
    logError(LOG_TAG, "vksift_setLogLevel(): unknown level, nothing changed");
    break;
  }
}

