/*
 * vksift_ext.c — additive extensions of include/vksift_ext.h (profiling accessors, descriptor export, synthetic inputs)
 */
#include "vksift_internal.h"
#include <stddef.h>

/* ------------------------------------------------------------------------------------------------ */
/* extensions                                                                                       */
/* ------------------------------------------------------------------------------------------------ */
void vksift_ext_setProfiling(vksift_Instance instance, bool enabled)
{
  defer_sync(instance);
  instance->profiling = enabled;
  instance->prof[0].valid = instance->prof[1].valid = false;
  instance->prof[0].accounted = instance->prof[1].accounted = false;
  instance->match_timing_valid = false;
  memset(instance->acc_ms, 0, sizeof(instance->acc_ms));
  instance->acc_calls = 0;
  instance->acc_blur_launches = 0;
  instance->acc_blur_launches_all = 0;
  instance->acc_alg_bytes = 0;
  instance->acc_scan_bytes = 0;
}

_Static_assert(offsetof(vksift_ext_DetectTimings, scan_ms) == VKSIFT_EXT_DETECT_TIMINGS_V1_BYTES, "the v1 prefix of vksift_ext_DetectTimings moved");

static void copy_timings(vksift_ext_DetectTimings *dst, size_t dst_bytes, const vksift_ext_DetectTimings *src)
{
  memcpy(dst, src, dst_bytes < sizeof(*src) ? dst_bytes : sizeof(*src));
}

static void accumulated_timings(vksift_Instance instance, vksift_ext_DetectTimings *sum, uint32_t *nb_calls, bool reset)
{
  memset(sum, 0, sizeof(*sum));
  *nb_calls = 0;
  defer_sync(instance);
  if (!instance->profiling)
    return;
  vksift_hip_set_device(instance->device);
  wait_all(instance);
  account_timings(instance);
  sum->upload_ms = (float)instance->acc_ms[0];
  sum->pyramid_ms = (float)instance->acc_ms[1];
  sum->extrema_ms = (float)instance->acc_ms[2];
  sum->orientation_ms = (float)instance->acc_ms[3];
  sum->descriptor_ms = (float)instance->acc_ms[4];
  sum->total_ms = (float)instance->acc_ms[5];
  sum->scan_ms = (float)instance->acc_ms[6];
  sum->pyramid_all_ms = (float)instance->acc_ms[7];
  sum->nb_blur_launches_all = (uint32_t)instance->acc_blur_launches_all;
  sum->scan_algorithmic_bytes = instance->acc_scan_bytes;
  sum->nb_blur_launches = (uint32_t)instance->acc_blur_launches;
  sum->pyramid_algorithmic_bytes = instance->acc_alg_bytes;
  *nb_calls = instance->acc_calls;
  if (reset)
  {
    memset(instance->acc_ms, 0, sizeof(instance->acc_ms));
    instance->acc_calls = 0;
    instance->acc_blur_launches = 0;
    instance->acc_blur_launches_all = 0;
    instance->acc_alg_bytes = 0;
    instance->acc_scan_bytes = 0;
  }
}

void vksift_ext_getAccumulatedDetectTimingsSized(vksift_Instance instance, vksift_ext_DetectTimings *sum, size_t sum_bytes, uint32_t *nb_calls, bool reset)
{
  vksift_ext_DetectTimings t;
  accumulated_timings(instance, &t, nb_calls, reset);
  copy_timings(sum, sum_bytes, &t);
}

void vksift_ext_getAccumulatedDetectTimings(vksift_Instance instance, vksift_ext_DetectTimings *sum, uint32_t *nb_calls, bool reset)
{
  vksift_ext_getAccumulatedDetectTimingsSized(instance, sum, VKSIFT_EXT_DETECT_TIMINGS_V1_BYTES, nb_calls, reset);
}

void vksift_ext_getDeferredStats(vksift_Instance instance, uint64_t *nb_batches, uint64_t *nb_images)
{
  /* (no defer_sync: reading the statistics launches nothing) */
  if (nb_batches)
    *nb_batches = instance->defer_batches;
  if (nb_images)
    *nb_images = instance->defer_images;
}

vksift_Result vksift_ext_pinHostMemory(void *ptr, size_t bytes)
{
  if (!ptr || !bytes || !vksift_g_loaded)
    return VKSIFT_VULKAN_ERROR;
  return vksift_hip_host_register(ptr, bytes) == 0 ? VKSIFT_SUCCESS : VKSIFT_VULKAN_ERROR;
}

vksift_Result vksift_ext_unpinHostMemory(void *ptr)
{
  if (!ptr || !vksift_g_loaded)
    return VKSIFT_VULKAN_ERROR;
  return vksift_hip_host_unregister(ptr) == 0 ? VKSIFT_SUCCESS : VKSIFT_VULKAN_ERROR;
}

uint32_t vksift_ext_getScaleSpacePlacement(vksift_Instance instance, float gbps[8], uint32_t chosen[2])
{
  if (instance == NULL)
  {
    for (uint32_t i = 0; i < 8u; i++)
      gbps[i] = 0.f;
    chosen[0] = chosen[1] = 0;
    return 0;
  }
  for (uint32_t i = 0; i < 8u; i++)
    gbps[i] = i < instance->place_n ? instance->place_gbps[i] : 0.f;
  chosen[0] = instance->place_chosen[0], chosen[1] = instance->place_chosen[1];
  return instance->place_n;
}

static void last_timings(vksift_Instance instance, vksift_ext_DetectTimings *out)
{
  memset(out, 0, sizeof(*out));
  defer_sync(instance);
  const ProfSet *ps = &instance->prof[instance->prof_cur];
  if (!instance->profiling || !ps->valid)
    return;
  vksift_hip_set_device(instance->device);
  wait_all(instance);
  const vksift_hip_event *e = ps->ev_t;
  out->upload_ms = vksift_hip_event_elapsed_ms(e[0], e[1]);
  out->pyramid_ms = ps->overlap ? vksift_hip_event_elapsed_ms(ps->ev_pt[0], ps->ev_pt[1]) : vksift_hip_event_elapsed_ms(e[1], e[2]);
  out->extrema_ms = vksift_hip_event_elapsed_ms(e[2], e[3]);
  out->orientation_ms = vksift_hip_event_elapsed_ms(e[3], e[4]);
  out->descriptor_ms = vksift_hip_event_elapsed_ms(e[4], e[5]);
  out->total_ms = vksift_hip_event_elapsed_ms(e[0], e[6]);
  out->scan_ms = vksift_hip_event_elapsed_ms(e[2], ps->ev_scan);
  out->pyramid_all_ms = vksift_hip_event_elapsed_ms(ps->ev_pt[0], ps->ev_pt[2]);
  out->nb_blur_launches_all = ps->blur_launches_all;
  out->scan_algorithmic_bytes = instance->last_scan_bytes;
  out->nb_blur_launches = instance->last_blur_launches;
  out->pyramid_algorithmic_bytes = instance->last_alg_bytes;
}

void vksift_ext_getDetectTimingsSized(vksift_Instance instance, vksift_ext_DetectTimings *out, size_t out_bytes)
{
  vksift_ext_DetectTimings t;
  last_timings(instance, &t);
  copy_timings(out, out_bytes, &t);
}

void vksift_ext_getDetectTimings(vksift_Instance instance, vksift_ext_DetectTimings *out)
{
  vksift_ext_getDetectTimingsSized(instance, out, VKSIFT_EXT_DETECT_TIMINGS_V1_BYTES);
}

float vksift_ext_getMatchTime(vksift_Instance instance)
{
  defer_sync(instance);
  if (!instance->profiling || !instance->match_timing_valid)
    return -1.f;
  vksift_hip_set_device(instance->device);
  wait_all(instance);
  return vksift_hip_event_elapsed_ms(instance->ev_m[0], instance->ev_m[1]);
}

uint32_t vksift_ext_exportDescriptorsDevice(vksift_Instance instance, uint32_t gpu_buffer_id, uint8_t *d_descriptors)
{
  if (!buffer_idx_valid(instance, gpu_buffer_id) || d_descriptors == NULL)
  {
    logError(LOG_TAG, "vksift_ext_exportDescriptorsDevice() error: invalid input.");
    instance->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return 0;
  }
  vksift_Instance inst = instance;
  vksift_hip_set_device(inst->device);
  defer_sync(inst);
  uint32_t n = 0;
  HIP_CHECK(wait_all(inst), "stream synchronisation");
  {
    /* the matcher's cache entry of the buffer holds exactly these rows (filled now if the buffer changed since its last matching) */
    HIP_CHECK(refresh_match_cache(inst, &gpu_buffer_id, 1), "descriptor gather");
    HIP_CHECK(vksift_hip_post_words(inst->h_match_n + 2, inst->d_cache_n + gpu_buffer_id, 1, inst->stream), "descriptor gather");
    HIP_CHECK(vksift_hip_stream_sync(inst->stream), "descriptor gather");
    n = inst->h_match_n[2];
    HIP_CHECK(vksift_hip_memcpy_d2d(d_descriptors, inst->d_cache_desc + (uint64_t)gpu_buffer_id * inst->desc_slot_stride, (size_t)n * 128u, inst->stream),
              "descriptor gather");
    HIP_CHECK(vksift_hip_stream_sync(inst->stream), "descriptor gather");
  }
  return n;
gpu_error:
  logError(LOG_TAG, "vksift_ext_exportDescriptorsDevice() error when exporting descriptors.");
  instance->error_cb(VKSIFT_VULKAN_ERROR);
  return 0;
}

