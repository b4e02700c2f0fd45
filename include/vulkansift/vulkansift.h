/*
 * vulkansift.h — the vksift_* C API, served by hand-written HIP kernels on AMD MI355X (gfx950).
 *
 * Drop-in boundary: the 20 entry points below have the same names, signatures and blocking /
 * error behaviour as include/vulkansift/vulkansift.h:23-111 of maelaubert/VulkanSift. There is
 * no Vulkan underneath: "loadVulkan" initialises the HIP runtime, an "instance" owns one GPU,
 * its pyramid planes, its SIFT buffers and two HIP streams.
 *
 * Per-function reference citations (reference file src/vulkansift/vulkansift.c):
 *   vksift_loadVulkan :68  unloadVulkan :112  getAvailableGPUs :114  setLogLevel :132
 *   createInstance :168  destroyInstance :270  getDefaultConfig :66  isBufferAvailable :296
 *   detectFeatures :315  getFeaturesNumber :346  downloadFeatures :371  uploadFeatures :394
 *   matchFeatures :417  getMatchesNumber :443  downloadMatches :450
 *   getScaleSpaceNbOctaves :464  getScaleSpaceOctaveResolution :466
 *   downloadScaleSpaceImage :480  downloadDoGImage :500  presentDebugFrame :521
 */
#ifndef VULKAN_SIFT_H
#define VULKAN_SIFT_H

#ifdef __cplusplus
extern "C"
{
#endif

#if defined(_WIN32) || defined(_WIN64)
#define VKSIFT_EXPORT __declspec(dllexport)
#else
#define VKSIFT_EXPORT __attribute__((__visibility__("default")))
#endif

#include "vulkansift/vulkansift_types.h"

#include <stdbool.h>
#include <stdint.h>

  /* ---- runtime life-cycle ------------------------------------------------------------------ */

  /* Must be the first call. Fails with VKSIFT_VULKAN_ERROR when no usable GPU runtime/device is
   * present (callers use this to fall back to another SIFT) or when already loaded. */
  VKSIFT_EXPORT vksift_Result vksift_loadVulkan();
  VKSIFT_EXPORT void vksift_unloadVulkan();

  /* gpu_names == NULL: store the device count in *gpu_count.
   * gpu_names != NULL: copy *gpu_count names; index i is the value to put in gpu_device_index. */
  VKSIFT_EXPORT void vksift_getAvailableGPUs(uint32_t *gpu_count, VKSIFT_GPU_NAME *gpu_names);
  VKSIFT_EXPORT void vksift_setLogLevel(const vksift_LogLevel level);

  /* ---- instance ---------------------------------------------------------------------------- */

  typedef struct vksift_Instance_T *vksift_Instance;
  /* *instance_ptr must be NULL on entry. All device memory is reserved here for the configured
   * maxima, so steady-state detect/match calls never allocate. */
  VKSIFT_EXPORT vksift_Result vksift_createInstance(vksift_Instance *instance_ptr, const vksift_Config *config);
  /* Waits for the GPU, frees everything, sets *instance_ptr to NULL. */
  VKSIFT_EXPORT void vksift_destroyInstance(vksift_Instance *instance_ptr);
  VKSIFT_EXPORT vksift_Config vksift_getDefaultConfig();

  /* ---- pipelines (asynchronous) ------------------------------------------------------------
   * Both calls return once the work is queued on the instance's stream. A new detect/match call
   * first waits for the previous pipeline of the same instance. */

  /* image_data: row-major 8-bit grayscale, copied before the call returns. */
  VKSIFT_EXPORT void vksift_detectFeatures(vksift_Instance instance, const uint8_t *image_data, const uint32_t image_width, const uint32_t image_height,
                                           const uint32_t gpu_buffer_id);

  /* For every feature of buffer A: indices and L2 distances of its two nearest descriptors in B. */
  VKSIFT_EXPORT void vksift_matchFeatures(vksift_Instance instance, const uint32_t gpu_buffer_id_A, const uint32_t gpu_buffer_id_B);

  /* ---- transfers (blocking) ---------------------------------------------------------------- */

  VKSIFT_EXPORT uint32_t vksift_getFeaturesNumber(vksift_Instance instance, const uint32_t gpu_buffer_id);
  /* feats_ptr must hold vksift_getFeaturesNumber() records. */
  VKSIFT_EXPORT void vksift_downloadFeatures(vksift_Instance instance, vksift_Feature *feats_ptr, const uint32_t gpu_buffer_id);
  VKSIFT_EXPORT void vksift_uploadFeatures(vksift_Instance instance, const vksift_Feature *feats_ptr, const uint32_t nb_feats,
                                           const uint32_t gpu_buffer_id);
  /* Number of features buffer A held at the last vksift_matchFeatures() call; no GPU sync. */
  VKSIFT_EXPORT uint32_t vksift_getMatchesNumber(vksift_Instance instance);
  VKSIFT_EXPORT void vksift_downloadMatches(vksift_Instance instance, vksift_Match_2NN *matches);

  /* Non-blocking poll: false while a queued pipeline still reads or writes the buffer. */
  VKSIFT_EXPORT bool vksift_isBufferAvailable(vksift_Instance instance, const uint32_t gpu_buffer_id);

  /* ---- scale-space inspection (blocking) ---------------------------------------------------- */

  VKSIFT_EXPORT uint8_t vksift_getScaleSpaceNbOctaves(vksift_Instance instance);
  VKSIFT_EXPORT void vksift_getScaleSpaceOctaveResolution(vksift_Instance instance, const uint8_t octave, uint32_t *octave_images_width,
                                                          uint32_t *octave_images_height);
  /* scale in [0, nb_scales_per_octave+3); blurred_image holds width*height floats of that octave. */
  VKSIFT_EXPORT void vksift_downloadScaleSpaceImage(vksift_Instance instance, const uint8_t octave, const uint8_t scale, float *blurred_image);
  /* scale in [0, nb_scales_per_octave+2). */
  VKSIFT_EXPORT void vksift_downloadDoGImage(vksift_Instance instance, const uint8_t octave, const uint8_t scale, float *dog_image);

  /* Frame delimiter for graphics debuggers in the reference; here: warning + no-op. */
  VKSIFT_EXPORT void vksift_presentDebugFrame(vksift_Instance instance);

#ifdef __cplusplus
}
#endif

#endif /* VULKAN_SIFT_H */
