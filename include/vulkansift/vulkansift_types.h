/*
 * vulkansift_types.h — public POD types of the vksift_* C API (MI355X / HIP build).
 *
 * ABI contract: every struct, enum and typedef below has the same size, alignment, field
 * order and enumerator values as the reference declarations in
 * include/vulkansift/vulkansift_types.h:15-162 of maelaubert/VulkanSift, so that code compiled
 * against the reference headers can be relinked against this library unchanged.
 * The numbers are pinned by the _Static_asserts at the bottom of this file (x86-64 SysV).
 *
 * Identifiers that mention "Vulkan" are kept for source compatibility only: in this build
 * VKSIFT_VULKAN_ERROR means "GPU runtime (HIP) error".
 */
#ifndef VKSIFT_TYPES_H
#define VKSIFT_TYPES_H

#ifdef __cplusplus
extern "C"
{
#endif

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

/* Descriptor geometry: 4x4 spatial histograms of 8 orientation bins = 128 bytes. */
#define VKSIFT_FEATURE_NB_HIST 4
#define VKSIFT_FEATURE_NB_ORI 8

  /* Fixed-size device name slot used by vksift_getAvailableGPUs(). */
  typedef char VKSIFT_GPU_NAME[256];

  /* One SIFT feature, 164 bytes, identical to the on-device record.
   * descriptor byte index = y_hist*32 + x_hist*8 + orientation_bin. */
  typedef struct
  {
    float x;            /* position in input-image pixels */
    float y;
    float scale_x;      /* sub-pixel position inside the octave image it was found in */
    float scale_y;
    uint32_t scale_idx; /* Gaussian layer index inside the octave */
    int32_t octave_idx; /* -1 is the 2x up-sampled octave */
    float sigma;        /* blur level in input-image units */
    float orientation;  /* radians, [0, 2*pi) */
    float intensity;    /* refined DoG response */
    uint8_t descriptor[VKSIFT_FEATURE_NB_HIST * VKSIFT_FEATURE_NB_HIST * VKSIFT_FEATURE_NB_ORI];
  } vksift_Feature;

  /* Result of the brute-force 2-nearest-neighbour search for one feature of set A (20 bytes). */
  typedef struct
  {
    uint32_t idx_a;
    uint32_t idx_b1; /* nearest neighbour in B */
    uint32_t idx_b2; /* second nearest neighbour in B */
    float dist_a_b1; /* L2 distances (not squared) between the uint8 descriptors */
    float dist_a_b2;
  } vksift_Match_2NN;

  typedef enum
  {
    VKSIFT_NO_LOG,
    VKSIFT_LOG_ERROR,
    VKSIFT_LOG_WARNING,
    VKSIFT_LOG_INFO,
    VKSIFT_LOG_DEBUG
  } vksift_LogLevel;

  typedef enum
  {
    VKSIFT_DESCRIPTOR_FORMAT_UBC,   /* Lowe / OpenCV / SiftGPU bin direction */
    VKSIFT_DESCRIPTOR_FORMAT_VLFEAT /* VLFeat / PopSift bin direction */
  } vksift_DescriptorFormat;

  typedef enum
  {
    VKSIFT_PYRAMID_PRECISION_FLOAT32,
    VKSIFT_PYRAMID_PRECISION_FLOAT16 /* every scale-space image holds IEEE binary16 texels, arithmetic stays fp32 */
  } vksift_PyramidPrecisionMode;

  typedef enum
  {
    VKSIFT_SUCCESS,
    /* Rejected before any state change; the instance stays usable. */
    VKSIFT_INVALID_INPUT_ERROR,
    /* GPU runtime failure (HIP here); the instance must be destroyed. */
    VKSIFT_VULKAN_ERROR
  } vksift_Result;

  /* Opaque window handles for the graphics-debugger frame delimiter of the reference.
   * Kept for layout compatibility; this build has no presenter. */
  typedef struct
  {
    void *context;
    void *window;
  } vksift_ExternalWindowInfo;

  typedef struct
  {
    /* -- capacity -- */
    uint32_t input_image_max_size;   /* max width*height of an input image (default 1920*1080) */
    uint32_t sift_buffer_count;      /* number of device-resident SIFT buffers (default 2) */
    uint32_t max_nb_sift_per_buffer; /* feature capacity of one SIFT buffer (default 100000) */

    /* -- algorithm -- */
    bool use_input_upsampling;       /* build the pyramid from a 2x up-sampled image (default true) */
    uint8_t nb_octaves;              /* 0 = derive from the image resolution (default 0) */
    uint8_t nb_scales_per_octave;    /* default 3 */
    float input_image_blur_level;    /* assumed blur of the input image (default 0.5) */
    float seed_scale_sigma;          /* blur of scale 0 of the first octave (default 1.6) */
    float intensity_threshold;       /* DoG contrast threshold, divided by nb_scales_per_octave (default 0.04) */
    float edge_threshold;            /* principal-curvature ratio threshold (default 10) */
    uint32_t max_nb_orientation_per_keypoint; /* 0 = unlimited (library default 4) */
    vksift_DescriptorFormat descriptor_format; /* default UBC */

    /* -- device / implementation -- */
    int32_t gpu_device_index;        /* index into vksift_getAvailableGPUs(); <0 = auto (default -1) */
    bool use_hardware_interpolated_blur; /* paired-tap blur kernel of the reference's sampler path (default true) */
    vksift_PyramidPrecisionMode pyramid_precision_mode; /* default FLOAT32 */

    /* Called by every function that does not return a vksift_Result when it detects an error.
     * May throw through the C frames (the library is built with -fexceptions). Default: log + abort(). */
    void (*on_error_callback_function)(vksift_Result);

    /* -- graphics-debugger support (inert in this build) -- */
    bool use_gpu_debug_functions;
    vksift_ExternalWindowInfo gpu_debug_external_window_info;
  } vksift_Config;

/* ABI pins (values measured by compiling the reference headers with gcc 11.4 on x86-64). */
#if defined(__x86_64__) && !defined(VKSIFT_NO_ABI_ASSERTS)
#ifdef __cplusplus
#define VKSIFT_SASSERT(c, m) static_assert(c, m)
#else
#define VKSIFT_SASSERT(c, m) _Static_assert(c, m)
#endif
  VKSIFT_SASSERT(sizeof(vksift_Feature) == 164, "vksift_Feature must be 164 bytes");
  VKSIFT_SASSERT(offsetof(vksift_Feature, scale_idx) == 16, "scale_idx offset");
  VKSIFT_SASSERT(offsetof(vksift_Feature, intensity) == 32, "intensity offset");
  VKSIFT_SASSERT(offsetof(vksift_Feature, descriptor) == 36, "descriptor offset");
  VKSIFT_SASSERT(sizeof(vksift_Match_2NN) == 20, "vksift_Match_2NN must be 20 bytes");
  VKSIFT_SASSERT(sizeof(vksift_ExternalWindowInfo) == 16, "vksift_ExternalWindowInfo size");
  VKSIFT_SASSERT(sizeof(VKSIFT_GPU_NAME) == 256, "VKSIFT_GPU_NAME size");
  VKSIFT_SASSERT(sizeof(vksift_Config) == 88, "vksift_Config must be 88 bytes");
  VKSIFT_SASSERT(offsetof(vksift_Config, use_input_upsampling) == 12, "use_input_upsampling offset");
  VKSIFT_SASSERT(offsetof(vksift_Config, nb_octaves) == 13, "nb_octaves offset");
  VKSIFT_SASSERT(offsetof(vksift_Config, nb_scales_per_octave) == 14, "nb_scales_per_octave offset");
  VKSIFT_SASSERT(offsetof(vksift_Config, input_image_blur_level) == 16, "input_image_blur_level offset");
  VKSIFT_SASSERT(offsetof(vksift_Config, max_nb_orientation_per_keypoint) == 32, "max_nb_orientation offset");
  VKSIFT_SASSERT(offsetof(vksift_Config, descriptor_format) == 36, "descriptor_format offset");
  VKSIFT_SASSERT(offsetof(vksift_Config, gpu_device_index) == 40, "gpu_device_index offset");
  VKSIFT_SASSERT(offsetof(vksift_Config, use_hardware_interpolated_blur) == 44, "hw interp offset");
  VKSIFT_SASSERT(offsetof(vksift_Config, pyramid_precision_mode) == 48, "precision mode offset");
  VKSIFT_SASSERT(offsetof(vksift_Config, on_error_callback_function) == 56, "callback offset");
  VKSIFT_SASSERT(offsetof(vksift_Config, use_gpu_debug_functions) == 64, "debug flag offset");
  VKSIFT_SASSERT(offsetof(vksift_Config, gpu_debug_external_window_info) == 72, "window info offset");
#endif

#ifdef __cplusplus
}
#endif

#endif /* VKSIFT_TYPES_H */
