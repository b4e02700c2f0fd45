/*
 * sift_oracle.h — CPU oracle for the vksift detect/match hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the algorithm that maelaubert/VulkanSift implements in its
 * GLSL compute shaders and Vulkan fixed-function blits. Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product library (libvulkansift.so) never does.
 *
 * PARITY UNPINNED: the reference ships no golden vectors, known-answer tests or fixtures
 *  and cannot be built or run in this environment (no Vulkan headers, loader,
 * ICD or glslc). The oracle is pinned only by (i) closed-form host values derivable from the
 * reference source (tests/test_oracle_host_math.py), (ii) a second, independent numpy restatement
 * of the dense stages (tests/test_oracle_crosscheck.py), (iii) internal invariants.
 */
#ifndef SIFT_ORACLE_H
#define SIFT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C"
{
#endif

#define ORC_MAX_KERNEL 20 /* VKSIFT_DETECTOR_MAX_GAUSSIAN_KERNEL_SIZE, sift_detector.h:9 */
#define ORC_MAX_OCTAVES 16

  /* Mirrors the algorithmic fields of vksift_Config (vulkansift_types.h:97-162). */
  typedef struct
  {
    uint32_t input_image_max_size;
    uint32_t max_nb_sift_per_buffer;
    int32_t use_input_upsampling;
    int32_t nb_octaves; /* 0 = auto */
    int32_t nb_scales_per_octave;
    float input_image_blur_level;
    float seed_scale_sigma;
    float intensity_threshold;
    float edge_threshold;
    uint32_t max_nb_orientation_per_keypoint;
    int32_t use_vlfeat_format;
    int32_t use_hardware_interpolated_blur;
    /* 0: libm expf/atan2f/... (independent of the product code)
     * 1: vulkansift_amd/csrc/detmath.h (bit-exact against the HIP kernels) */
    int32_t math_mode;
    /* VKSIFT_PYRAMID_PRECISION_FLOAT16 as this build defines it (the reference's own FLOAT16 mode binds R16 images to r32f
     * shader declarations, vulkansift_types.h:57-61 / sift_memory.c:139 vs GaussianBlur.comp:5 — undefined in Vulkan): every
     * image the reference allocates in the pyramid format — the blit target, the blur temporaries (horizontal-pass output), the
     * scale-space layers and the DoG layers — holds IEEE binary16 texels (round-to-nearest-even of the fp32 result), every
     * read widens exactly, all arithmetic stays fp32. */
    int32_t pyramid_fp16;
    /* 0: the interpolated blur's bilinear fetches are exact arithmetic taps (what the HIP kernels compute, bit for bit).
     * 1: a model of what a GPU's texture unit does with GaussianBlurInterpolated.comp:32-44 — every fetch interpolates its two
     *    texels with the fractional offset rounded to 8 bits (1/256 steps, the sub-texel precision of current desktop GPUs; Vulkan
     *    only guarantees >= 4 bits), and the two fetches of a tap pair are summed before the multiplication by the pair's weight.
     *    Measurement aid only (tests/test_sampler_model.py bounds how far the keypoints and descriptors move); nothing is
     *    bit-exact against it. */
    int32_t sampler_model;
  } orc_Config;

  /* 164-byte feature record == vksift_Feature. */
  typedef struct
  {
    float x, y, scale_x, scale_y;
    uint32_t scale_idx;
    int32_t octave_idx;
    float sigma, orientation, intensity;
    uint8_t descriptor[128];
  } orc_Feature;

  typedef struct
  {
    uint32_t idx_a, idx_b1, idx_b2;
    float dist_a_b1, dist_a_b2;
  } orc_Match;

  void orc_default_config(orc_Config *cfg);

  /* ---- host maths ---- */
  /* sift_memory.c:644-660 */
  uint32_t orc_max_nb_octaves(const orc_Config *cfg, uint32_t *rounded_max_image_size);
  /* sift_memory.c:15-38; returns the octave count, fills widths/heights (ORC_MAX_OCTAVES entries). */
  uint32_t orc_scale_space_info(const orc_Config *cfg, uint32_t w, uint32_t h, uint32_t *ow, uint32_t *oh);
  /* sift_memory.c:40-87 (capacities only). */
  void orc_section_caps(uint32_t max_nb_sift, uint32_t nb_octaves, uint32_t *caps);
  /* sift_detector.c:52-145. kernels: (S+3)*ORC_MAX_KERNEL floats exactly as the reference uploads
   * them (direct taps, or [c0,0,c12,off12,...] pairs in hardware-interpolated mode). */
  void orc_gaussian_kernels(const orc_Config *cfg, float *kernels, uint32_t *sizes, float *sigmas);
  /* One-sided direct tap weights equivalent (in exact arithmetic) to what the blur shader applies:
   * direct mode = the kernel itself; interpolated mode = pair (c,off) expanded to c*(1-f), c*f with
   * f = off - floor(off); an unpaired last tap is dropped (quirk Q9). ntaps includes the centre. */
  void orc_effective_taps(const orc_Config *cfg, float *taps, uint32_t *ntaps);

  /* ---- pyramid ---- */
  typedef struct orc_Pyramid orc_Pyramid;
  orc_Pyramid *orc_pyramid_build(const orc_Config *cfg, const uint8_t *img, uint32_t w, uint32_t h);
  void orc_pyramid_free(orc_Pyramid *p);
  uint32_t orc_pyramid_nb_octaves(const orc_Pyramid *p);
  void orc_pyramid_resolution(const orc_Pyramid *p, uint32_t o, uint32_t *w, uint32_t *h);
  const float *orc_pyramid_gauss(const orc_Pyramid *p, uint32_t o, uint32_t s); /* s < S+3, w*h floats */
  const float *orc_pyramid_dog(const orc_Pyramid *p, uint32_t o, uint32_t s);   /* s < S+2 */

  /* ---- detection ---- */
  /* Runs K4+K5+K6 on a built pyramid. Features come out in the reference's packed order: octave
   * sections concatenated; inside a section the extrema in raster order (scale, y, x), then the
   * extra-orientation copies in (keypoint, histogram bin) order. counts_found[o] is the un-clamped
   * per-octave counter (may exceed the section capacity, like nb_elem in the reference).
   * Returns the number of records written to out (each section clamped to its capacity). */
  uint32_t orc_detect_from_pyramid(const orc_Config *cfg, const orc_Pyramid *p, orc_Feature *out, uint32_t out_cap, uint32_t *counts_found);
  uint32_t orc_detect(const orc_Config *cfg, const uint8_t *img, uint32_t w, uint32_t h, orc_Feature *out, uint32_t out_cap, uint32_t *counts_found);
  /* Stage-level entry points used by the unit tests. */
  uint32_t orc_extract_keypoints(const orc_Config *cfg, const orc_Pyramid *p, uint32_t o, orc_Feature *out, uint32_t cap);
  /* returns number of orientations found (>=0); angles[] gets up to 36 values in bin order */
  uint32_t orc_orientations(const orc_Config *cfg, const orc_Pyramid *p, uint32_t o, const orc_Feature *kp, float *angles, uint32_t *hist36);
  void orc_descriptor(const orc_Config *cfg, const orc_Pyramid *p, uint32_t o, orc_Feature *kp, uint32_t *raw128);

  /* ---- matcher: Get2NearestNeighbors.comp:43-103 ---- */
  void orc_match_2nn(const orc_Feature *a, uint32_t na, const orc_Feature *b, uint32_t nb, orc_Match *out);
  /* same on bare 128-byte descriptor rows */
  void orc_match_2nn_desc(const uint8_t *a, uint32_t na, const uint8_t *b, uint32_t nb, orc_Match *out);
  /* cross-check + Lowe ratio filter (test_sift_match.cpp:90-107, perf_common.cpp:123-169) */
  uint32_t orc_filter_matches(const orc_Match *m12, uint32_t n12, const orc_Match *m21, uint32_t n21, float ratio, int cross_check, uint32_t *out_a,
                              uint32_t *out_b);

  /* detmath.h functions, exported for unit tests */
  float orc_dm_expf(float x);
  float orc_dm_exp2f(float x);
  float orc_dm_atan2f(float y, float x);
  float orc_dm_div_2pi(float x);
  uint32_t orc_check_div_3(void);
  float orc_dm_expf_nb(float x);
  float orc_dm_expf_nb_nonpos(float x);
  float orc_dm_sinf(float t);
  float orc_dm_cosf(float t);
  int orc_dm_ceil_log2f(float m);

#ifdef __cplusplus
}
#endif
#endif
