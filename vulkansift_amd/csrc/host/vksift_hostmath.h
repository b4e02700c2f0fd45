/* vksift_hostmath.h — host-side scalar maths of the detector (octave geometry, SIFT-buffer section
 * sizes, Gaussian tap tables, descriptor fixed-point table). fp32 like the reference so that integer
 * truncations land on the same values. */
#ifndef VKSIFT_HOSTMATH_H
#define VKSIFT_HOSTMATH_H

#include <stdbool.h>
#include <stdint.h>

#include "vulkansift/vulkansift_types.h"

#define VKSIFT_MAX_TAPS 20     /* reference: VKSIFT_DETECTOR_MAX_GAUSSIAN_KERNEL_SIZE (sift_detector.h:9) */
#define VKSIFT_MAX_OCTAVES 16
#define VKSIFT_MAX_SCALES 13   /* nb_scales_per_octave + 3 <= 16 tap tables */

/* sift_memory.c:644-660: side of the square that bounds the configured max image, its area, and the
 * octave count that square would get (capped by config->nb_octaves when > 0). */
uint32_t vksift_hm_max_octaves(const vksift_Config *cfg, uint32_t *rounded_max_image_size);
/* sift_memory.c:15-38 */
uint32_t vksift_hm_octaves_for(const vksift_Config *cfg, uint32_t max_octaves, uint32_t w, uint32_t h, uint32_t *ow, uint32_t *oh);
/* sift_memory.c:61-87 (capacities only; the Vulkan offset alignment is not reproduced) */
void vksift_hm_section_caps(uint32_t max_nb_sift, uint32_t nb_octaves, uint32_t *caps);
/* sift_detector.c:52-145 reduced to what the blur shaders actually apply: one-sided direct tap
 * weights per scale, centre first. taps: (S+3) rows of VKSIFT_MAX_TAPS floats. */
void vksift_hm_blur_taps(const vksift_Config *cfg, float *taps, uint32_t *ntaps);
/* ComputeDescriptors.comp:116-124: fixed-point multiplier as a function of n = int_radius/2.
 * Returns the number of entries written (<= cap). */
uint32_t vksift_hm_desc_fp_table(const vksift_Config *cfg, float *tab, uint32_t cap);

#endif
