/*
 * vksift_hip.h — the thin C-ABI between the C host (vulkansift_amd/csrc/host/) and the hand-written
 * HIP kernels for gfx950 (vulkansift_amd/csrc/hip/). Plain pointers, sizes and an opaque stream
 * handle only: no HIP, torch or C++ types appear in any signature, so the same entry points can be
 * bound from C, ctypes or any FFI.
 *
 * Each launch shim replaces one recorded Vulkan command of the reference's detection / matching
 * command buffers (reference file:line given per function; paths relative to src/vulkansift/).
 * All shims are asynchronous on `stream` and return 0 on success or a non-zero hipError_t value.
 *
 * Data layout in HBM:
 *   plane      : fp32, row-major, row pitch `pitch` floats (multiple of 64 floats = 256 B)
 *   octave     : (S+3) Gaussian planes, plane stride = pitch*h floats. DoG planes are not stored: D[s] = G[s+1] - G[s] is
 *                formed in registers where it is consumed (extrema scan, refinement), bit-identically
 *   batch      : image b of a batched detect lives `img_stride` floats after image b-1
 *   SIFT buffer: per octave section of 164-byte vksift_Feature records; counters live in a
 *                separate u32 array (found[o], un-clamped like nb_elem in the reference)
 */
#ifndef VKSIFT_HIP_H
#define VKSIFT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C"
{
#endif

#define VKSIFT_HIP_MAX_TAPS 20 /* VKSIFT_DETECTOR_MAX_GAUSSIAN_KERNEL_SIZE, sift_detector.h:9 */
#define VKSIFT_HIP_MATCH_CHUNKS 32   /* partial top-2 lists per A row of the single-pair matcher (merged exactly) */
#define VKSIFT_HIP_MATCH_SMALL_NA 1536u /* single pairs with N_A <= this (and, where the host knows it, N_B <= ..._SMALL_NB) */
#define VKSIFT_HIP_MATCH_SMALL_NB 4096u /* take the one-launch small kernel and need no partial lists */
/* u32 words of scratch vksift_hip_match_2nn_desc needs for na query rows against nb reference rows (norms of A and B, row flags /
 * row list, per-row partial lists of the decomposed kernels: 16 words per piece for the cell scan of large reference sets) */
#define VKSIFT_HIP_MATCH_SCRATCH_U32(na, nb) (2u * (size_t)(na) + (size_t)(nb) + 72u + (size_t)(na) * 16u * VKSIFT_HIP_MATCH_CHUNKS)
#define VKSIFT_HIP_ABI_VERSION 6u      /* bumped whenever a signature or a scratch contract of this header changes (vksift_hip_abi_version) */
#define VKSIFT_HIP_GATHER_SLOTS 512u   /* SIFT buffers one vksift_hip_gather_sections launch serves */
#define VKSIFT_HIP_MATCH_SLOTS 256u    /* pairs one vksift_hip_match_2nn_async launch sequence serves */
#define VKSIFT_HIP_MATCH_PK_NB 32768u  /* reference sets of at most this many rows take the branch-free packed-key kernel (k_match_pk) */
#define VKSIFT_HIP_MAX_ORI 18  /* a 36-bin circular histogram has at most 18 strict local maxima */

  typedef void *vksift_hip_stream;
  typedef void *vksift_hip_graph; /* an instantiated hipGraph (hipGraphExec_t) */
  typedef void *vksift_hip_event;

  /* ------------------------------------------------------------------ runtime (replaces the vkenv directory) */
  int vksift_hip_init(void);                          /* vulkan_device.c:17 vkenv_createInstance */
  uint32_t vksift_hip_abi_version(void);              /* VKSIFT_HIP_ABI_VERSION the library was built with */
  int vksift_hip_device_count(void);                  /* vulkan_device.c: vkenv_getPhysicalDevicesProperties */
  int vksift_hip_device_name(int idx, char *out256);
  int vksift_hip_set_device(int idx);
  size_t vksift_hip_device_free_mem(void);
  void *vksift_hip_malloc(size_t bytes);              /* NULL on failure */
  void vksift_hip_free(void *p);
  void *vksift_hip_host_malloc(size_t bytes);         /* pinned; replaces HOST_VISIBLE staging buffers */
  void vksift_hip_host_free(void *p);
  int vksift_hip_host_register(void *p, size_t bytes); /* page-lock caller memory: copies into it become DMA transfers */
  int vksift_hip_host_unregister(void *p);
  int vksift_hip_is_pinned(const void *p);             /* 1: page-locked host memory (registered or vksift_hip_host_malloc) */
  vksift_hip_stream vksift_hip_stream_create(void);
  void vksift_hip_stream_destroy(vksift_hip_stream s);
  int vksift_hip_stream_sync(vksift_hip_stream s);    /* vkWaitForFences */
  int vksift_hip_stream_busy(vksift_hip_stream s);    /* vkGetFenceStatus: 1 busy, 0 idle, <0 error */
  vksift_hip_event vksift_hip_event_create(void);
  void vksift_hip_event_destroy(vksift_hip_event e);
  int vksift_hip_event_record(vksift_hip_event e, vksift_hip_stream s);
  int vksift_hip_event_sync(vksift_hip_event e);
  int vksift_hip_event_busy(vksift_hip_event e);
  float vksift_hip_event_elapsed_ms(vksift_hip_event a, vksift_hip_event b);
  int vksift_hip_stream_wait_event(vksift_hip_stream s, vksift_hip_event e);
  /* hipGraph capture of everything enqueued to s (and to streams forked from it through events) between begin and end;
   * the launch-bound single-image pipeline (~100 short kernels) is replayed with one vksift_hip_graph_launch. */
  int vksift_hip_capture_begin(vksift_hip_stream s);
  int vksift_hip_capture_end(vksift_hip_stream s, vksift_hip_graph *out);
  int vksift_hip_graph_launch(vksift_hip_graph g, vksift_hip_stream s);
  void vksift_hip_graph_destroy(vksift_hip_graph g);
  int vksift_hip_memcpy_h2d(void *dst, const void *src, size_t n, vksift_hip_stream s);
  int vksift_hip_memcpy_d2h(void *dst, const void *src, size_t n, vksift_hip_stream s);
  int vksift_hip_memcpy_d2d(void *dst, const void *src, size_t n, vksift_hip_stream s);
  int vksift_hip_memcpy2d_d2h(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t height, vksift_hip_stream s);
  /* kernel store of n_words into a vksift_hip_host_malloc allocation (see runtime.hip: keeps dependent read-backs off the copy engine) */
  int vksift_hip_post_words(uint32_t *host_mapped_dst, const uint32_t *src, size_t n_words, vksift_hip_stream s);
  int vksift_hip_memset(void *dst, int value, size_t n, vksift_hip_stream s);
  const char *vksift_hip_error_string(int err);
  void vksift_hip_range_push(const char *name);       /* roctx marker == VK_EXT_debug_marker region */
  void vksift_hip_range_pop(void);

  /* Development / test knobs of the launch shims (A/B tools, tests of fallback paths). Every setting produces identical results.
   * PROCESS-WIDE and not synchronised: they act on every instance of the process; set them before instances are created (tests, tools),
   * never while another thread is inside a library call. */
  enum
  {
    VKSIFT_TUNE_WG_TARGET = 0,  /* waves per strip-march launch aimed at (0 = built-in) */
    VKSIFT_TUNE_WIDE_MASK = 1,  /* bit n: n-tap launches take the four-texels-per-lane form (-1 = built-in) */
    VKSIFT_TUNE_MULTI_MAX = 2,  /* octaves per multi-octave launch, 1..8 (0 = built-in 8): the cutting of longer octave lists into runs is
                                 * otherwise only reached by images of 4097 pixels and more on the shortest side */
    VKSIFT_TUNE_REFINE_PTR = 3, /* 1: the refinement kernels address the scale-space through pointers everywhere (the form octaves beyond
                                 * 2 GiB take) instead of one buffer resource per image and octave */
    VKSIFT_TUNE_SCAN_FORM = 4,  /* development: launch form of the cell-scan matcher (0 = built-in) */
    VKSIFT_TUNE_PAIR_FORM = 5,  /* two-scale blur launch: 0 built-in, 1 two texels per lane, 2 four texels per lane */
    VKSIFT_TUNE_PYR_GATE = 6,   /* 1: the next detection's scale-space starts behind the matching queued before it (default: beside it) */
    VKSIFT_TUNE_DENSE_ROWS = 7, /* 1: the descriptor launch does not write the matcher's dense rows (the gather pass of the first matching does, as
                                 * for uploaded buffers); A/B and the bit-identity matrix */
    VKSIFT_TUNE_SEED_WG = 8,    /* waves aimed at by the fused up-sampling + seed launch (0 = built-in) */
    VKSIFT_TUNE_SCAN_BAND = 9,  /* rows per wave of the streaming extrema scan (0 = built-in: 32 on large octaves, 16 on small ones) */
    VKSIFT_TUNE_TAIL_MULTI = 10, /* 1: no multi-octave launches for scales S+1, S+2 (batches queue every octave in full; forked detections one launch per
                                  * octave and scale) */
    VKSIFT_TUNE_ZERO_COPY = 11,  /* 1: a single host image is copied into device memory in front of the seed launch (default: the launch reads the pinned
                                  * staging buffer itself) */
    VKSIFT_TUNE_MIN_MARCH = 12,  /* output rows per wave of the smallest strip-march launches (a single image's): 0 = built-in */
    VKSIFT_TUNE_COUNT = 16
  };
  int vksift_hip_tune(int knob, int value);
  int vksift_hip_tune_get(int knob);

  /* ------------------------------------------------------------------ pyramid */
  /* A batch of same-sized planes. */
  typedef struct
  {
    float *base;         /* plane of image 0 */
    uint32_t w, h;       /* valid extent */
    uint32_t pitch;      /* floats per row */
    uint64_t img_stride; /* texels between consecutive images of the batch */
    uint32_t fp16;       /* 0: fp32 texels; 1: IEEE binary16 texels (VKSIFT_PYRAMID_PRECISION_FLOAT16: stored round-to-nearest-even,
                          * widened exactly on every read, all arithmetic fp32); base then points at 2-byte texels */
    uint32_t reverse;    /* dispatch-order hint, read from the DESTINATION plane of a launch: 1 = every XCD walks its share of the
                          * (image, row segment, strip) space back to front. Consecutive launches of a chain alternate it, so that
                          * a launch starts on the texels its predecessor touched last — the ones still in the 256 MiB Infinity
                          * Cache — instead of on the ones it evicted first. Results do not depend on it */
  } vksift_hip_Plane;

  /* vkCmdCopyBufferToImage + vkCmdBlitImage(LINEAR) of sift_detector.c:881,909-916:
   * u8 row-major images (src_stride bytes between images) -> fp32 plane, value/255, bilinear 2x (or
   * 1:1 copy when the sizes match), clamp-to-edge. */
  int vksift_hip_input_blit(const uint8_t *src, uint32_t sw, uint32_t sh, uint64_t src_img_stride, vksift_hip_Plane dst, uint32_t batch,
                            vksift_hip_stream s);

  /* One Gaussian scale step = the H and V GaussianBlur*.comp dispatches of sift_detector.c:927-1001
   * fused through LDS. taps[0..ntaps) are one-sided direct weights, centre first; the borders use
   * mirrored-repeat addressing. src and dst must not alias. */
  int vksift_hip_blur(vksift_hip_Plane src, vksift_hip_Plane dst, const float *taps, uint32_t ntaps, uint32_t batch, vksift_hip_stream s);

  /* TWO consecutive scale steps in one launch: dst1 = blur(src, taps1), dst2 = blur(dst1, taps2) — the source plane is read once and
   * scale s never re-read (12 bytes per texel instead of 16). Bit-identical to two vksift_hip_blur calls. Returns -1 without
   * launching anything when the tap counts, the texel type or the shape are not covered: the caller then issues the two calls. */
  int vksift_hip_blur_pair(vksift_hip_Plane src, vksift_hip_Plane dst1, vksift_hip_Plane dst2, const float *taps1, uint32_t ntaps1, const float *taps2,
                           uint32_t ntaps2, uint32_t batch, vksift_hip_stream s);

  /* vksift_hip_blur that also seeds the next octave: next(x, y) = dst(2x+1, 2y+1), the vkCmdBlitImage(NEAREST) of
   * sift_detector.c:1003-1034 for exactly halved sizes, stored from the registers that hold the blurred rows (the separate
   * pass re-reads the whole plane). Bit-identical to vksift_hip_blur + vksift_hip_downsample. Returns -1 without launching
   * anything when the shape or the selected kernel does not cover it: the caller then issues the two separate calls. */
  /* One scale of n <= 8 octaves in ONE launch: dst[i] = blur(src[i]) for every i with the same taps (scales S+1 and S+2 of a detection's octaves
   * feed nothing but the extrema scan, so they can be queued per scale instead of per octave). Returns -1 without launching anything when a
   * plane is outside the strip-march kernel's domain, the texel types differ or the tap count has no multi-octave instantiation (9, 11, 13,
   * 15 taps exist): the caller then takes vksift_hip_blur per plane. Same kernel body: bit-identical to those launches. */
  /* the kernel vksift_hip_blur takes for this shape: 0 generic tiles, 1 two texels per lane (what vksift_hip_blur_multi launches), 2 four texels per lane */
  int vksift_hip_blur_form(vksift_hip_Plane src, vksift_hip_Plane dst, uint32_t ntaps, uint32_t batch);
  int vksift_hip_blur_multi(const vksift_hip_Plane *src, const vksift_hip_Plane *dst, uint32_t n, const float *taps, uint32_t ntaps, uint32_t batch,
                            vksift_hip_stream s);
  int vksift_hip_blur_downsample(vksift_hip_Plane src, vksift_hip_Plane dst, vksift_hip_Plane next, const float *taps, uint32_t ntaps, uint32_t batch,
                                 vksift_hip_stream s);

  /* vkCmdCopyBufferToImage + vkCmdBlitImage(LINEAR, exact 2:1) + the seed blur (sift_detector.c:881-1001 for octave 0) in one
   * pass: dst = blur(upsample2x(src / 255)); the up-sampled plane is never written. Bit-identical to vksift_hip_input_blit
   * followed by vksift_hip_blur. Returns -1 when the shape is not covered (the caller then issues the two separate calls). */
  int vksift_hip_seed_upsampled(const uint8_t *src, uint32_t sw, uint32_t sh, uint64_t src_img_stride, vksift_hip_Plane dst, const float *taps, uint32_t ntaps,
                                uint32_t batch, vksift_hip_stream s);

  /* The same without up-sampling (use_input_upsampling = false: the blit of sift_detector.c:909-916 is a 1:1 copy): dst = blur(src / 255)
   * straight from the u8 images. Bit-identical to vksift_hip_input_blit + vksift_hip_blur; -1 when the shape is not covered. */
  int vksift_hip_seed_direct(const uint8_t *src, uint32_t sw, uint32_t sh, uint64_t src_img_stride, vksift_hip_Plane dst, const float *taps, uint32_t ntaps,
                             uint32_t batch, vksift_hip_stream s);

  /* vkCmdBlitImage(NEAREST) of sift_detector.c:1003-1034: dst(x,y) = src(floor((x+.5)*sw/dw), ...). */
  int vksift_hip_downsample(vksift_hip_Plane src, vksift_hip_Plane dst, uint32_t batch, vksift_hip_stream s);

  /* The scale-space of n_oct consecutive (trailing) octaves in ONE launch, one workgroup per image, planes held in LDS:
   * layers[o * n_layers + l] = layer l of octave o; layer 0 of octave 0 of the run is the input (already in memory), every other plane is
   * written: layer l = blur(layer l - 1, taps[l]) (sift_detector.c:927-1001), layer 0 of the next octave = nearest 2:1 of layer S
   * (:1003-1034). taps: n_layers rows of VKSIFT_HIP_MAX_TAPS. Bit-identical to the vksift_hip_blur / vksift_hip_downsample sequence.
   * Returns -1 without launching anything when the planes are not covered (fp16, width not a multiple of 4, too large for the LDS). */
  int vksift_hip_octave_chain(const vksift_hip_Plane *layers, uint32_t n_oct, uint32_t n_layers, uint32_t S, const float *taps, const uint32_t *ntaps,
                              uint32_t batch, vksift_hip_stream s);

  /* DifferenceOfGaussian.comp:13-17 for one layer of one image: out (dense w x h, fp32) = hi - lo (rounded to binary16 for an
   * fp16 pyramid); hi == NULL: the layer lo itself, widened. Only the debug downloads use it — the detection path never
   * materialises a DoG plane. */
  int vksift_hip_dog_plane(const float *lo, const float *hi, uint32_t w, uint32_t h, uint32_t pitch, uint32_t fp16, float *out_dense, vksift_hip_stream s);

  /* ------------------------------------------------------------------ keypoints */
  typedef struct
  {
    float *gauss;        /* Gaussian layer 0 of image 0 of this octave (S+3 layers; DoG layer s = layer s+1 - layer s) */
    uint32_t fp16;       /* the layers hold binary16 texels (see vksift_hip_Plane); strides stay in texels */
    uint32_t w, h, pitch;
    uint64_t plane_stride; /* floats between layers */
    uint64_t img_stride;   /* floats between images */
    uint32_t S;            /* scales per octave */
    int32_t octave_idx;    /* octave index minus 1 when up-sampling (sift_detector.c:1134) */
    float seed_sigma;
    float dog_threshold;   /* intensity_threshold / S (sift_detector.c:1136) */
    float edge_limit;      /* (edge+1)^2/edge (ExtractKeypoints.comp:203) */
    /* SIFT buffer section of this octave for image 0; image b uses + b*feat_img_stride bytes */
    uint8_t *feats;        /* vksift_Feature records */
    uint64_t feat_img_stride; /* bytes */
    uint32_t cap;          /* section capacity (max_nb_feat) */
    uint32_t *found;       /* per image: counter of this octave, image b at found[b*found_img_stride] */
    uint32_t found_img_stride;
    /* scratch, per image: */
    uint64_t *seg_mask;    /* S*h*nseg words, nseg = ceil(w/64) */
    uint32_t *seg_off;     /* same count */
    uint64_t seg_img_stride; /* elements between images (both arrays); must equal S*h*nseg (contiguous batch) */
    uint32_t *cand_xy;     /* cand_cap packed candidate coordinates */
    uint32_t *cand_flag;   /* cand_cap accept flags */
    uint32_t *cand_n;      /* one counter per image (consecutive) */
    uint64_t cand_img_stride; /* elements between images (cand_xy, cand_flag) */
    uint32_t cand_cap;
    float *ori_ang;        /* cap*VKSIFT_HIP_MAX_ORI floats */
    uint32_t *ori_cnt;     /* cap */
    uint64_t ori_img_stride; /* in keypoints */
    uint32_t max_ori;      /* max_nb_orientation_per_keypoint (0 = unlimited) */
    uint32_t use_vlfeat;
    const float *desc_fp_tab; /* fixed-point multipliers indexed by R/2 (ComputeDescriptors.comp:116-124) */
    uint32_t desc_fp_tab_len;
    uint32_t scan_reverse;    /* dispatch-order hint of the streaming extrema scan, like vksift_hip_Plane::reverse: set when the last
                               * blur launch of the octave ran forward, so that the scan starts on the planes written last */
    uint32_t masks_cleared;   /* the caller has cleared seg_mask for this launch itself (vksift_hip_clear_segment_masks, e.g. on another
                               * stream, off the critical path): vksift_hip_extract_keypoints_multi skips its own fill */
    uint32_t sec_index;       /* sections of the SIFT buffer in front of this octave's (vksift_hip_DenseRows); found[-sec_index .. ] are the
                               * counters of the buffer's sections in order */
  } vksift_hip_OctaveJob;

  /* The matcher's view of a freshly detected SIFT buffer, written by the descriptor launch itself (pack_BufferMemory,
   * sift_memory.c:957-1047, without a pass of its own): feature k of section o is row sum_{j<o} min(found[j], sec_cap[j]) + k of the
   * buffer's dense 128-byte descriptor rows — download order —, its shifted norm beside it, the row total in n[]; rows below 2 of a
   * buffer with fewer features are zero-filled (quirk Q6). Image b of the batch: desc + b*desc_img_stride bytes, norm + b*norm_img_stride
   * words, n[b*n_img_stride]. Every job of the call names its section (sec_index) and all jobs share the section table. */
  typedef struct
  {
    uint8_t *desc;
    uint64_t desc_img_stride;
    uint32_t *norm;
    uint64_t norm_img_stride;
    uint32_t *n;
    uint32_t n_img_stride;
    uint32_t nsec;
    uint32_t sec_cap[16];
    /* feature posting (single-image detections): the same rows as dense 164-byte RECORDS into host-mapped memory — what
     * vksift_hip_pack_features would store there — and the buffer's found_post_n section counters beside them (image b: post +
     * b*post_img_stride bytes, found_post + b*found_post_n words). NULL: off. desc / norm / n may then be NULL as well (rows not wanted). */
    uint8_t *post;
    uint64_t post_img_stride;
    uint32_t *found_post;
    uint32_t found_post_n;
  } vksift_hip_DenseRows;

  /* ExtractKeypoints.comp (sift_detector.c:1106-1189) as a deterministic, atomic-free pipeline: streaming
   * 26-neighbour test -> per-64-pixel-segment candidate ballots -> exclusive scan -> compact candidate list ->
   * dense refinement -> per-image scan + emit in raster order. found[] receives the un-clamped keypoint count.
   * scan_done (may be NULL): recorded right after the streaming scan kernel, the one bandwidth-bound launch of the stage. */
  int vksift_hip_extract_keypoints(const vksift_hip_OctaveJob *job, uint32_t batch, vksift_hip_stream s, vksift_hip_event scan_done);
  /* ComputeOrientation.comp (sift_detector.c:1191-1241): main orientation written in place, extra
   * orientations appended in (keypoint, bin) order; found[] updated. */
  int vksift_hip_orientations(const vksift_hip_OctaveJob *job, uint32_t batch, vksift_hip_stream s);
  /* ComputeDescriptors.comp (sift_detector.c:1243-1259). */
  int vksift_hip_descriptors(const vksift_hip_OctaveJob *job, uint32_t batch, vksift_hip_stream s);
  /* The same three stages for SEVERAL octaves of one detection in one chain of launches: the reference records the dispatches of
   * all octaves of a stage into one command buffer (sift_detector.c:1106-1259); here the workgroups of all octaves share one flat
   * grid per kernel (csrc/hip/multi.h), so a stage costs 1-8 launches whatever the number of octaves, and the small octaves' work
   * fills the gaps of the large one's instead of trickling through launches of their own. jobs[0..n_jobs): same S, same texel
   * type, same batch; results are identical to calling the single-octave form once per job. scan_done as above (all octaves). */
  int vksift_hip_extract_keypoints_multi(const vksift_hip_OctaveJob *jobs, uint32_t n_jobs, uint32_t batch, vksift_hip_stream s, vksift_hip_event scan_done);
  /* the clear of the candidate-ballot masks that vksift_hip_extract_keypoints_multi starts with, on its own (see masks_cleared) */
  int vksift_hip_clear_segment_masks(const vksift_hip_OctaveJob *jobs, uint32_t n_jobs, uint32_t batch, vksift_hip_stream s);
  /* test entry (tests/test_gpu_descriptor_ranges.py): the in-range forms of sqrtf, '/' and x / 2 pi that the orientation and descriptor kernels
   * use (features.hip: sqrt_inrange, div_inrange, div_2pi_inrange) against the compiler's general forms on n pseudo-random operands
   * inside their ranges, incl. the range ends; *d_mismatches (device memory, one word) = results that differ in any bit */
  int vksift_hip_selftest_inrange(uint32_t n, uint32_t seed, uint32_t *d_mismatches, vksift_hip_stream s);
  int vksift_hip_orientations_multi(const vksift_hip_OctaveJob *jobs, uint32_t n_jobs, uint32_t batch, vksift_hip_stream s);
  int vksift_hip_descriptors_multi(const vksift_hip_OctaveJob *jobs, uint32_t n_jobs, uint32_t batch, vksift_hip_stream s);
  /* ... and the dense matcher rows of every buffer with them (dense == NULL: as above). The counters of ALL sections must be final:
   * queue it behind vksift_hip_orientations_multi of every octave of the detection. */
  int vksift_hip_descriptors_multi_dense(const vksift_hip_OctaveJob *jobs, uint32_t n_jobs, uint32_t batch, const vksift_hip_DenseRows *dense,
                                         vksift_hip_stream s);

  /* ------------------------------------------------------------------ matcher */
  /* vksift_Feature records (stride 164 B) -> dense 128-byte descriptor rows (16-byte aligned). Replaces the
   * section packing of sift_memory.c:957-1047 as the matcher's input preparation. */
  int vksift_hip_gather_descriptors(const uint8_t *feats, uint32_t n, uint8_t *desc, vksift_hip_stream s);
  /* Get2NearestNeighbors.comp (sift_matcher.c:246-279) on dense descriptor matrices in HBM, as an exact int8
   * MFMA contraction with a fused top-2 epilogue. desc_a: na rows, desc_b: nb >= 2 rows (callers pad, quirk Q6).
   * norm_scratch: scratch_u32 words of scratch, at least vksift_hip_match_scratch_u32(na, nb) (= the macro
   * VKSIFT_HIP_MATCH_SCRATCH_U32; the call returns hipErrorInvalidValue WITHOUT launching anything when it is given less: the
   * requirement grew between ABI versions 4 and 5, and a buffer sized by an old formula must fail loudly, not be overrun).
   * matches: na records of 20 B {idx_a = a_index_base + row, idx_b1,
   * idx_b2, dist1, dist2}; B rows are scanned in index order, so sharding A rows over GPUs (a_index_base =
   * shard offset) gives bit-identical results to a single call. */
  size_t vksift_hip_match_scratch_u32(uint32_t na, uint32_t nb);
  int vksift_hip_match_2nn_desc(const uint8_t *desc_a, uint32_t na, uint32_t a_index_base, const uint8_t *desc_b, uint32_t nb, uint32_t *norm_scratch,
                                size_t scratch_u32, uint8_t *matches, vksift_hip_stream s);
  /* The two halves of vksift_hip_match_2nn_desc, for callers that overlap the pre-pass of A with the arrival of B (the sharded
   * matcher: RCCL all-gather of B): norms[i] = sum over the 128 bytes of (byte - 128)^2; scratch:
   * scratch_u32 >= vksift_hip_match_scratch_u32(na, nb) - 2*na - nb words (checked like above). */
  int vksift_hip_shifted_norms(const uint8_t *desc, uint32_t n, uint32_t *norms, vksift_hip_stream s);
  int vksift_hip_match_2nn_prenormed(const uint8_t *desc_a, const uint32_t *norm_a, uint32_t na, uint32_t a_index_base, const uint8_t *desc_b,
                                     const uint32_t *norm_b, uint32_t nb, uint32_t *scratch, size_t scratch_u32, uint8_t *matches, vksift_hip_stream s);

  /* Cross-check + Lowe ratio over forward (A->B) and optional reverse (B->A, rev != NULL) 2-NN records, the CPU loop of
   * src/examples/test_sift_match.cpp:90-107 / src/perf/perf_common.cpp:123-169: keep record i iff d1/d2 < ratio and (with
   * rev) rev[idx_b1].idx_b1 == i and its own d1/d2 < ratio. n_fwd[slot*n_stride + {0,1}] = {N_A, N_B} on the device.
   * out: per slot 16-byte records {idx_a, idx_b, dist_a_b1, dist_a_b2} in increasing idx_a order, out_n[slot] their number.
   * Strides in bytes. */
  int vksift_hip_filter_matches(const uint8_t *fwd, uint64_t fwd_slot_stride, const uint8_t *rev, uint64_t rev_slot_stride, const uint32_t *n_fwd,
                                uint32_t n_stride, float ratio, uint32_t nslots, uint8_t *out, uint64_t out_slot_stride, uint32_t *out_n, vksift_hip_stream s);

  /* Asynchronous (and batched) matching pipeline used by vksift_matchFeatures / vksift_ext_matchFeaturesBatch — no
   * host round trip for the feature counts.
   * gather_sections: fills the matcher's per-buffer cache entries of the SIFT buffers buf_ids[0..nslots), nslots <= VKSIFT_HIP_GATHER_SLOTS (feats_base +
   * id*buf_stride, counters found_base + id*found_buf_stride; all buffers of one call share the section table): walks up
   * to 16 sections whose stored counts are min(found[o], sec_cap[o]) (or fixed_counts[o] when found_base is NULL), writes
   * the dense descriptor rows in download order to desc + id*desc_stride, their shifted norms to norms + id*norm_stride and
   * the row total to n_out_dev[id*n_stride]; rows below pad_rows_to are zero-filled (quirk Q6). max_rows bounds the launch.
   * match_2nn_async (nslots <= VKSIFT_HIP_MATCH_SLOTS): slot i matches cache entry ids_a[i] against ids_b[i]; it first writes {N_A, N_B} of every slot to
   * n_dev[i*n_slot_stride + 0..1] (read by the kernels, the filter and the host). Strides in bytes for desc/matches and in
   * u32 elements for norms/redo/n. partial_scratch (may be NULL): 5*max_na*VKSIFT_HIP_MATCH_CHUNKS u32 used by the
   * stream-decomposed single-pair kernel (nslots == 1; without it a single pair takes the batch kernels). redo: max_na u32 per slot of row flags for the exact scalar replay.
   * max_na / max_nb: host-side bounds on the row counts of any slot (the counts themselves stay on the device); a batch whose max_nb is within the
   * packed-key kernel's range launches that kernel only — the pruning kernels' grids for larger reference sets are not queued at all. nb_exact: max_nb
   * is the largest N_B itself (every count has reached the host), not a capacity bound: only then are the pruning kernels queued for the slots
   * beyond the packed-key range; with a mere bound the packed-key kernel serves every slot of the batch whatever its N_B (exact for any size: it
   * walks B in super-chunks of 4096 columns; beyond 32 768 rows the pruning kernel is the faster one, which is all the regime was for). */
  /* Download packing for a batch of up to 64 SIFT buffers that share one section table (the buffers of one batched detection):
   * slot i copies the stored records of buffer buf_ids[i] — sections in order, min(found, capacity) each, the order
   * vksift_downloadFeatures returns (sift_memory.c:957-1047, 1160-1196) — as dense 164-byte records to out + out_rows[i] * 164.
   * The host then reads every buffer of the detection with one device-to-host copy instead of one per section and buffer.
   * found_post (or NULL): a host-mapped mirror of found_base; the counters of the packed buffers are stored there as well. */
  int vksift_hip_pack_features(const uint8_t *feats_base, uint64_t buf_stride, const uint32_t *buf_ids, const uint32_t *out_rows, uint32_t nslots, uint32_t nsec,
                               const uint32_t *sec_off, const uint32_t *sec_cap, const uint32_t *found_base, uint32_t found_buf_stride, uint8_t *out,
                               uint32_t max_rows, uint32_t *found_post, vksift_hip_stream s);
  int vksift_hip_gather_sections(const uint8_t *feats_base, uint64_t buf_stride, const uint32_t *buf_ids, uint32_t nslots, uint32_t nsec,
                                 const uint32_t *sec_off, const uint32_t *sec_cap, const uint32_t *fixed_counts, const uint32_t *found_base,
                                 uint32_t found_buf_stride, uint32_t max_rows, uint32_t pad_rows_to, uint8_t *desc, uint64_t desc_stride,
                                 uint32_t *norms, uint64_t norm_stride, uint32_t *n_out_dev, uint32_t n_stride, vksift_hip_stream s);
  int vksift_hip_match_2nn_async(const uint8_t *cache_desc, const uint32_t *cache_norm, const uint32_t *cache_n, const uint32_t *ids_a, const uint32_t *ids_b,
                                 uint32_t max_na, uint32_t max_nb, uint32_t nb_exact, uint32_t *redo, uint32_t *n_dev, uint8_t *matches, uint32_t nslots, uint64_t cache_desc_stride,
                                 uint64_t cache_norm_stride, uint64_t redo_slot_stride, uint64_t match_slot_stride, uint32_t n_slot_stride,
                                 uint32_t *partial_scratch, vksift_hip_stream s);

#ifdef __cplusplus
}
#endif
#endif /* VKSIFT_HIP_H */
