/*
 * vksift_internal.h — private definitions shared by the host translation units behind the vksift_* C API
 * (vksift_api.c, vksift_instance.c, vksift_detect.c, vksift_buffers.c, vksift_match.c, vksift_ext.c).
 * Nothing here is part of the public ABI; every function is hidden from the shared library's export table.
 */
#ifndef VKSIFT_INTERNAL_H
#define VKSIFT_INTERNAL_H

#include "vksift_ext.h"
#include "vksift_hip.h"
#include "vksift_hostmath.h"
#include "vksift_log.h"
#include "vulkansift/vulkansift.h"

#include <assert.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>


#define VKSIFT_INTERNAL __attribute__((visibility("hidden")))

#define LOG_TAG "VulkanSift"

#define VKSIFT_DL_BATCH_MIN 8u
#define VKSIFT_DL_CHUNKS 8u
#define VKSIFT_UP_GROUPS 8u
#define VKSIFT_FORK_MAX_COUNT 4u /* forked scale-space (and the LDS chain): detections of at most this many images */
#define FEAT_BYTES 164u
#define MATCH_BYTES 20u
#define PITCH_ALIGN 64u
#define DESC_FP_TAB_MAX 1024u

typedef struct
{
  bool is_packed;       /* true: one section [0, nb_stored) (after upload / after matching) */
  uint32_t nb_stored;   /* valid when is_packed */
  uint32_t nb_sections; /* octaves of the detection that last filled the buffer */
  uint32_t sec_off[VKSIFT_MAX_OCTAVES]; /* in features */
  uint32_t sec_cap[VKSIFT_MAX_OCTAVES];
  uint32_t in_w, in_h;  /* resolution of that detection */
  uint64_t seq;         /* sequence number of the detection that last filled the buffer (0: filled from the host / never). The host
                         * mirror of its per-octave counters is valid once that detection has completed (seq <= det_done) */
} BufferInfo;

/* One slot per detection in flight. A caller that pipelines (detection N+1 queued while it downloads the results of detection N,
 * the way vksift_isBufferAvailable is meant to be used) waits for the detection that filled the buffer it reads, not for the
 * latest one. The instance stream is in-order: a later detection's completion implies every earlier one's. */
#define VKSIFT_DETECT_RING 4
typedef struct
{
  vksift_hip_event ev;
  uint64_t seq;
  uint32_t first, count; /* SIFT buffers it filled */
} DetectSlot;

typedef struct
{
  uint32_t n_oct;
  uint32_t w[VKSIFT_MAX_OCTAVES], h[VKSIFT_MAX_OCTAVES], pitch[VKSIFT_MAX_OCTAVES];
  uint64_t plane_stride[VKSIFT_MAX_OCTAVES]; /* floats */
  uint64_t gauss_off[VKSIFT_MAX_OCTAVES];    /* floats from the image's pyramid base: S+3 Gaussian planes per octave (no DoG planes) */
  uint64_t img_floats; /* floats used by one image */
  uint64_t seg_off[VKSIFT_MAX_OCTAVES], seg_total;   /* per-octave slices of the segment scratch (elements) */
  uint64_t cand_off[VKSIFT_MAX_OCTAVES], cand_cap[VKSIFT_MAX_OCTAVES], cand_total;
} PyrLayout;

/* HIP-event stage timings of one detection (vksift_ext_setProfiling) */
/* A captured detection launch sequence (hipGraph), valid for one (resolution, batch, first buffer, input pointer) */
#define VKSIFT_GRAPH_CACHE 8
#define VKSIFT_POST_IDLE 16u
typedef struct
{
  vksift_hip_graph exec;
  uint32_t w, h, count, first_buf;
  const uint8_t *d_src;
  bool post;  /* the sequence ends with the feature posting (see h_post) */
  bool dense; /* its descriptor launch writes the buffer's matcher cache entry (captured before the first matching: it does not) */
  uint64_t stamp;
} DetectGraph;

/* device scratch of one set of matching slots (slot i serves pair i of a batched call) */
typedef struct
{
  uint8_t *matches;
  uint32_t *redo, *match_n;
} MatchScratch;

typedef struct
{
  vksift_hip_event ev_t[8];  /* instance stream: start, upload end, pyramid end, extrema end, orientation end, descriptor end, call end */
  vksift_hip_event ev_pt[3]; /* start / end of octave 0's scale-space construction on its own stream (overlapping detections); [2]: end of the last octave's */
  vksift_hip_event ev_scan;  /* octave 0's streaming extrema scan (the kernel that forms the DoG values) has run */
  bool valid, accounted, overlap;
  uint32_t blur_launches, blur_launches_all;
  uint64_t alg_bytes, scan_bytes;
} ProfSet;

struct vksift_Instance_T
{
  vksift_Config cfg;
  void (*error_cb)(vksift_Result);
  int device;
  uint32_t S;
  bool fp16;               /* VKSIFT_PYRAMID_PRECISION_FLOAT16: the scale-space holds IEEE binary16 texels (2 bytes), arithmetic stays fp32 */
  uint32_t max_image_size; /* rounded up to a square, sift_memory.c:644-647 */
  uint32_t max_octaves;
  uint32_t batch_cap;      /* capacity the instance was created with: match slots, and the bound of the vksift_ext_*Batch entries */
  uint32_t det_cap;        /* images one detection launch sequence can take (>= batch_cap): input staging, scale-space, extraction scratch.
                            * Grows when a caller of the plain API turns out to batch (deferred submission, below) */

  /* Deferred submission of vksift_detectFeatures (vksift_detect.c: defer_detect). The reference's caller hands over ONE image per call
   * (vulkansift.c:315-344); a caller that issues several such calls in a row, into consecutive SIFT buffers, before it asks for
   * anything gets them launched as ONE batched detection: the call stages its image in the pinned input block and returns, and the
   * batch is launched by the first call that needs a result (every other entry point: defer_sync) or when it is full. The first
   * detection after an accessor is launched at once unless the caller is known to batch (batch_mode), so detect + read keeps its path. */
  bool defer_enabled;      /* VKSIFT_DEFER (default 1) and sift_buffer_count >= 2 */
  uint32_t defer_max;      /* VKSIFT_DEFER_MAX (default 128): a deferred batch is launched when it holds this many images (or det_cap) */
  uint32_t defer_chunk;    /* VKSIFT_DEFER_CHUNK (default 16; 0: off): ... or this many while the GPU has no detection to work on */
  uint32_t pend_n, pend_first, pend_w, pend_h; /* staged images: SIFT buffers [pend_first, pend_first + pend_n) */
  uint32_t epoch_detects;  /* vksift_detectFeatures calls since the last call of any other entry point */
  bool batch_mode;         /* the last such run held at least two: the next one is deferred from its first call on */
  bool defer_grow;         /* the last deferred batch filled det_cap: double it before the next one is staged */
  uint64_t defer_batches, defer_images; /* statistics (vksift_ext_getDeferredStats) */

  /* blur taps */
  float taps[(VKSIFT_MAX_SCALES + 3) * VKSIFT_MAX_TAPS];
  uint32_t ntaps[VKSIFT_MAX_SCALES + 3];

  /* current scale-space */
  uint32_t cur_w, cur_h, cur_batch;
  PyrLayout lay;

  /* device memory */
  float *d_pyr;            /* pyramid storage of the current detection (= d_pyr_buf[pyr_cur]); texel offsets scale with pyr_texel_bytes() */
  uint32_t place_n;        /* candidate ranges timed by place_pyramid_buffers (0: plain allocation), their rates, the chosen ones */
  float place_gbps[8];
  uint32_t place_chosen[2];
  float *d_pyr_buf[2];     /* the scale-space buffer(s): ONE by default (the overlap gate makes a second unnecessary), two with VKSIFT_PYR_PINGPONG=2 */
  int pyr_cur;
  bool pyr_pingpong;       /* overlap mode: the scale-space of a detection is built on its own stream, beside what is queued behind the previous detection's descriptors */
  uint32_t overlap_min_count; /* ... for detections of at least this many images (1; 8 on an instance whose capacity grew by deferred submission) */
  bool overlap_forced;     /* VKSIFT_PYR_PINGPONG was given: the mode is the caller's */
  uint32_t pyr_nbuf;       /* scale-space buffers of the instance: 1, or 2 with VKSIFT_PYR_PINGPONG=2 (the next scale-space may then start before the previous detection's readers are done) */
  bool pyr_free_valid[2];
  uint64_t pyr_img_stride; /* floats reserved per image */
  uint8_t *d_input, *h_input;
  uint8_t *d_feats;
  uint64_t buf_stride; /* bytes */
  uint32_t *d_found, *h_found;
  uint64_t *d_seg_mask;
  uint32_t *d_seg_off;
  uint64_t seg_cap; /* elements reserved per image */
  uint32_t *d_cand_xy, *d_cand_flag, *d_cand_n;
  uint64_t cand_cap; /* candidates reserved per image */
  float *d_ori_ang;
  uint32_t *d_ori_cnt;
  uint64_t ori_cap; /* keypoints reserved per image */
  float *d_desc_fp;
  uint32_t desc_fp_len;
  uint8_t *d_matches, *h_matches;
  uint32_t *d_redo;        /* per match slot: row flags for the exact scalar replay (k_match_redo) */
  /* per SIFT buffer: the matcher's view of it (dense descriptor rows in download order, shifted norms, row count), filled by a
   * device-side gather when the buffer is first matched after a detection / upload */
  uint8_t *d_cache_desc;
  uint32_t *d_cache_norm, *d_cache_n;
  uint64_t cache_norm_stride; /* u32 elements */
  bool *cache_valid;
  bool *cache_queued;      /* scratch of refresh_match_cache: the buffer is already in the current gather pass */
  uint32_t *d_match_partial; /* partial top-2 lists of the stream-decomposed single-pair matcher (NULL when max_nb <= VKSIFT_HIP_MATCH_SMALL_NA) */
  uint32_t *d_match_n, *h_match_n; /* per match slot: {N_A, N_B, spare, spare} of the last matching pipeline */
  /* filtered matching (vksift_ext_matchFeaturesFiltered): scratch of the reverse (B->A) matching and the survivors; allocated on first use */
  MatchScratch rev;
  /* hipGraph replay of the detection launch sequence (latency of small workloads is launch bound) */
  bool use_graphs;
  uint64_t graph_max_pixels; /* detections of at most this many input pixels (batch total) are replayed from a graph */
  DetectGraph graphs[VKSIFT_GRAPH_CACHE];
  uint64_t graph_stamp;
  uint32_t graph_miss_run; /* consecutive cache misses: a caller that never repeats a key gets no replays, only capture costs */
  uint8_t *d_filtered;
  uint32_t *d_filtered_n, *h_filtered_n;
  uint64_t filtered_slot_stride;
  uint32_t filtered_slots_used;
  uint64_t desc_slot_stride, match_slot_stride; /* bytes */
  uint64_t redo_slot_stride;                    /* u32 elements */
  uint32_t match_slots_used;
  vksift_hip_event ev_staging;      /* host image staging buffer consumed by the H2D copy */
  vksift_hip_event ev_up[VKSIFT_UP_GROUPS]; /* group g of a batch has arrived in d_input */
  bool staging_pending;
  BufferInfo *bufs;

  vksift_hip_stream stream;
  vksift_hip_stream pyr_stream; /* scale-space construction when detections overlap (two pyramid buffers) */
  vksift_hip_stream up_stream;  /* host-to-device copies of the input images */
  vksift_hip_stream dl_stream;  /* result downloads: the accessors have waited for the pipeline that produced what they read; their copies must
                                 * not queue behind a LATER detection already on the instance stream (a caller that pipelines) */
  vksift_hip_event ev_pyr_done;
  vksift_hip_event ev_pyr_free[2]; /* last reader of pyramid buffer i has finished */
  vksift_hip_event ev_desc_start;  /* the previous detection's descriptor stage has been issued up to here: the next scale-space may start */
  bool desc_start_valid;
  vksift_hip_event ev_input_free;  /* the last reader of d_input (seed pass of the most recent detection) has run */
  bool input_free_valid;
  /* small detections (one image): the scales behind scale S of an octave are off the path to the next octave; they run on the
   * (otherwise idle) scale-space stream beside the octaves below — forked per octave, joined in front of the keypoint stages */
  vksift_hip_event ev_fork[VKSIFT_MAX_OCTAVES], ev_join[2];
  vksift_hip_stream side_stream; /* second branch stream (the first is pyr_stream) */
  int fork_streams;              /* branch streams of a forked detection: 1 (two measured no faster in stream order, slower in a graph) */
  bool fork_scales; /* VKSIFT_FORK_SCALES (default 1) */
  uint64_t fork_max_pixels; /* ... for detections of at most this many input pixels (16 Mpx) */
  uint32_t lds_chain_max; /* largest plane (texels) an octave of the chain may have: VKSIFT_LDS_CHAIN_MAX, at most 19200 (the LDS) */
  bool lds_chain_refuse; /* VKSIFT_LDS_CHAIN=refuse: test hook, the chain launch declines and the per-scale launches take over */
  bool lds_chain; /* VKSIFT_LDS_CHAIN (default 1): the trailing octaves that fit the LDS are built by one launch (vksift_hip_octave_chain) */
  bool alt_order; /* always on: launches of a blur chain alternate their dispatch direction (vksift_hip_Plane::reverse) */
  vksift_hip_event ev_match;
  bool match_pending;
  DetectSlot det_ring[VKSIFT_DETECT_RING];
  uint64_t det_seq, det_done; /* last detection issued / highest one known to have completed */
  /* batched download: the first vksift_downloadFeatures() after a detection of VKSIFT_DL_BATCH_MIN images and more packs the
   * features of ALL its buffers on the device and fetches them with one copy into pinned memory; the downloads of the
   * other buffers are host copies out of it (one copy per section and buffer costs 80 us per buffer otherwise) */
  uint8_t *d_dl, *h_dl;
  size_t dl_cap;       /* bytes of each */
  uint32_t *dl_row;    /* sift_buffer_count + 1 row offsets of the cached buffers */
  uint32_t dl_first, dl_count;
  bool dl_valid;
  bool dl_direct;            /* the packed copy of the current detection is fetched by DMA into page-locked destinations: nothing went to h_dl */
  vksift_hip_event dl_ev[VKSIFT_DL_CHUNKS]; /* the packed copy arrives in pieces */
  size_t dl_chunk_end[VKSIFT_DL_CHUNKS];
  uint32_t dl_chunks, dl_chunks_done;
  uint64_t dl_seq;      /* the detection the packed copy belongs to */
  /* Feature posting: a single-image detection ends with the pack kernel storing the dense records of its buffer straight into
   * mapped pinned memory (slot = buffer index & 1), so that vksift_getFeaturesNumber + vksift_downloadFeatures cost ONE host wait
   * and a host copy — instead of wait, pack launch, device-to-host copy, second wait (~60 us of a 0.45 ms detection).
   * Costs bus time when nobody downloads: switched off after VKSIFT_POST_IDLE posted detections in a row that were never
   * fetched, on again by the next download of a single detection. VKSIFT_POST_FEATURES=0 disables. */
  uint8_t *h_post[2];
  size_t post_cap;        /* bytes of each */
  uint64_t post_seq[2];   /* detection whose records the slot holds (0: none) */
  uint32_t post_buf[2];
  bool post_fetched[2];   /* the slot's records were downloaded at least once */
  bool post_enabled, post_on;
  uint32_t post_idle;
  uint64_t dl_hits_seq; /* the detection dl_hits counts for */
  uint32_t dl_hits; /* vksift_downloadFeatures calls on buffers of that detection, before its packed copy exists */
  /* packed download of the records of a batched matching (h_matches: pinned) */
  size_t md_cap, md_pitch;
  bool md_valid;
  bool md_asked, md_direct;  /* (asked once per matching) the caller's destination of this matching's records is page-locked: per-pair DMA, no packed copy */
  uint32_t md_hits;
  bool *match_busy; /* per SIFT buffer: read by the matching pipeline in flight (all pairs of a batched call) */
  uint32_t curr_nb_matches;

  /* profiling */
  bool profiling;
  ProfSet prof[2]; /* two event sets: the host may enqueue one detection ahead of the one being timed */
  int prof_cur;
  vksift_hip_event ev_m[2];
  bool match_timing_valid;
  double acc_ms[8]; /* upload, pyramid (octave 0), extrema stage, orientation, descriptor, total, extrema scan kernel alone, pyramid (all octaves) */
  uint32_t acc_calls;
  uint64_t acc_blur_launches, acc_blur_launches_all, acc_alg_bytes, acc_scan_bytes;
  uint32_t last_blur_launches, last_blur_launches_all;
  uint64_t last_alg_bytes, last_scan_bytes;
  bool device_input_last;
};


#define HIP_CHECK(expr, what)                                                      \
  do                                                                               \
  {                                                                                \
    int _e = (expr);                                                               \
    if (_e != 0)                                                                   \
    {                                                                              \
      logError(LOG_TAG, "%s failed: %s", what, vksift_hip_error_string(_e));       \
      goto gpu_error;                                                              \
    }                                                                              \
  } while (0)

static inline size_t pyr_texel_bytes(const struct vksift_Instance_T *inst) { return inst->fp16 ? 2u : 4u; }
/* address of texel `off` (texels from the start of the pyramid buffer) */
static inline float *pyr_at(const struct vksift_Instance_T *inst, uint64_t off) { return (float *)((uint8_t *)inst->d_pyr + off * pyr_texel_bytes(inst)); }

static inline bool counts_valid(const struct vksift_Instance_T *inst, uint32_t buf) { return inst->bufs[buf].seq <= inst->det_done; }

/* vksift_api.c */
extern VKSIFT_INTERNAL bool vksift_g_loaded;
VKSIFT_INTERNAL bool config_is_valid(const vksift_Config *c);
VKSIFT_INTERNAL void default_error_callback(vksift_Result err);
VKSIFT_INTERNAL bool buffer_idx_valid(vksift_Instance inst, uint32_t idx);
VKSIFT_INTERNAL bool resolution_valid(vksift_Instance inst, uint32_t w, uint32_t h);

/* vksift_instance.c */
VKSIFT_INTERNAL void compute_layout(vksift_Instance inst, uint32_t w, uint32_t h, PyrLayout *L);
VKSIFT_INTERNAL void set_buffer_sections(vksift_Instance inst, uint32_t buf, uint32_t n_oct, uint32_t w, uint32_t h);
VKSIFT_INTERNAL void mark_detect_done(vksift_Instance inst);
VKSIFT_INTERNAL bool detect_running(vksift_Instance inst);
VKSIFT_INTERNAL int wait_detect_seq(vksift_Instance inst, uint64_t seq);
static inline bool counts_valid(const struct vksift_Instance_T *inst, uint32_t buf);
VKSIFT_INTERNAL bool match_running(vksift_Instance inst);
VKSIFT_INTERNAL int wait_all(vksift_Instance inst);
VKSIFT_INTERNAL int grow_image_scratch(vksift_Instance inst, const PyrLayout *L);
VKSIFT_INTERNAL int resize_detect_scratch(vksift_Instance inst, const PyrLayout *L, uint32_t new_cap);
VKSIFT_INTERNAL vksift_hip_Plane plane_at(vksift_Instance inst, uint32_t o, uint64_t base_off, uint32_t layer);

/* vksift_detect.c */
VKSIFT_INTERNAL void flush_deferred(vksift_Instance inst);
/* Every entry point but vksift_detectFeatures starts with this: what was deferred is launched, and the run of detect calls ends */
static inline void defer_sync(vksift_Instance inst)
{
  if (inst->pend_n)
    flush_deferred(inst);
  if (inst->epoch_detects)
  {
    inst->batch_mode = inst->epoch_detects >= 2u;
    inst->epoch_detects = 0;
  }
}
VKSIFT_INTERNAL void account_set(vksift_Instance inst, ProfSet *ps);
VKSIFT_INTERNAL void account_timings(vksift_Instance inst);

/* vksift_buffers.c */
VKSIFT_INTERNAL void wait_for_buffer(vksift_Instance inst, uint32_t buf);
VKSIFT_INTERNAL uint32_t buffer_counts(vksift_Instance inst, uint32_t buf, uint32_t *cnt, bool log_lost);

/* vksift_match.c */
VKSIFT_INTERNAL MatchScratch fwd_scratch(vksift_Instance inst);
VKSIFT_INTERNAL int refresh_match_cache(vksift_Instance inst, const uint32_t *ids, uint32_t count);

#endif
