/*
 * vksift_api.c — the vksift_* C API over the HIP kernel layer (include/vksift_hip.h).
 *
 * Mirrors the behaviour of the reference façade src/vulkansift/vulkansift.c (validation, error
 * callback contract, blocking/asynchronous split) and the bookkeeping part of
 * src/vulkansift/sift_memory.c (octave geometry, SIFT-buffer sections, packed/sectioned state,
 * count read-back), with Vulkan objects replaced by plain HBM allocations:
 *
 *   pyramid   one allocation; per image: for each octave (S+3) Gaussian planes then (S+2) DoG planes,
 *             fp32, row pitch padded to 64 floats (256 B)
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
 */
#include "vksift_ext.h"
#include "vksift_hip.h"
#include "vksift_hostmath.h"
#include "vksift_log.h"
#include "vulkansift/vulkansift.h"

#include <assert.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const char LOG_TAG[] = "VulkanSift";

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
  bool counts_valid;    /* the host mirror of the per-octave counters is up to date */
} BufferInfo;

typedef struct
{
  uint32_t n_oct;
  uint32_t w[VKSIFT_MAX_OCTAVES], h[VKSIFT_MAX_OCTAVES], pitch[VKSIFT_MAX_OCTAVES];
  uint64_t plane_stride[VKSIFT_MAX_OCTAVES]; /* floats */
  uint64_t gauss_off[VKSIFT_MAX_OCTAVES];    /* floats from the image's pyramid base */
  uint64_t dog_off[VKSIFT_MAX_OCTAVES];
  uint64_t img_floats; /* floats used by one image */
  uint64_t seg_off[VKSIFT_MAX_OCTAVES], seg_total;   /* per-octave slices of the segment scratch (elements) */
  uint64_t cand_off[VKSIFT_MAX_OCTAVES], cand_cap[VKSIFT_MAX_OCTAVES], cand_total;
} PyrLayout;

/* HIP-event stage timings of one detection (vksift_ext_setProfiling) */
/* A captured detection launch sequence (hipGraph), valid for one (resolution, batch, first buffer, input pointer) */
#define VKSIFT_GRAPH_CACHE 8
typedef struct
{
  vksift_hip_graph exec;
  uint32_t w, h, count, first_buf;
  const uint8_t *d_src;
  bool top_scale_stale[VKSIFT_MAX_OCTAVES];
  uint64_t stamp;
} DetectGraph;

/* device scratch of one set of matching slots (slot i serves pair i of a batched call) */
typedef struct
{
  uint8_t *desc_a, *desc_b, *matches;
  uint32_t *norms, *match_n;
} MatchScratch;

typedef struct
{
  vksift_hip_event ev_t[8];  /* instance stream: start, upload end, pyramid end, extrema end, orientation end, descriptor end, call end */
  vksift_hip_event ev_pt[2]; /* start / end of octave 0's scale-space construction on its own stream (overlapping detections) */
  bool valid, accounted, overlap;
  uint32_t blur_launches;
  uint64_t alg_bytes;
} ProfSet;

struct vksift_Instance_T
{
  vksift_Config cfg;
  void (*error_cb)(vksift_Result);
  int device;
  uint32_t S;
  uint32_t max_image_size; /* rounded up to a square, sift_memory.c:644-647 */
  uint32_t max_octaves;
  uint32_t batch_cap;

  /* blur taps */
  float taps[(VKSIFT_MAX_SCALES + 3) * VKSIFT_MAX_TAPS];
  uint32_t ntaps[VKSIFT_MAX_SCALES + 3];

  /* current scale-space */
  uint32_t cur_w, cur_h, cur_batch;
  PyrLayout lay;

  /* device memory */
  float *d_pyr;            /* pyramid storage of the current detection (= d_pyr_buf[pyr_cur]) */
  float *d_pyr_buf[2];     /* ping-pong: detection N+1 builds its pyramid while detection N still reads its own */
  int pyr_cur;
  bool pyr_pingpong;
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
  uint8_t *d_desc_a, *d_desc_b, *d_matches, *h_matches;
  uint32_t *d_norms;
  uint32_t *d_match_partial; /* partial top-2 lists of the B-chunked large-N matcher (NULL when max_nb <= 32768) */
  uint32_t *d_match_n, *h_match_n; /* per match slot: {N_A, N_B, spare, spare} of the last matching pipeline */
  /* filtered matching (vksift_ext_matchFeaturesFiltered): scratch of the reverse (B->A) matching and the survivors; allocated on first use */
  MatchScratch rev;
  /* hipGraph replay of the detection launch sequence (latency of small workloads is launch bound) */
  bool use_graphs;
  DetectGraph graphs[VKSIFT_GRAPH_CACHE];
  uint64_t graph_stamp;
  uint8_t *d_filtered;
  uint32_t *d_filtered_n, *h_filtered_n;
  uint64_t filtered_slot_stride;
  uint32_t filtered_slots_used;
  uint64_t desc_slot_stride, match_slot_stride; /* bytes */
  uint64_t norm_slot_stride;                    /* u32 elements */
  uint32_t match_slots_used;
  vksift_hip_event ev_staging;      /* host image staging buffer consumed by the H2D copy */
  bool staging_pending;
  BufferInfo *bufs;

  vksift_hip_stream stream;
  /* octave-parallel execution inside a stage: octave o >= 1 runs on oct_stream[o] (oct_stream[0] == stream), forked from
   * and joined back into the main stream with events, so the latency-bound small octaves overlap the large ones */
  vksift_hip_stream oct_stream[VKSIFT_MAX_OCTAVES];
  vksift_hip_stream pyr_stream[VKSIFT_MAX_OCTAVES]; /* scale-space construction of octave o when detections overlap */
  vksift_hip_event ev_pyr_done[VKSIFT_MAX_OCTAVES];
  vksift_hip_event ev_pyr_free[2]; /* last reader of pyramid buffer i has finished */
  vksift_hip_event ev_desc_start;  /* octave 0 of the previous detection has reached its (compute-bound) descriptor stage */
  bool desc_start_valid;
  int overlap_gate;                /* 0: next pyramid starts as early as possible, 1: not before the previous descriptor stage */
  vksift_hip_event ev_fork[4], ev_join[4][VKSIFT_MAX_OCTAVES], ev_oct_ready[VKSIFT_MAX_OCTAVES];
  bool serial_octaves;
  bool lazy_top_scale;    /* do not store Gaussian scale S+2 (only its DoG layer is consumed); re-created on download */
  bool top_scale_stale[VKSIFT_MAX_OCTAVES];
  bool coarse_after;      /* coarse octaves start after octave 0's pyramid instead of after its scale S */
  bool stage_sync;        /* debug: join all octaves at every stage boundary instead of per-octave pipelines */
  bool use_chain;         /* fused per-octave scale chain (pyramid_fused.hip) available for this tap set */
  uint32_t chain_min_rows; /* octaves shorter than this keep the per-scale kernels (pipeline ramp dominates) */
  vksift_hip_event ev_detect, ev_match;
  bool detect_pending, match_pending;
  uint32_t detect_first_buf, detect_count;
  uint32_t match_a, match_b;
  uint32_t curr_nb_matches;

  /* profiling */
  bool profiling;
  ProfSet prof[2]; /* two event sets: the host may enqueue one detection ahead of the one being timed */
  int prof_cur;
  vksift_hip_event ev_m[2];
  bool match_timing_valid;
  double acc_ms[6];
  uint32_t acc_calls;
  uint64_t acc_blur_launches, acc_alg_bytes;
  uint32_t last_blur_launches;
  uint64_t last_alg_bytes;
  bool device_input_last;
};

static bool g_loaded = false;

/* ------------------------------------------------------------------------------------------------ */
/* defaults + validation (vulkansift.c:31-66, 550-661)                                              */
/* ------------------------------------------------------------------------------------------------ */
static void default_error_callback(vksift_Result err)
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

static bool config_is_valid(const vksift_Config *c)
{
  bool ok = true;
  ok &= cfg_check(c->input_image_max_size >= 1024, "Invalid configuration: input image size must be greater than or equal to 1024");
  ok &= cfg_check(c->sift_buffer_count > 0, "Invalid configuration: number of SIFT buffers must be greater than zero");
  ok &= cfg_check(c->max_nb_sift_per_buffer > 0, "Invalid configuration: number of SIFT features per buffers must be greater than zero");
  ok &= cfg_check(c->nb_scales_per_octave > 0, "Invalid configuration: number of scales per octave must be greater than zero");
  ok &= cfg_check(c->nb_scales_per_octave <= VKSIFT_MAX_SCALES, "Invalid configuration: number of scales per octave is limited to 13 in this build");
  ok &= cfg_check(c->input_image_blur_level >= 0.f, "Invalid configuration: input image blur level cannot be negative");
  ok &= cfg_check(c->seed_scale_sigma >= 0, "Invalid configuration: seed scale blur level cannot be negative");
  ok &= cfg_check(((c->use_input_upsampling ? 2.f : 1.f) * c->input_image_blur_level) <= c->seed_scale_sigma,
                  "Invalid configuration: the input image blur level (2x if upscaling activated) must be less than the seed scale blur level");
  ok &= cfg_check(c->intensity_threshold >= 0.f, "Invalid configuration: the DoG intensity threshold cannot be negative");
  ok &= cfg_check(c->edge_threshold >= 0.f, "Invalid configuration: the DoG edge threshold cannot be negative");
  ok &= cfg_check(c->on_error_callback_function != NULL, "Invalid configuration: the error callback function must not be NULL");
  switch (c->pyramid_precision_mode)
  {
  case VKSIFT_PYRAMID_PRECISION_FLOAT16:
  case VKSIFT_PYRAMID_PRECISION_FLOAT32:
    break;
  default:
    logError(LOG_TAG, "Invalid configuration: invalid scale-space pyramid format precision specified)");
    ok = false;
  }
  return ok;
}

/* Note: the reference tests `idx > count` (vulkansift.c:588), letting idx == count through to an
 * out-of-bounds access; its header and error-handling demo document `idx >= count` as invalid, which
 * is what this build enforces. */
static bool buffer_idx_valid(vksift_Instance inst, uint32_t idx)
{
  if (idx >= inst->cfg.sift_buffer_count)
  {
    logError(LOG_TAG, "Provided target buffer index is (%d) but the number of reserved buffers is (%d).", idx, inst->cfg.sift_buffer_count);
    return false;
  }
  return true;
}

static bool resolution_valid(vksift_Instance inst, uint32_t w, uint32_t h)
{
  uint64_t size = (uint64_t)w * h;
  if (size > inst->max_image_size)
  {
    logError(LOG_TAG, "Provided input image size (%d*%d=%llu) is greater than the configured maximum image size (%d).", w, h, (unsigned long long)size,
             inst->max_image_size);
    return false;
  }
  if (size < 1024u)
  {
    logError(LOG_TAG, "Invalid input image size (%d*%d=%llu). Input image size must be greater than or equal to 1024", w, h, (unsigned long long)size);
    return false;
  }
  return true;
}

/* ------------------------------------------------------------------------------------------------ */
/* runtime life-cycle                                                                               */
/* ------------------------------------------------------------------------------------------------ */
vksift_Result vksift_loadVulkan()
{
  if (g_loaded)
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
  g_loaded = true;
  logInfo(LOG_TAG, "vksift_loadVulkan() success");
  return VKSIFT_SUCCESS;
}

void vksift_unloadVulkan() { g_loaded = false; }

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
    logError(LOG_TAG, "vksift_LogLevel in vksift_setLogLevel() is not handled");
    break;
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* layout helpers                                                                                   */
/* ------------------------------------------------------------------------------------------------ */
static uint32_t round_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }

static void compute_layout(vksift_Instance inst, uint32_t w, uint32_t h, PyrLayout *L)
{
  memset(L, 0, sizeof(*L));
  L->n_oct = vksift_hm_octaves_for(&inst->cfg, inst->max_octaves, w, h, L->w, L->h);
  uint64_t off = 0;
  for (uint32_t o = 0; o < L->n_oct; o++)
  {
    L->pitch[o] = round_up(L->w[o], PITCH_ALIGN);
    L->plane_stride[o] = (uint64_t)L->pitch[o] * L->h[o];
    L->gauss_off[o] = off;
    off += L->plane_stride[o] * (inst->S + 3);
    L->dog_off[o] = off;
    off += L->plane_stride[o] * (inst->S + 2);
  }
  L->img_floats = off;
  uint64_t so = 0, co = 0;
  for (uint32_t o = 0; o < L->n_oct; o++)
  {
    L->seg_off[o] = so;
    so += (uint64_t)inst->S * L->h[o] * ((L->w[o] + 63) / 64);
    L->cand_off[o] = co;
    /* strict 3x3x3 extrema cannot be denser than 1/4 of the texels; 1/8 (after the contrast pre-filter) is reserved,
     * excess candidates of a pathological image are dropped in raster order */
    L->cand_cap[o] = (uint64_t)inst->S * L->w[o] * L->h[o] / 8u + 64u;
    co += L->cand_cap[o];
  }
  L->seg_total = so;
  L->cand_total = co;
}

static void set_buffer_sections(vksift_Instance inst, uint32_t buf, uint32_t n_oct, uint32_t w, uint32_t h)
{
  BufferInfo *b = &inst->bufs[buf];
  memset(b->sec_off, 0, sizeof(b->sec_off));
  memset(b->sec_cap, 0, sizeof(b->sec_cap));
  b->is_packed = false;
  b->nb_stored = 0;
  b->nb_sections = n_oct;
  b->in_w = w;
  b->in_h = h;
  b->counts_valid = false;
  vksift_hm_section_caps(inst->cfg.max_nb_sift_per_buffer, n_oct, b->sec_cap);
  uint32_t off = 0;
  for (uint32_t o = 0; o < n_oct; o++)
  {
    b->sec_off[o] = off;
    off += b->sec_cap[o];
  }
}

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

/* ------------------------------------------------------------------------------------------------ */
/* instance                                                                                         */
/* ------------------------------------------------------------------------------------------------ */
static vksift_Result create_instance(vksift_Instance *instance_ptr, const vksift_Config *config, uint32_t batch_cap)
{
  assert(instance_ptr != NULL);
  assert(*instance_ptr == NULL);
  assert(config != NULL);

  if (!g_loaded)
  {
    logError(LOG_TAG, "vksift_createInstance() failure: GPU runtime not available. vksift_loadVulkan() must be called before using this function.");
    return VKSIFT_VULKAN_ERROR;
  }
  if (!config_is_valid(config))
  {
    logError(LOG_TAG, "vksift_createInstance() failure: Invalid configuration detected.");
    return VKSIFT_INVALID_INPUT_ERROR;
  }
  if (batch_cap == 0 || batch_cap > config->sift_buffer_count)
  {
    logError(LOG_TAG, "vksift_createInstance() failure: batch capacity (%u) must be in [1, sift_buffer_count=%u].", batch_cap, config->sift_buffer_count);
    return VKSIFT_INVALID_INPUT_ERROR;
  }

  vksift_Instance inst = (vksift_Instance)calloc(1, sizeof(struct vksift_Instance_T));
  if (!inst)
    return VKSIFT_VULKAN_ERROR;
  *instance_ptr = inst;
  inst->cfg = *config;
  inst->error_cb = config->on_error_callback_function;
  inst->S = config->nb_scales_per_octave;
  inst->batch_cap = batch_cap;

  int ndev = vksift_hip_device_count();
  int dev = config->gpu_device_index;
  if (dev < 0)
    dev = 0; /* all MI355X of a node are identical: "best" = first (reference scores by type/VRAM, vulkan_device.c:394-494) */
  if (dev >= ndev)
  {
    logError(LOG_TAG, "vksift_createInstance() failure: gpu_device_index %d but only %d device(s) available", dev, ndev);
    vksift_destroyInstance(instance_ptr);
    return VKSIFT_VULKAN_ERROR;
  }
  inst->device = dev;
  if (vksift_hip_set_device(dev) != 0)
  {
    vksift_destroyInstance(instance_ptr);
    return VKSIFT_VULKAN_ERROR;
  }
  if (config->pyramid_precision_mode == VKSIFT_PYRAMID_PRECISION_FLOAT16)
    logWarning(LOG_TAG, "VKSIFT_PYRAMID_PRECISION_FLOAT16 requested: this build keeps the scale-space in fp32 (superset precision).");
  if (config->use_gpu_debug_functions)
    logWarning(LOG_TAG, "use_gpu_debug_functions requested: there is no frame presenter in the HIP build; use rocprofv3 / roctx ranges instead.");

  inst->max_octaves = vksift_hm_max_octaves(config, &inst->max_image_size);
  vksift_hm_blur_taps(config, inst->taps, inst->ntaps);

  /* ---- reserve device memory for the configured maxima (sift_memory.c:133-360 equivalent) ---- */
  uint32_t side = (uint32_t)ceilf(sqrtf((float)config->input_image_max_size));
  PyrLayout L;
  compute_layout(inst, side, side, &L);
  /* Non-square images of the same area need a little more because of the row-pitch padding: keep slack. */
  inst->pyr_img_stride = L.img_floats + L.img_floats / 4 + 4096;
  inst->seg_cap = L.seg_total + L.seg_total / 4 + 1024;
  inst->cand_cap = L.cand_total + L.cand_total / 4 + 4096u;
  uint32_t caps[VKSIFT_MAX_OCTAVES] = {0};
  vksift_hm_section_caps(config->max_nb_sift_per_buffer, 1, caps);
  inst->ori_cap = config->max_nb_sift_per_buffer; /* a single-octave detection gives the largest section */
  inst->buf_stride = ((uint64_t)config->max_nb_sift_per_buffer * FEAT_BYTES + 255u) & ~(uint64_t)255u;

  float fp_tab[DESC_FP_TAB_MAX];
  inst->desc_fp_len = vksift_hm_desc_fp_table(config, fp_tab, DESC_FP_TAB_MAX);

  bool ok = true;
#define ALLOC_D(ptr, bytes) ok = ok && ((ptr = vksift_hip_malloc(bytes)) != NULL)
#define ALLOC_H(ptr, bytes) ok = ok && ((ptr = vksift_hip_host_malloc(bytes)) != NULL)
  {
    /* 1: two pyramid buffers, so that the scale-space construction of detection N+1 may run under the descriptor and
     * matching work of detection N. Off by default: on MI355X two large kernels sharing the CUs each slow down by about
     * what the overlap wins (measured -5 % frames/s), and the second buffer doubles the largest allocation. */
    const char *e = getenv("VKSIFT_PYR_PINGPONG");
    inst->pyr_pingpong = e && e[0] == '1';
  }
  ALLOC_D(inst->d_pyr_buf[0], sizeof(float) * inst->pyr_img_stride * batch_cap);
  if (inst->pyr_pingpong)
    ALLOC_D(inst->d_pyr_buf[1], sizeof(float) * inst->pyr_img_stride * batch_cap);
  inst->d_pyr = inst->d_pyr_buf[0];
  ALLOC_D(inst->d_input, (size_t)inst->max_image_size * batch_cap);
  ALLOC_H(inst->h_input, (size_t)inst->max_image_size * batch_cap);
  ALLOC_D(inst->d_feats, inst->buf_stride * config->sift_buffer_count);
  ALLOC_D(inst->d_found, sizeof(uint32_t) * VKSIFT_MAX_OCTAVES * config->sift_buffer_count);
  ALLOC_H(inst->h_found, sizeof(uint32_t) * VKSIFT_MAX_OCTAVES * config->sift_buffer_count);
  ALLOC_D(inst->d_seg_mask, sizeof(uint64_t) * inst->seg_cap * batch_cap);
  ALLOC_D(inst->d_seg_off, sizeof(uint32_t) * inst->seg_cap * batch_cap);
  ALLOC_D(inst->d_cand_xy, sizeof(uint32_t) * inst->cand_cap * batch_cap);
  ALLOC_D(inst->d_cand_flag, sizeof(uint32_t) * inst->cand_cap * batch_cap);
  ALLOC_D(inst->d_cand_n, sizeof(uint32_t) * batch_cap * VKSIFT_MAX_OCTAVES);
  ALLOC_D(inst->d_ori_ang, sizeof(float) * VKSIFT_HIP_MAX_ORI * inst->ori_cap * batch_cap);
  ALLOC_D(inst->d_ori_cnt, sizeof(uint32_t) * inst->ori_cap * batch_cap);
  ALLOC_D(inst->d_desc_fp, sizeof(float) * DESC_FP_TAB_MAX);
  /* matching scratch: one slot per batch entry (slot 0 serves vksift_matchFeatures) */
  inst->desc_slot_stride = (((uint64_t)config->max_nb_sift_per_buffer * 128u + 256u) + 255u) & ~(uint64_t)255u;
  inst->match_slot_stride = (((uint64_t)config->max_nb_sift_per_buffer * MATCH_BYTES) + 255u) & ~(uint64_t)255u;
  inst->norm_slot_stride = 3u * (uint64_t)config->max_nb_sift_per_buffer + 96u; /* norms of A, norms of B, redo flags */
  ALLOC_D(inst->d_desc_a, inst->desc_slot_stride * batch_cap);
  ALLOC_D(inst->d_desc_b, inst->desc_slot_stride * batch_cap);
  ALLOC_D(inst->d_matches, inst->match_slot_stride * batch_cap);
  ALLOC_D(inst->d_norms, sizeof(uint32_t) * inst->norm_slot_stride * batch_cap);
  ALLOC_D(inst->d_match_n, sizeof(uint32_t) * 4 * batch_cap);
  if (config->max_nb_sift_per_buffer > 32768u)
    ALLOC_D(inst->d_match_partial, sizeof(uint32_t) * (size_t)config->max_nb_sift_per_buffer * 5u * VKSIFT_HIP_MATCH_CHUNKS);
  ALLOC_H(inst->h_match_n, sizeof(uint32_t) * 4 * batch_cap);
  inst->h_matches = NULL;
  inst->bufs = (BufferInfo *)calloc(config->sift_buffer_count, sizeof(BufferInfo));
  ok = ok && inst->bufs != NULL;
  /* All streams at the default priority: a high-priority instance stream with low-priority octave streams was measured
   * 20 % slower on MI355X (11.3k vs 14.1k frames/s). */
  inst->stream = vksift_hip_stream_create();
  inst->oct_stream[0] = inst->stream;
  for (int o = 1; o < VKSIFT_MAX_OCTAVES; o++)
    inst->oct_stream[o] = vksift_hip_stream_create();
  for (int o = 0; o < VKSIFT_MAX_OCTAVES; o++)
  {
    inst->ev_oct_ready[o] = vksift_hip_event_create();
    for (int g = 0; g < 4; g++)
      inst->ev_join[g][o] = vksift_hip_event_create();
  }
  for (int g = 0; g < 4; g++)
    inst->ev_fork[g] = vksift_hip_event_create();
  for (int o = 0; o < VKSIFT_MAX_OCTAVES; o++)
  {
    inst->pyr_stream[o] = vksift_hip_stream_create();
    inst->ev_pyr_done[o] = vksift_hip_event_create();
  }
  inst->ev_desc_start = vksift_hip_event_create();
  {
    const char *e = getenv("VKSIFT_OVERLAP_GATE");
    inst->overlap_gate = e ? atoi(e) : 1;
  }
  for (int i = 0; i < 2; i++)
  {
    inst->ev_pyr_free[i] = vksift_hip_event_create();
  }
  {
    const char *e = getenv("VKSIFT_SERIAL_OCTAVES"); /* debug: everything on the main stream */
    inst->serial_octaves = e && e[0] == '1';
    e = getenv("VKSIFT_LAZY_TOP"); /* 0: always store the last Gaussian scale of every octave */
    inst->lazy_top_scale = !(e && e[0] == '0');
    /* Octave 1 starts after octave 0's last blur instead of right after its scale S: the two bandwidth-bound pyramids no
     * longer compete (octave 0 runs 5-8 % faster alone; frames/s unchanged within noise), and the coarse octaves then
     * overlap octave 0's extraction and descriptor stages. VKSIFT_COARSE_AFTER=0 restores the earliest possible start. */
    e = getenv("VKSIFT_COARSE_AFTER");
    inst->coarse_after = !(e && e[0] == '0');
    /* 1: capture the detection launch sequence in a hipGraph and replay it. Off by default: measured on MI355X / ROCm 7.2 it
     * buys 6 % on one 640x480 image (0.78 vs 0.83 ms) and loses 10 % from 1536x1024 up (the graph runs the per-octave
     * branches less concurrently than the streams do). */
    e = getenv("VKSIFT_GRAPH");
    inst->use_graphs = e && e[0] == '1';
    e = getenv("VKSIFT_STAGE_SYNC");
    inst->stage_sync = e && e[0] == '1';
    /* 1 selects the experimental fused scale-chain kernel (pyramid_fused.hip): bit-identical, but measured slower than the
     * per-scale kernels on MI355X (VALU-issue bound) -> off by default */
    e = getenv("VKSIFT_CHAIN");
    inst->use_chain = (e && e[0] == '1') && vksift_hip_octave_chain_supported(inst->ntaps, inst->S);
    e = getenv("VKSIFT_CHAIN_MIN_ROWS");
    inst->chain_min_rows = e ? (uint32_t)atoi(e) : 200u;
  }
  inst->ev_detect = vksift_hip_event_create();
  inst->ev_match = vksift_hip_event_create();
  inst->ev_staging = vksift_hip_event_create();
  for (int i = 0; i < 8; i++)
  {
    inst->prof[0].ev_t[i] = vksift_hip_event_create();
    inst->prof[1].ev_t[i] = vksift_hip_event_create();
  }
  for (int i = 0; i < 2; i++)
  {
    inst->prof[0].ev_pt[i] = vksift_hip_event_create();
    inst->prof[1].ev_pt[i] = vksift_hip_event_create();
  }
  inst->ev_m[0] = vksift_hip_event_create();
  inst->ev_m[1] = vksift_hip_event_create();
  ok = ok && inst->stream && inst->ev_detect && inst->ev_match;
  if (!ok)
  {
    logError(LOG_TAG, "vksift_createInstance() failure: Failed to setup the required memory objects");
    vksift_destroyInstance(instance_ptr);
    return VKSIFT_VULKAN_ERROR;
  }
  memset(inst->h_found, 0, sizeof(uint32_t) * VKSIFT_MAX_OCTAVES * config->sift_buffer_count);
  memset(inst->h_match_n, 0, sizeof(uint32_t) * 4 * batch_cap);
  if (vksift_hip_memset(inst->d_found, 0, sizeof(uint32_t) * VKSIFT_MAX_OCTAVES * config->sift_buffer_count, inst->stream) != 0 ||
      vksift_hip_memcpy_h2d(inst->d_desc_fp, fp_tab, sizeof(float) * inst->desc_fp_len, inst->stream) != 0 || vksift_hip_stream_sync(inst->stream) != 0)
  {
    logError(LOG_TAG, "vksift_createInstance() failure: device initialisation failed");
    vksift_destroyInstance(instance_ptr);
    return VKSIFT_VULKAN_ERROR;
  }

  /* default scale-space = the square of maximal area, like the reference (sift_memory.c:644-662) */
  inst->cur_w = side;
  inst->cur_h = side;
  inst->cur_batch = 1;
  inst->lay = L;
  for (uint32_t b = 0; b < config->sift_buffer_count; b++)
  {
    set_buffer_sections(inst, b, L.n_oct, side, side);
    inst->bufs[b].counts_valid = true;
  }

  logInfo(LOG_TAG, "vksift_createInstance() success");
  return VKSIFT_SUCCESS;
}

vksift_Result vksift_createInstance(vksift_Instance *instance_ptr, const vksift_Config *config) { return create_instance(instance_ptr, config, 1); }

vksift_Result vksift_ext_createInstanceBatched(vksift_Instance *instance_ptr, const vksift_Config *config, uint32_t batch_capacity)
{
  return create_instance(instance_ptr, config, batch_capacity);
}

void vksift_destroyInstance(vksift_Instance *instance_ptr)
{
  assert(instance_ptr != NULL);
  assert(*instance_ptr != NULL);
  vksift_Instance inst = *instance_ptr;
  vksift_hip_set_device(inst->device);
  for (int o = 0; o < VKSIFT_MAX_OCTAVES; o++)
  {
    if (o > 0 && inst->oct_stream[o])
      vksift_hip_stream_sync(inst->oct_stream[o]);
    if (inst->pyr_stream[o])
      vksift_hip_stream_sync(inst->pyr_stream[o]);
  }
  if (inst->stream)
    vksift_hip_stream_sync(inst->stream);
  vksift_hip_free(inst->d_pyr_buf[0]);
  vksift_hip_free(inst->d_pyr_buf[1]);
  vksift_hip_free(inst->d_input);
  vksift_hip_host_free(inst->h_input);
  vksift_hip_free(inst->d_feats);
  vksift_hip_free(inst->d_found);
  vksift_hip_host_free(inst->h_found);
  vksift_hip_free(inst->d_seg_mask);
  vksift_hip_free(inst->d_seg_off);
  vksift_hip_free(inst->d_cand_xy);
  vksift_hip_free(inst->d_cand_flag);
  vksift_hip_free(inst->d_cand_n);
  vksift_hip_free(inst->d_ori_ang);
  vksift_hip_free(inst->d_ori_cnt);
  vksift_hip_free(inst->d_desc_fp);
  vksift_hip_free(inst->d_desc_a);
  vksift_hip_free(inst->d_desc_b);
  vksift_hip_free(inst->d_matches);
  vksift_hip_free(inst->d_norms);
  vksift_hip_free(inst->d_match_n);
  vksift_hip_free(inst->d_match_partial);
  for (int i = 0; i < VKSIFT_GRAPH_CACHE; i++)
    vksift_hip_graph_destroy(inst->graphs[i].exec);
  vksift_hip_free(inst->rev.desc_a);
  vksift_hip_free(inst->rev.desc_b);
  vksift_hip_free(inst->rev.matches);
  vksift_hip_free(inst->rev.norms);
  vksift_hip_free(inst->rev.match_n);
  vksift_hip_free(inst->d_filtered);
  vksift_hip_free(inst->d_filtered_n);
  vksift_hip_host_free(inst->h_filtered_n);
  vksift_hip_host_free(inst->h_match_n);
  vksift_hip_host_free(inst->h_matches);
  free(inst->bufs);
  vksift_hip_event_destroy(inst->ev_detect);
  vksift_hip_event_destroy(inst->ev_match);
  vksift_hip_event_destroy(inst->ev_staging);
  for (int i = 0; i < 8; i++)
  {
    vksift_hip_event_destroy(inst->prof[0].ev_t[i]);
    vksift_hip_event_destroy(inst->prof[1].ev_t[i]);
  }
  vksift_hip_event_destroy(inst->ev_m[0]);
  vksift_hip_event_destroy(inst->ev_m[1]);
  for (int o = 1; o < VKSIFT_MAX_OCTAVES; o++)
    vksift_hip_stream_destroy(inst->oct_stream[o]);
  for (int o = 0; o < VKSIFT_MAX_OCTAVES; o++)
  {
    vksift_hip_stream_destroy(inst->pyr_stream[o]);
    vksift_hip_event_destroy(inst->ev_pyr_done[o]);
  }
  vksift_hip_event_destroy(inst->ev_desc_start);
  for (int i = 0; i < 2; i++)
  {
    vksift_hip_event_destroy(inst->ev_pyr_free[i]);
    vksift_hip_event_destroy(inst->prof[0].ev_pt[i]);
    vksift_hip_event_destroy(inst->prof[1].ev_pt[i]);
  }
  for (int o = 0; o < VKSIFT_MAX_OCTAVES; o++)
  {
    vksift_hip_event_destroy(inst->ev_oct_ready[o]);
    for (int g = 0; g < 4; g++)
      vksift_hip_event_destroy(inst->ev_join[g][o]);
  }
  for (int g = 0; g < 4; g++)
    vksift_hip_event_destroy(inst->ev_fork[g]);
  vksift_hip_stream_destroy(inst->stream);
  free(inst);
  *instance_ptr = NULL;
}

/* ------------------------------------------------------------------------------------------------ */
/* synchronisation helpers (fences of the reference)                                                */
/* ------------------------------------------------------------------------------------------------ */
/* The stream is in-order: once the most recent detection has completed, every earlier one has too, so all the
 * host-side counter mirrors are valid. */
static void mark_detect_done(vksift_Instance inst)
{
  inst->detect_pending = false;
  for (uint32_t b = 0; b < inst->cfg.sift_buffer_count; b++)
    inst->bufs[b].counts_valid = true;
}
static bool detect_running(vksift_Instance inst)
{
  if (!inst->detect_pending)
    return false;
  if (vksift_hip_event_busy(inst->ev_detect) == 1)
    return true;
  mark_detect_done(inst);
  return false;
}
static bool match_running(vksift_Instance inst)
{
  if (!inst->match_pending)
    return false;
  if (vksift_hip_event_busy(inst->ev_match) == 1)
    return true;
  inst->match_pending = false;
  return false;
}
static int wait_all(vksift_Instance inst)
{
  vksift_hip_set_device(inst->device);
  int e = vksift_hip_stream_sync(inst->stream);
  mark_detect_done(inst);
  inst->match_pending = false;
  return e;
}

bool vksift_isBufferAvailable(vksift_Instance instance, const uint32_t gpu_buffer_id)
{
  vksift_hip_set_device(instance->device);
  if (gpu_buffer_id >= instance->cfg.sift_buffer_count)
    return true;
  if (detect_running(instance) && !instance->bufs[gpu_buffer_id].counts_valid)
    return false;
  if (match_running(instance) && (gpu_buffer_id == instance->match_a || gpu_buffer_id == instance->match_b))
    return false;
  return true;
}

/* ------------------------------------------------------------------------------------------------ */
/* detection (vulkansift.c:315-344 + sift_memory.c:891-955 + sift_detector.c:1313-1410,1462-1542)   */
/* ------------------------------------------------------------------------------------------------ */
static vksift_hip_Plane plane_at(vksift_Instance inst, uint32_t o, uint64_t base_off, uint32_t layer)
{
  vksift_hip_Plane p;
  p.base = inst->d_pyr + base_off + (uint64_t)layer * inst->lay.plane_stride[o];
  p.w = inst->lay.w[o];
  p.h = inst->lay.h[o];
  p.pitch = inst->lay.pitch[o];
  p.img_stride = inst->pyr_img_stride;
  return p;
}

static uint64_t algorithmic_pyramid_bytes(vksift_Instance inst, uint32_t w, uint32_t h, uint32_t nb_octaves)
{
  /* (S+3 Gaussian writes + S+2 blur reads + S+2 DoG writes) * 4 B per octave pixel,
   * plus on octave 0: input read (1 B/px of input), up-sample plane write and seed-blur read (4 B each). */
  const PyrLayout *L = &inst->lay;
  uint64_t bytes = 0;
  for (uint32_t o = 0; o < L->n_oct && o < nb_octaves; o++)
    bytes += (uint64_t)L->w[o] * L->h[o] * 4u * ((inst->S + 3) + (inst->S + 2) + (inst->S + 2));
  bytes += (uint64_t)w * h + (uint64_t)L->w[0] * L->h[0] * 8u;
  return bytes;
}

/* fold the (completed) event timings of a detect call into the running sums */
static void account_set(vksift_Instance inst, ProfSet *ps)
{
  if (!inst->profiling || !ps->valid || ps->accounted)
    return;
  vksift_hip_event *e = ps->ev_t;
  inst->acc_ms[0] += vksift_hip_event_elapsed_ms(e[0], e[1]);
  inst->acc_ms[1] += ps->overlap ? vksift_hip_event_elapsed_ms(ps->ev_pt[0], ps->ev_pt[1]) : vksift_hip_event_elapsed_ms(e[1], e[2]);
  inst->acc_ms[2] += vksift_hip_event_elapsed_ms(e[2], e[3]);
  inst->acc_ms[3] += vksift_hip_event_elapsed_ms(e[3], e[4]);
  inst->acc_ms[4] += vksift_hip_event_elapsed_ms(e[4], e[5]);
  inst->acc_ms[5] += vksift_hip_event_elapsed_ms(e[0], e[6]);
  inst->acc_calls++;
  inst->acc_blur_launches += ps->blur_launches;
  inst->acc_alg_bytes += ps->alg_bytes;
  ps->accounted = true;
}

/* all detections have completed (caller waited): account both event sets, oldest first */
static void account_timings(vksift_Instance inst)
{
  account_set(inst, &inst->prof[inst->prof_cur ^ 1]);
  account_set(inst, &inst->prof[inst->prof_cur]);
}

static void detect_impl(vksift_Instance inst, const uint8_t *const *images, const uint8_t *d_images, uint32_t count, uint32_t w, uint32_t h,
                        uint32_t first_buf, const char *fn)
{
  /* declared first: the error path below may be entered before the enqueue section */
  vksift_hip_stream st = inst->stream;
  DetectGraph *dg = NULL;
  bool capturing = false;

  bool valid = count >= 1 && count <= inst->batch_cap && buffer_idx_valid(inst, first_buf) && buffer_idx_valid(inst, first_buf + count - 1) &&
               resolution_valid(inst, w, h);
  if (valid)
  {
    uint32_t shortest = w < h ? w : h;
    if (shortest < 16)
    {
      logError(LOG_TAG, "Input image %ux%u is too small to build a single octave.", w, h);
      valid = false;
    }
  }
  if (!valid)
  {
    logError(LOG_TAG, "%s error: invalid input.", fn);
    inst->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return;
  }

  /* The reference makes a new pipeline wait on the host for the running ones (vulkansift.c:326-327) because its
   * command buffers and staging memory are single-instanced. Here the instance's HIP stream is in-order, so GPU
   * work is already serialised; the host only has to wait for the resources it is about to overwrite: the pinned
   * image staging buffer, and (when profiling) the event set of the previous detection. */
  if (inst->staging_pending && images)
  {
    HIP_CHECK(vksift_hip_event_sync(inst->ev_staging), "staging synchronisation");
    inst->staging_pending = false;
  }
  ProfSet *PS = &inst->prof[inst->prof_cur];
  if (inst->profiling)
  {
    /* recycle the event set of the detection before the previous one: the host never waits for the call it just queued */
    inst->prof_cur ^= 1;
    PS = &inst->prof[inst->prof_cur];
    if (PS->valid && !PS->accounted)
    {
      HIP_CHECK(vksift_hip_event_sync(PS->ev_t[6]), "profiling synchronisation");
      account_set(inst, PS);
    }
    PS->valid = false;
  }

  if (inst->cur_w != w || inst->cur_h != h)
  {
    PyrLayout L;
    compute_layout(inst, w, h, &L);
    if (L.n_oct == 0 || L.img_floats > inst->pyr_img_stride || L.seg_total > inst->seg_cap || L.cand_total > inst->cand_cap)
    {
      logError(LOG_TAG, "Failed to fit the scale-space of a %ux%u image in the memory reserved for input_image_max_size", w, h);
      goto gpu_error;
    }
    inst->lay = L;
    inst->cur_w = w;
    inst->cur_h = h;
  }
  inst->cur_batch = count;
  const PyrLayout *L = &inst->lay;
  for (uint32_t i = 0; i < count; i++)
    set_buffer_sections(inst, first_buf + i, L->n_oct, w, h);

  const bool prof = inst->profiling;
  const size_t img_bytes = (size_t)w * h;
  if (prof)
    vksift_hip_event_record(PS->ev_t[0], st);

  /* Overlapping detections: with two pyramid buffers the scale-space construction of this call does not depend on
   * anything the previous call (or a matching still in flight) reads or writes, so it runs on its own streams, ordered
   * only behind the last reader of the pyramid buffer it recycles; everything that touches the SIFT buffers and the
   * extraction scratch stays in instance-stream order. A caller that issues detect(N+1) right after match(N) gets the
   * bandwidth-bound pyramid of N+1 under the compute-bound descriptor and matching work of N. */
  const bool overlap = inst->pyr_pingpong && !inst->serial_octaves && !inst->stage_sync && L->n_oct > 1;
  PS->overlap = overlap;
  if (overlap)
  {
    inst->pyr_cur ^= 1;
    inst->d_pyr = inst->d_pyr_buf[inst->pyr_cur];
    if (inst->pyr_free_valid[inst->pyr_cur])
      HIP_CHECK(vksift_hip_stream_wait_event(inst->pyr_stream[0], inst->ev_pyr_free[inst->pyr_cur]), "pyramid buffer recycle");
    /* Pair the bandwidth-bound pyramid with the compute-bound tail of the previous detection (descriptors, matching),
     * not with its equally bandwidth-bound extraction stage. */
    if (inst->overlap_gate && inst->desc_start_valid)
      HIP_CHECK(vksift_hip_stream_wait_event(inst->pyr_stream[0], inst->ev_desc_start), "overlap gate");
  }

  /* stage the images; the caller may reuse its memory as soon as we return (sift_memory.c:943) */
  const uint8_t *d_src = d_images;
  if (images)
  {
    for (uint32_t i = 0; i < count; i++)
      memcpy(inst->h_input + i * img_bytes, images[i], img_bytes);
    d_src = inst->d_input;
  }

  /* hipGraph replay: the launch sequence below depends only on (resolution, batch, first buffer, input pointer) — counts
   * and candidate lists live on the device — so it is captured once per such key and replayed with a single launch.
   * One 640x480 detection is ~100 short kernels on 5 streams: launch bound without it. Host-visible events (staging,
   * completion, profiling) stay outside the captured region. */
  if (inst->use_graphs && !prof && !overlap)
  {
    DetectGraph *victim = &inst->graphs[0];
    for (int i = 0; i < VKSIFT_GRAPH_CACHE; i++)
    {
      DetectGraph *g = &inst->graphs[i];
      if (g->exec && g->w == w && g->h == h && g->count == count && g->first_buf == first_buf && g->d_src == d_src)
      {
        dg = g;
        break;
      }
      if (g->stamp < victim->stamp)
        victim = g;
    }
    if (dg)
    {
      dg->stamp = ++inst->graph_stamp;
      HIP_CHECK(vksift_hip_graph_launch(dg->exec, st), "detection graph launch");
      memcpy(inst->top_scale_stale, dg->top_scale_stale, sizeof(inst->top_scale_stale));
      if (images)
      {
        HIP_CHECK(vksift_hip_event_record(inst->ev_staging, st), "event record");
        inst->staging_pending = true;
      }
      inst->device_input_last = images == NULL;
      goto enqueued;
    }
    dg = victim;
    vksift_hip_graph_destroy(dg->exec);
    memset(dg, 0, sizeof(*dg));
    if (vksift_hip_capture_begin(st) == 0)
      capturing = true;
    else
      dg = NULL;
  }
  if (images)
  {
    vksift_hip_stream s_up = overlap ? inst->pyr_stream[0] : st; /* behind the previous reader of d_input either way */
    HIP_CHECK(vksift_hip_memcpy_h2d(inst->d_input, inst->h_input, img_bytes * count, s_up), "image upload");
    if (!capturing)
    {
      HIP_CHECK(vksift_hip_event_record(inst->ev_staging, s_up), "event record");
      inst->staging_pending = true;
    }
  }
  inst->device_input_last = images == NULL;
  if (prof)
    vksift_hip_event_record(PS->ev_t[1], st);

  /* recClearBufferDataCmds (sift_detector.c:1081-1104) */
  HIP_CHECK(vksift_hip_memset(inst->d_found + (size_t)first_buf * VKSIFT_MAX_OCTAVES, 0, sizeof(uint32_t) * VKSIFT_MAX_OCTAVES * count, st),
            "counter reset");

  /* ---- keypoints ---- */
  vksift_hip_OctaveJob jobs[VKSIFT_MAX_OCTAVES];
  const BufferInfo *b0 = &inst->bufs[first_buf];
  for (uint32_t o = 0; o < L->n_oct; o++)
  {
    vksift_hip_OctaveJob *j = &jobs[o];
    memset(j, 0, sizeof(*j));
    j->dog = inst->d_pyr + L->dog_off[o];
    j->gauss = inst->d_pyr + L->gauss_off[o];
    j->w = L->w[o], j->h = L->h[o], j->pitch = L->pitch[o];
    j->plane_stride = L->plane_stride[o];
    j->img_stride = inst->pyr_img_stride;
    j->S = inst->S;
    j->octave_idx = (int32_t)o - (inst->cfg.use_input_upsampling ? 1 : 0);
    j->seed_sigma = inst->cfg.seed_scale_sigma;
    j->dog_threshold = inst->cfg.intensity_threshold / (float)inst->S;
    j->edge_limit = ((inst->cfg.edge_threshold + 1.f) * (inst->cfg.edge_threshold + 1.f)) / inst->cfg.edge_threshold;
    j->feats = inst->d_feats + (uint64_t)first_buf * inst->buf_stride + (uint64_t)b0->sec_off[o] * FEAT_BYTES;
    j->feat_img_stride = inst->buf_stride;
    j->cap = b0->sec_cap[o];
    j->found = inst->d_found + (size_t)first_buf * VKSIFT_MAX_OCTAVES + o;
    j->found_img_stride = VKSIFT_MAX_OCTAVES;
    /* segment scratch is octave-major: [octave][image][segment], so one octave's masks of the whole batch are contiguous */
    {
      const uint64_t nsegs_o = (uint64_t)inst->S * L->h[o] * ((L->w[o] + 63) / 64);
      j->seg_mask = inst->d_seg_mask + L->seg_off[o] * count;
      j->seg_off = inst->d_seg_off + L->seg_off[o] * count;
      j->seg_img_stride = nsegs_o;
    }
    j->cand_xy = inst->d_cand_xy + L->cand_off[o];
    j->cand_flag = inst->d_cand_flag + L->cand_off[o];
    j->cand_n = inst->d_cand_n + (size_t)o * inst->batch_cap;
    j->cand_img_stride = inst->cand_cap;
    j->cand_cap = (uint32_t)L->cand_cap[o];
    j->ori_ang = inst->d_ori_ang + (size_t)b0->sec_off[o] * VKSIFT_HIP_MAX_ORI;
    j->ori_cnt = inst->d_ori_cnt + b0->sec_off[o];
    j->ori_img_stride = inst->ori_cap;
    j->max_ori = inst->cfg.max_nb_orientation_per_keypoint;
    j->use_vlfeat = inst->cfg.descriptor_format == VKSIFT_DESCRIPTOR_FORMAT_VLFEAT ? 1u : 0u;
    j->desc_fp_tab = inst->d_desc_fp;
    j->desc_fp_tab_len = inst->desc_fp_len;
  }

  /* ---- scale-space construction + DoG, keypoints, orientations, descriptors ----
   * Octave o+1 only needs scale S of octave o, and everything after the pyramid is per octave (own SIFT-buffer section,
   * own scratch). Two schedules:
   *   pipelined (default): octave 0 runs on the instance stream, every other octave runs its whole chain
   *     pyramid -> ExtractKeypoints -> ComputeOrientation -> ComputeDescriptors on its own stream, started by the
   *     event "scale S of the previous octave is ready"; the instance stream joins them before the count read-back.
   *     The latency-bound launch chains of the coarse octaves hide behind the bandwidth-bound work of the fine ones.
   *     Profiling events then time octave 0's stages (the other octaves overlap them).
   *   stage-synchronous (VKSIFT_STAGE_SYNC=1): fork per octave inside each stage, join at every stage boundary.
   *   serial (VKSIFT_SERIAL_OCTAVES=1): everything on the instance stream. */
  uint32_t nblur = 0;
  const vksift_hip_Plane no_dog = {NULL, 0, 0, 0, 0};
  const bool par = !inst->serial_octaves && L->n_oct > 1;
  const bool pipelined = par && !inst->stage_sync;
  bool g0_done = false; /* plane 0 of the current octave was already written by the previous octave's chain kernel */
  if (par)
    HIP_CHECK(vksift_hip_event_record(inst->ev_fork[0], st), "event record");
  for (uint32_t o = 0; o < L->n_oct; o++)
  {
    vksift_hip_stream so = st;
    if (par && (o > 0 || !pipelined))
    {
      so = inst->oct_stream[o];
      HIP_CHECK(vksift_hip_stream_wait_event(so, inst->ev_fork[0]), "octave fork");
    }
    /* sp: stream of this octave's scale-space construction; so: stream of its keypoint stages */
    vksift_hip_stream sp = overlap ? inst->pyr_stream[o] : so;
    if (overlap && o > 0 && inst->pyr_free_valid[inst->pyr_cur])
      HIP_CHECK(vksift_hip_stream_wait_event(sp, inst->ev_pyr_free[inst->pyr_cur]), "pyramid buffer recycle");
    vksift_hip_range_push("Scale space construction");
    uint32_t nb_o = 0;
    if (o == 0)
    {
      if (overlap && prof)
        vksift_hip_event_record(PS->ev_pt[0], sp);
      /* blit into the (still unused) layer-1 slot, then seed-blur it into layer 0 */
      int fused = -1;
      if (L->w[0] == 2 * w && L->h[0] == 2 * h)
      {
        fused = vksift_hip_seed_upsampled(d_src, w, h, img_bytes, plane_at(inst, 0, L->gauss_off[0], 0), &inst->taps[0], inst->ntaps[0], count, sp);
        if (fused > 0)
          HIP_CHECK(fused, "fused up-sampling + seed blur");
      }
      if (fused < 0)
      {
        vksift_hip_Plane tmp = plane_at(inst, 0, L->gauss_off[0], 1);
        HIP_CHECK(vksift_hip_input_blit(d_src, w, h, img_bytes, tmp, count, sp), "input blit");
        HIP_CHECK(vksift_hip_blur(tmp, plane_at(inst, 0, L->gauss_off[0], 0), no_dog, &inst->taps[0], inst->ntaps[0], count, sp), "seed blur");
      }
      nb_o++;
    }
    else
    {
      if (par)
        HIP_CHECK(vksift_hip_stream_wait_event(sp, inst->ev_oct_ready[o - 1]), "octave dependency");
      if (!g0_done)
        HIP_CHECK(vksift_hip_downsample(plane_at(inst, o - 1, L->gauss_off[o - 1], inst->S), plane_at(inst, o, L->gauss_off[o], 0), count, sp), "downsample");
    }
    g0_done = false;
    inst->top_scale_stale[o] = false;
    if (inst->use_chain && L->h[o] >= inst->chain_min_rows)
    {
      /* one launch for scales 1..S+2 and all DoG layers; it also seeds the next octave when the sizes are exactly 2:1 */
      vksift_hip_Plane next = {NULL, 0, 0, 0, 0};
      if (o + 1 < L->n_oct && L->w[o + 1] * 2 == L->w[o] && L->h[o + 1] * 2 == L->h[o])
      {
        next = plane_at(inst, o + 1, L->gauss_off[o + 1], 0);
        g0_done = true;
      }
      HIP_CHECK(vksift_hip_octave_chain(plane_at(inst, o, L->gauss_off[o], 0), L->plane_stride[o], inst->d_pyr + L->dog_off[o], next, inst->taps,
                                        VKSIFT_MAX_TAPS, count, sp),
                "octave chain");
      nb_o++;
      if (par && o + 1 < L->n_oct)
        HIP_CHECK(vksift_hip_event_record(inst->ev_oct_ready[o], sp), "event record");
    }
    else
      for (uint32_t s = 1; s < inst->S + 3; s++)
      {
        vksift_hip_Plane dstp = plane_at(inst, o, L->gauss_off[o], s);
        inst->top_scale_stale[o] = false;
        if (s == inst->S + 2 && inst->lazy_top_scale)
        {
          /* nothing reads Gaussian scale S+2 (keypoints use scales 1..S, the next octave scale S): keep its DoG layer only;
           * vksift_downloadScaleSpaceImage() re-creates the plane on demand */
          dstp.base = NULL;
          inst->top_scale_stale[o] = true;
        }
        HIP_CHECK(vksift_hip_blur(plane_at(inst, o, L->gauss_off[o], s - 1), dstp, plane_at(inst, o, L->dog_off[o], s - 1),
                                  &inst->taps[s * VKSIFT_MAX_TAPS], inst->ntaps[s], count, sp),
                  "blur");
        nb_o++;
        if (par && o + 1 < L->n_oct && s == ((inst->coarse_after && o == 0) ? inst->S + 2 : inst->S))
          HIP_CHECK(vksift_hip_event_record(inst->ev_oct_ready[o], sp), "event record");
      }
    vksift_hip_range_pop();
    if (!pipelined || o == 0)
      nblur += nb_o;
    if (overlap)
    {
      if (o == 0 && prof)
        vksift_hip_event_record(PS->ev_pt[1], sp);
      HIP_CHECK(vksift_hip_event_record(inst->ev_pyr_done[o], sp), "event record");
      HIP_CHECK(vksift_hip_stream_wait_event(so, inst->ev_pyr_done[o]), "scale space ready");
    }
    if (pipelined)
    {
      if (o == 0 && prof)
        vksift_hip_event_record(PS->ev_t[2], st);
      vksift_hip_range_push("ExtractKeypoints");
      HIP_CHECK(vksift_hip_extract_keypoints(&jobs[o], count, so), "keypoint extraction");
      vksift_hip_range_pop();
      if (o == 0 && prof)
        vksift_hip_event_record(PS->ev_t[3], st);
      vksift_hip_range_push("ComputeOrientation");
      HIP_CHECK(vksift_hip_orientations(&jobs[o], count, so), "orientation");
      vksift_hip_range_pop();
      if (o == 0 && prof)
        vksift_hip_event_record(PS->ev_t[4], st);
      vksift_hip_range_push("ComputeDescriptors");
      if (overlap && o == 0)
      {
        HIP_CHECK(vksift_hip_event_record(inst->ev_desc_start, st), "event record");
        inst->desc_start_valid = true;
      }
      HIP_CHECK(vksift_hip_descriptors(&jobs[o], count, so), "descriptor");
      vksift_hip_range_pop();
      if (o == 0 && prof)
        vksift_hip_event_record(PS->ev_t[5], st);
    }
    if (par && so != st)
      HIP_CHECK(vksift_hip_event_record(inst->ev_join[0][o], so), "event record");
  }
  if (par)
    for (uint32_t o = 0; o < L->n_oct; o++)
      if (o > 0 || !pipelined)
        HIP_CHECK(vksift_hip_stream_wait_event(st, inst->ev_join[0][o]), "octave join");
  if (overlap)
  {
    /* everything that reads this call's pyramid has been joined into the instance stream */
    HIP_CHECK(vksift_hip_event_record(inst->ev_pyr_free[inst->pyr_cur], st), "event record");
    inst->pyr_free_valid[inst->pyr_cur] = true;
  }
  inst->last_blur_launches = nblur;
  /* profiling: the pyramid interval is octave 0's when pipelined, the whole pyramid's otherwise */
  inst->last_alg_bytes = algorithmic_pyramid_bytes(inst, w, h, pipelined ? 1u : L->n_oct) * count;

  if (!pipelined)
  {
    if (prof)
      vksift_hip_event_record(PS->ev_t[2], st);
    /* Each of the three keypoint stages forks one stream per octave (per-octave scratch, no sharing) and joins back
     * into the main stream, so stage boundaries (and the stage timings) stay well defined. */
#define VKSIFT_STAGE(G, NAME, CALL, WHAT)                                                                   \
  vksift_hip_range_push(NAME);                                                                               \
  if (par)                                                                                                   \
    HIP_CHECK(vksift_hip_event_record(inst->ev_fork[G], st), "event record");                               \
  for (uint32_t o = 0; o < L->n_oct; o++)                                                                    \
  {                                                                                                          \
    vksift_hip_stream so = (par && o > 0) ? inst->oct_stream[o] : st;                                        \
    if (par && o > 0)                                                                                        \
      HIP_CHECK(vksift_hip_stream_wait_event(so, inst->ev_fork[G]), "octave fork");                          \
    HIP_CHECK(CALL(&jobs[o], count, so), WHAT);                                                              \
    if (par && o > 0)                                                                                        \
      HIP_CHECK(vksift_hip_event_record(inst->ev_join[G][o], so), "event record");                           \
  }                                                                                                          \
  if (par)                                                                                                   \
    for (uint32_t o = 1; o < L->n_oct; o++)                                                                  \
      HIP_CHECK(vksift_hip_stream_wait_event(st, inst->ev_join[G][o]), "octave join");                       \
  vksift_hip_range_pop();

    VKSIFT_STAGE(1, "ExtractKeypoints", vksift_hip_extract_keypoints, "keypoint extraction")
    if (prof)
      vksift_hip_event_record(PS->ev_t[3], st);
    VKSIFT_STAGE(2, "ComputeOrientation", vksift_hip_orientations, "orientation")
    if (prof)
      vksift_hip_event_record(PS->ev_t[4], st);
    VKSIFT_STAGE(3, "ComputeDescriptors", vksift_hip_descriptors, "descriptor")
    if (prof)
      vksift_hip_event_record(PS->ev_t[5], st);
#undef VKSIFT_STAGE
  }

  /* recCopySIFTCountCmds (sift_detector.c:1261-1291) */
  HIP_CHECK(vksift_hip_memcpy_d2h(inst->h_found + (size_t)first_buf * VKSIFT_MAX_OCTAVES, inst->d_found + (size_t)first_buf * VKSIFT_MAX_OCTAVES,
                                  sizeof(uint32_t) * VKSIFT_MAX_OCTAVES * count, st),
            "count read-back");
  if (capturing)
  {
    capturing = false;
    vksift_hip_graph exec = NULL;
    HIP_CHECK(vksift_hip_capture_end(st, &exec), "detection graph capture");
    dg->exec = exec;
    dg->w = w, dg->h = h, dg->count = count, dg->first_buf = first_buf, dg->d_src = d_src;
    memcpy(dg->top_scale_stale, inst->top_scale_stale, sizeof(inst->top_scale_stale));
    dg->stamp = ++inst->graph_stamp;
    HIP_CHECK(vksift_hip_graph_launch(dg->exec, st), "detection graph launch");
    if (images)
    {
      HIP_CHECK(vksift_hip_event_record(inst->ev_staging, st), "event record");
      inst->staging_pending = true;
    }
  }
enqueued:
  if (prof)
  {
    vksift_hip_event_record(PS->ev_t[6], st);
    PS->valid = true;
    PS->accounted = false;
    PS->blur_launches = inst->last_blur_launches;
    PS->alg_bytes = inst->last_alg_bytes;
  }
  HIP_CHECK(vksift_hip_event_record(inst->ev_detect, st), "event record");
  inst->detect_pending = true;
  inst->detect_first_buf = first_buf;
  inst->detect_count = count;
  return;

gpu_error:
  if (capturing)
  {
    vksift_hip_graph dead = NULL;
    (void)vksift_hip_capture_end(st, &dead);
    vksift_hip_graph_destroy(dead);
  }
  logError(LOG_TAG, "%s error: Failed to start the detection pipeline.", fn);
  inst->error_cb(VKSIFT_VULKAN_ERROR);
}

void vksift_detectFeatures(vksift_Instance instance, const uint8_t *image_data, const uint32_t image_width, const uint32_t image_height,
                           const uint32_t gpu_buffer_id)
{
  const uint8_t *imgs[1] = {image_data};
  vksift_hip_set_device(instance->device);
  detect_impl(instance, imgs, NULL, 1, image_width, image_height, gpu_buffer_id, "vksift_detectFeatures()");
}

void vksift_ext_detectFeaturesBatch(vksift_Instance instance, const uint8_t *const *images, uint32_t count, uint32_t image_width, uint32_t image_height,
                                    uint32_t first_gpu_buffer_id)
{
  vksift_hip_set_device(instance->device);
  detect_impl(instance, images, NULL, count, image_width, image_height, first_gpu_buffer_id, "vksift_ext_detectFeaturesBatch()");
}

void vksift_ext_detectFeaturesBatchDevice(vksift_Instance instance, const uint8_t *d_images, uint32_t count, uint32_t image_width, uint32_t image_height,
                                          uint32_t first_gpu_buffer_id)
{
  vksift_hip_set_device(instance->device);
  if (d_images == NULL)
  {
    logError(LOG_TAG, "vksift_ext_detectFeaturesBatchDevice() error: invalid input.");
    instance->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return;
  }
  detect_impl(instance, NULL, d_images, count, image_width, image_height, first_gpu_buffer_id, "vksift_ext_detectFeaturesBatchDevice()");
}

/* ------------------------------------------------------------------------------------------------ */
/* feature count / download / upload (sift_memory.c:1060-1272)                                      */
/* ------------------------------------------------------------------------------------------------ */
static void wait_for_buffer(vksift_Instance inst, uint32_t buf)
{
  vksift_hip_set_device(inst->device);
  if (!inst->bufs[buf].counts_valid || (inst->detect_pending && buf >= inst->detect_first_buf && buf < inst->detect_first_buf + inst->detect_count))
  {
    vksift_hip_event_sync(inst->ev_detect);
    mark_detect_done(inst);
  }
  if (inst->match_pending && (buf == inst->match_a || buf == inst->match_b))
  {
    vksift_hip_event_sync(inst->ev_match);
    inst->match_pending = false;
  }
}

/* per-section stored counts, clamped to the section capacity (sift_memory.c:1080-1095) */
static uint32_t buffer_counts(vksift_Instance inst, uint32_t buf, uint32_t *cnt, bool log_lost)
{
  const BufferInfo *b = &inst->bufs[buf];
  if (b->is_packed && b->nb_sections == 0)
    return b->nb_stored;
  uint32_t sum = 0, lost = 0;
  const uint32_t *found = inst->h_found + (size_t)buf * VKSIFT_MAX_OCTAVES;
  for (uint32_t o = 0; o < b->nb_sections; o++)
  {
    uint32_t n = found[o];
    if (n > b->sec_cap[o])
    {
      lost += n - b->sec_cap[o];
      n = b->sec_cap[o];
    }
    if (cnt)
      cnt[o] = n;
    sum += n;
  }
  if (lost > 0 && log_lost)
    logError(LOG_TAG,
             "%d feature(s) lost because the SIFT buffer was full, consider increasing "
             "the maximum number of SIFT features per buffer in the configuration.",
             lost);
  return sum;
}

uint32_t vksift_getFeaturesNumber(vksift_Instance instance, const uint32_t gpu_buffer_id)
{
  if (!buffer_idx_valid(instance, gpu_buffer_id))
  {
    logError(LOG_TAG, "vksift_getFeaturesNumber() error: invalid input.");
    instance->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return 0;
  }
  wait_for_buffer(instance, gpu_buffer_id);
  return buffer_counts(instance, gpu_buffer_id, NULL, true);
}

void vksift_downloadFeatures(vksift_Instance instance, vksift_Feature *feats_ptr, uint32_t gpu_buffer_id)
{
  if (!buffer_idx_valid(instance, gpu_buffer_id))
  {
    logError(LOG_TAG, "vksift_downloadFeatures() error: invalid input.");
    instance->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return;
  }
  vksift_Instance inst = instance;
  wait_for_buffer(inst, gpu_buffer_id);
  const BufferInfo *b = &inst->bufs[gpu_buffer_id];
  const uint8_t *base = inst->d_feats + (uint64_t)gpu_buffer_id * inst->buf_stride;
  if (b->nb_sections == 0)
  {
    HIP_CHECK(vksift_hip_memcpy_d2h(feats_ptr, base, (size_t)b->nb_stored * FEAT_BYTES, inst->stream), "feature download");
  }
  else
  {
    uint32_t cnt[VKSIFT_MAX_OCTAVES] = {0};
    buffer_counts(inst, gpu_buffer_id, cnt, false);
    uint32_t out = 0;
    for (uint32_t o = 0; o < b->nb_sections; o++)
    {
      HIP_CHECK(vksift_hip_memcpy_d2h((uint8_t *)feats_ptr + (size_t)out * FEAT_BYTES, base + (size_t)b->sec_off[o] * FEAT_BYTES, (size_t)cnt[o] * FEAT_BYTES,
                                      inst->stream),
                "feature download");
      out += cnt[o];
    }
  }
  HIP_CHECK(vksift_hip_stream_sync(inst->stream), "feature download");
  return;
gpu_error:
  logError(LOG_TAG, "vksift_downloadFeatures() error when downloading detection results.");
  instance->error_cb(VKSIFT_VULKAN_ERROR);
}

void vksift_uploadFeatures(vksift_Instance instance, const vksift_Feature *feats_ptr, const uint32_t nb_feats, const uint32_t gpu_buffer_id)
{
  if (!buffer_idx_valid(instance, gpu_buffer_id) || nb_feats > instance->cfg.max_nb_sift_per_buffer)
  {
    if (nb_feats > instance->cfg.max_nb_sift_per_buffer)
      logError(LOG_TAG, "Provided features count (%d) is greater than the configured maximum number of features per GPU buffer size (%d).", nb_feats,
               instance->cfg.max_nb_sift_per_buffer);
    logError(LOG_TAG, "vksift_uploadFeatures() error: invalid input.");
    instance->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return;
  }
  vksift_Instance inst = instance;
  wait_for_buffer(inst, gpu_buffer_id);
  BufferInfo *b = &inst->bufs[gpu_buffer_id];
  HIP_CHECK(vksift_hip_memcpy_h2d(inst->d_feats + (uint64_t)gpu_buffer_id * inst->buf_stride, feats_ptr, (size_t)nb_feats * FEAT_BYTES, inst->stream),
            "feature upload");
  HIP_CHECK(vksift_hip_stream_sync(inst->stream), "feature upload");
  /* the buffer becomes one packed section (sift_memory.c:1262-1266) */
  b->is_packed = true;
  b->nb_stored = nb_feats;
  b->nb_sections = 0;
  b->counts_valid = true;
  return;
gpu_error:
  logError(LOG_TAG, "vksift_uploadFeatures() error when uploading SIFT features to GPU memory.");
  instance->error_cb(VKSIFT_VULKAN_ERROR);
}

/* ------------------------------------------------------------------------------------------------ */
/* matching (vulkansift.c:417-462, sift_memory.c:957-1058, sift_matcher.c:408-486)                  */
/* ------------------------------------------------------------------------------------------------ */
/* Same section layout (so one kernel launch can serve both buffers)? */
static bool same_layout(const BufferInfo *x, const BufferInfo *y)
{
  if (x->nb_sections != y->nb_sections)
    return false;
  if (x->nb_sections == 0)
    return x->nb_stored == y->nb_stored;
  for (uint32_t o = 0; o < x->nb_sections; o++)
    if (x->sec_off[o] != y->sec_off[o] || x->sec_cap[o] != y->sec_cap[o])
      return false;
  return true;
}

/* Device-side gather of `count` buffers (all with the layout of bufs[ids[0]]) into match slots first_slot.. .
 * The reference physically packs the octave sections (pack_BufferMemory, sift_memory.c:957-1047) after reading the
 * counts on the host; here the gather kernel reads the counters in HBM and walks the sections in the same order, so
 * nothing waits on the host. Returns the launch bound on the row count through *max_rows_out. */
static int gather_buffers(vksift_Instance inst, const MatchScratch *ms, const uint32_t *ids, uint32_t count, uint32_t first_slot, bool side_b, uint8_t *d_desc_base,
                          uint32_t n_index, uint32_t pad_rows_to, uint32_t *max_rows_out)
{
  const BufferInfo *b = &inst->bufs[ids[0]];
  const uint32_t cap = inst->cfg.max_nb_sift_per_buffer;
  uint8_t *d_desc = d_desc_base + (uint64_t)first_slot * inst->desc_slot_stride;
  uint32_t *d_norm = ms->norms + (uint64_t)first_slot * inst->norm_slot_stride + (side_b ? cap + 32u : 0u);
  uint32_t *d_n = ms->match_n + (size_t)first_slot * 4 + n_index;
  uint32_t max_rows = 0;
  int e;
  if (b->nb_sections == 0)
  {
    uint32_t zero_off = 0, cap1 = b->nb_stored, fixed1 = b->nb_stored;
    max_rows = b->nb_stored;
    e = vksift_hip_gather_sections(inst->d_feats, inst->buf_stride, ids, count, 1, &zero_off, &cap1, &fixed1, NULL, 0, max_rows, pad_rows_to, d_desc,
                                   inst->desc_slot_stride, d_norm, inst->norm_slot_stride, d_n, 4, inst->stream);
  }
  else
  {
    detect_running(inst); /* refreshes counts_valid if the last detection has finished */
    bool all_known = true;
    uint32_t known_max = 0, cap_sum = 0;
    for (uint32_t o = 0; o < b->nb_sections; o++)
      cap_sum += b->sec_cap[o];
    for (uint32_t i = 0; i < count; i++)
    {
      const BufferInfo *bi = &inst->bufs[ids[i]];
      if (!bi->counts_valid)
      {
        all_known = false;
        break;
      }
      uint32_t known = 0;
      const uint32_t *found = inst->h_found + (size_t)ids[i] * VKSIFT_MAX_OCTAVES;
      for (uint32_t o = 0; o < bi->nb_sections; o++)
        known += found[o] < bi->sec_cap[o] ? found[o] : bi->sec_cap[o];
      if (known > known_max)
        known_max = known;
    }
    max_rows = all_known ? known_max : cap_sum; /* counts already on the host? then bound the launch by the real total */
    e = vksift_hip_gather_sections(inst->d_feats, inst->buf_stride, ids, count, b->nb_sections, b->sec_off, b->sec_cap, NULL, inst->d_found,
                                   VKSIFT_MAX_OCTAVES, max_rows, pad_rows_to, d_desc, inst->desc_slot_stride, d_norm, inst->norm_slot_stride, d_n, 4,
                                   inst->stream);
  }
  *max_rows_out = max_rows;
  return e;
}

static MatchScratch fwd_scratch(vksift_Instance inst)
{
  MatchScratch ms = {inst->d_desc_a, inst->d_desc_b, inst->d_matches, inst->d_norms, inst->d_match_n};
  return ms;
}

static int match_slots(vksift_Instance inst, const MatchScratch *ms, const uint32_t *ids_a, const uint32_t *ids_b, uint32_t count, uint32_t first_slot)
{
  const uint32_t cap = inst->cfg.max_nb_sift_per_buffer;
  uint32_t max_na = 0, max_nb = 0;
  int e = gather_buffers(inst, ms, ids_a, count, first_slot, false, ms->desc_a, 0, 0u, &max_na);
  if (e)
    return e;
  /* Get2NearestNeighbors.comp:66-67 reads b[0] and b[1] unconditionally (stale memory in the reference when B holds
   * fewer than two features); here the missing rows are defined as all-zero descriptors. */
  e = gather_buffers(inst, ms, ids_b, count, first_slot, true, ms->desc_b, 1, 2u, &max_nb);
  if (e)
    return e;
  const uint32_t *norm_a = ms->norms + (uint64_t)first_slot * inst->norm_slot_stride;
  return vksift_hip_match_2nn_async(ms->desc_a + (uint64_t)first_slot * inst->desc_slot_stride, norm_a, max_na,
                                    ms->desc_b + (uint64_t)first_slot * inst->desc_slot_stride, norm_a + cap + 32u, (uint32_t *)norm_a + 2u * cap + 64u,
                                    ms->match_n + (size_t)first_slot * 4, ms->matches + (uint64_t)first_slot * inst->match_slot_stride, count,
                                    inst->desc_slot_stride, inst->norm_slot_stride, inst->match_slot_stride, 4, inst->d_match_partial, inst->stream);
}

/* reverse-matching scratch + survivor lists of vksift_ext_matchFeaturesFiltered, allocated on first use */
static bool ensure_filter_scratch(vksift_Instance inst)
{
  if (inst->d_filtered)
    return true;
  const uint32_t bc = inst->batch_cap;
  inst->filtered_slot_stride = (((uint64_t)inst->cfg.max_nb_sift_per_buffer * 16u) + 255u) & ~(uint64_t)255u;
  bool ok = true;
  ok = ok && (inst->rev.desc_a = vksift_hip_malloc(inst->desc_slot_stride * bc)) != NULL;
  ok = ok && (inst->rev.desc_b = vksift_hip_malloc(inst->desc_slot_stride * bc)) != NULL;
  ok = ok && (inst->rev.matches = vksift_hip_malloc(inst->match_slot_stride * bc)) != NULL;
  ok = ok && (inst->rev.norms = vksift_hip_malloc(sizeof(uint32_t) * inst->norm_slot_stride * bc)) != NULL;
  ok = ok && (inst->rev.match_n = vksift_hip_malloc(sizeof(uint32_t) * 4 * bc)) != NULL;
  ok = ok && (inst->d_filtered_n = vksift_hip_malloc(sizeof(uint32_t) * bc)) != NULL;
  ok = ok && (inst->h_filtered_n = vksift_hip_host_malloc(sizeof(uint32_t) * bc)) != NULL;
  ok = ok && (inst->d_filtered = vksift_hip_malloc(inst->filtered_slot_stride * bc)) != NULL;
  return ok;
}

static void match_impl(vksift_Instance inst, const uint32_t *ids_a, const uint32_t *ids_b, uint32_t count, const char *fn, bool filter, float ratio,
                       bool cross_check)
{
  bool valid = count >= 1 && count <= inst->batch_cap && count <= 64;
  for (uint32_t i = 0; valid && i < count; i++)
    valid = buffer_idx_valid(inst, ids_a[i]) && buffer_idx_valid(inst, ids_b[i]);
  if (!valid)
  {
    logError(LOG_TAG, "%s error: invalid input.", fn);
    inst->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return;
  }
  vksift_hip_set_device(inst->device);
  if (inst->profiling)
    vksift_hip_event_record(inst->ev_m[0], inst->stream);
  vksift_hip_range_push("Matching");
  /* one batched launch sequence when every A buffer and every B buffer share a section layout (always the case
   * after a batched detection), otherwise pair by pair */
  bool uniform = true;
  for (uint32_t i = 1; i < count && uniform; i++)
    uniform = same_layout(&inst->bufs[ids_a[0]], &inst->bufs[ids_a[i]]) && same_layout(&inst->bufs[ids_b[0]], &inst->bufs[ids_b[i]]);
  const MatchScratch fwd = fwd_scratch(inst);
  if (uniform)
    HIP_CHECK(match_slots(inst, &fwd, ids_a, ids_b, count, 0), "2-NN matching");
  else
    for (uint32_t i = 0; i < count; i++)
      HIP_CHECK(match_slots(inst, &fwd, ids_a + i, ids_b + i, 1, i), "2-NN matching");
  HIP_CHECK(vksift_hip_memcpy_d2h(inst->h_match_n, inst->d_match_n, sizeof(uint32_t) * 4 * count, inst->stream), "match count read-back");
  inst->filtered_slots_used = 0;
  if (filter)
  {
    /* the reverse matching, then cross-check + ratio test on the device; only the survivors are read back */
    if (!ensure_filter_scratch(inst))
    {
      logError(LOG_TAG, "%s error: out of device memory for the filtered-matching scratch.", fn);
      goto gpu_error;
    }
    if (cross_check)
    {
      if (uniform)
        HIP_CHECK(match_slots(inst, &inst->rev, ids_b, ids_a, count, 0), "reverse 2-NN matching");
      else
        for (uint32_t i = 0; i < count; i++)
          HIP_CHECK(match_slots(inst, &inst->rev, ids_b + i, ids_a + i, 1, i), "reverse 2-NN matching");
    }
    HIP_CHECK(vksift_hip_filter_matches(inst->d_matches, inst->match_slot_stride, cross_check ? inst->rev.matches : NULL, inst->match_slot_stride,
                                        inst->d_match_n, 4, ratio, count, inst->d_filtered, inst->filtered_slot_stride, inst->d_filtered_n, inst->stream),
              "match filtering");
    HIP_CHECK(vksift_hip_memcpy_d2h(inst->h_filtered_n, inst->d_filtered_n, sizeof(uint32_t) * count, inst->stream), "filtered count read-back");
    inst->filtered_slots_used = count;
  }
  vksift_hip_range_pop();
  if (inst->profiling)
  {
    vksift_hip_event_record(inst->ev_m[1], inst->stream);
    inst->match_timing_valid = true;
  }
  HIP_CHECK(vksift_hip_event_record(inst->ev_match, inst->stream), "event record");
  inst->match_pending = true;
  inst->match_slots_used = count;
  inst->match_a = ids_a[0];
  inst->match_b = ids_b[0];
  return;
gpu_error:
  logError(LOG_TAG, "%s error: Failed to start the matching pipeline.", fn);
  inst->error_cb(VKSIFT_VULKAN_ERROR);
}

void vksift_matchFeatures(vksift_Instance instance, uint32_t gpu_buffer_id_A, uint32_t gpu_buffer_id_B)
{
  match_impl(instance, &gpu_buffer_id_A, &gpu_buffer_id_B, 1, "vksift_matchFeatures()", false, 0.f, false);
}

void vksift_ext_matchFeaturesBatch(vksift_Instance instance, uint32_t count, const uint32_t *gpu_buffer_ids_A, const uint32_t *gpu_buffer_ids_B)
{
  match_impl(instance, gpu_buffer_ids_A, gpu_buffer_ids_B, count, "vksift_ext_matchFeaturesBatch()", false, 0.f, false);
}

void vksift_ext_matchFeaturesFiltered(vksift_Instance instance, uint32_t count, const uint32_t *gpu_buffer_ids_A, const uint32_t *gpu_buffer_ids_B, float ratio,
                                      bool cross_check)
{
  if (!(ratio > 0.f))
  {
    logError(LOG_TAG, "vksift_ext_matchFeaturesFiltered() error: invalid input.");
    instance->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return;
  }
  match_impl(instance, gpu_buffer_ids_A, gpu_buffer_ids_B, count, "vksift_ext_matchFeaturesFiltered()", true, ratio, cross_check);
}

static void wait_match(vksift_Instance inst)
{
  vksift_hip_set_device(inst->device);
  if (inst->match_pending)
  {
    vksift_hip_event_sync(inst->ev_match);
    inst->match_pending = false;
  }
  inst->curr_nb_matches = inst->h_match_n[0];
}

/* The reference knows N_A on the host when vksift_matchFeatures returns (it blocks while packing); here the count is
 * produced on the device, so this accessor waits for the matching pipeline if it is still running. */
uint32_t vksift_getMatchesNumber(vksift_Instance instance)
{
  wait_match(instance);
  return instance->curr_nb_matches;
}

uint32_t vksift_ext_getMatchesNumberBatch(vksift_Instance instance, uint32_t pair)
{
  wait_match(instance);
  if (pair >= instance->match_slots_used)
  {
    logError(LOG_TAG, "vksift_ext_getMatchesNumberBatch() error: invalid input.");
    instance->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return 0;
  }
  return instance->h_match_n[(size_t)pair * 4];
}

static void download_matches(vksift_Instance inst, uint32_t pair, vksift_Match_2NN *matches, const char *fn)
{
  wait_match(inst);
  if (pair >= inst->match_slots_used && !(pair == 0 && inst->match_slots_used == 0))
  {
    logError(LOG_TAG, "%s error: invalid input.", fn);
    inst->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return;
  }
  uint32_t n = inst->h_match_n[(size_t)pair * 4];
  if (n > 0)
  {
    HIP_CHECK(vksift_hip_memcpy_d2h(matches, inst->d_matches + (uint64_t)pair * inst->match_slot_stride, (size_t)n * MATCH_BYTES, inst->stream),
              "match read-back");
    HIP_CHECK(vksift_hip_stream_sync(inst->stream), "match read-back");
  }
  return;
gpu_error:
  logError(LOG_TAG, "%s error when downloading SIFT matches from GPU memory.", fn);
  inst->error_cb(VKSIFT_VULKAN_ERROR);
}

void vksift_downloadMatches(vksift_Instance instance, vksift_Match_2NN *matches) { download_matches(instance, 0, matches, "vksift_downloadMatches()"); }

uint32_t vksift_ext_getFilteredMatchesNumber(vksift_Instance instance, uint32_t pair)
{
  wait_match(instance);
  if (pair >= instance->filtered_slots_used)
  {
    logError(LOG_TAG, "vksift_ext_getFilteredMatchesNumber() error: invalid input.");
    instance->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return 0;
  }
  return instance->h_filtered_n[pair];
}

void vksift_ext_downloadFilteredMatches(vksift_Instance instance, uint32_t pair, vksift_ext_FilteredMatch *matches)
{
  vksift_Instance inst = instance;
  wait_match(inst);
  if (pair >= inst->filtered_slots_used)
  {
    logError(LOG_TAG, "vksift_ext_downloadFilteredMatches() error: invalid input.");
    inst->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return;
  }
  const uint32_t n = inst->h_filtered_n[pair];
  if (n > 0)
  {
    HIP_CHECK(vksift_hip_memcpy_d2h(matches, inst->d_filtered + (uint64_t)pair * inst->filtered_slot_stride, (size_t)n * sizeof(vksift_ext_FilteredMatch),
                                    inst->stream),
              "filtered match read-back");
    HIP_CHECK(vksift_hip_stream_sync(inst->stream), "filtered match read-back");
  }
  return;
gpu_error:
  logError(LOG_TAG, "vksift_ext_downloadFilteredMatches() error when downloading the filtered matches from GPU memory.");
  inst->error_cb(VKSIFT_VULKAN_ERROR);
}

void vksift_ext_downloadMatchesBatch(vksift_Instance instance, uint32_t pair, vksift_Match_2NN *matches)
{
  download_matches(instance, pair, matches, "vksift_ext_downloadMatchesBatch()");
}

/* ------------------------------------------------------------------------------------------------ */
/* scale-space inspection (vulkansift.c:464-518, sift_memory.c:1303-1383)                           */
/* ------------------------------------------------------------------------------------------------ */
uint8_t vksift_getScaleSpaceNbOctaves(vksift_Instance instance) { return (uint8_t)instance->lay.n_oct; }

void vksift_getScaleSpaceOctaveResolution(vksift_Instance instance, const uint8_t octave, uint32_t *octave_images_width, uint32_t *octave_images_height)
{
  if (octave >= instance->lay.n_oct)
  {
    logError(LOG_TAG, "vksift_getScaleSpaceOctaveResolution() error: invalid input. Requested octave idx is %d but the current number of octave is %d",
             octave, instance->lay.n_oct);
    instance->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return;
  }
  *octave_images_width = instance->lay.w[octave];
  *octave_images_height = instance->lay.h[octave];
}

static void download_plane(vksift_Instance inst, uint8_t octave, uint8_t scale, bool is_dog, float *dst, const char *fn)
{
  uint32_t nscales = inst->S + (is_dog ? 2 : 3);
  if (octave >= inst->lay.n_oct || scale >= nscales)
  {
    if (octave >= inst->lay.n_oct)
      logError(LOG_TAG, "Requested octave idx is %d but the current number of octaves is %d", octave, inst->lay.n_oct);
    else
      logError(LOG_TAG, "Requested scale idx is %d but the number of %s scales is %d", scale, is_dog ? "DoG" : "blurred", nscales);
    logError(LOG_TAG, "%s error: invalid input.", fn);
    inst->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return;
  }
  vksift_hip_set_device(inst->device);
  /* images cannot be read while a detection runs (vulkansift.c:490-491) */
  HIP_CHECK(wait_all(inst), "stream synchronisation");
  const PyrLayout *L = &inst->lay;
  if (!is_dog && scale == inst->S + 2 && inst->top_scale_stale[octave])
  {
    /* the detection pipeline kept only the DoG layer of the last scale: blur it now (image 0, the one this API exposes) */
    const vksift_hip_Plane no_dog = {NULL, 0, 0, 0, 0};
    HIP_CHECK(vksift_hip_blur(plane_at(inst, octave, L->gauss_off[octave], scale - 1), plane_at(inst, octave, L->gauss_off[octave], scale), no_dog,
                              &inst->taps[scale * VKSIFT_MAX_TAPS], inst->ntaps[scale], 1, inst->stream),
              "top scale blur");
    inst->top_scale_stale[octave] = false;
  }
  const float *src = inst->d_pyr + (is_dog ? L->dog_off[octave] : L->gauss_off[octave]) + (uint64_t)scale * L->plane_stride[octave];
  HIP_CHECK(vksift_hip_memcpy2d_d2h(dst, sizeof(float) * L->w[octave], src, sizeof(float) * L->pitch[octave], sizeof(float) * L->w[octave], L->h[octave],
                                    inst->stream),
            "plane download");
  HIP_CHECK(vksift_hip_stream_sync(inst->stream), "plane download");
  return;
gpu_error:
  logError(LOG_TAG, "%s error when downloading pyramid image from GPU memory.", fn);
  inst->error_cb(VKSIFT_VULKAN_ERROR);
}

void vksift_downloadScaleSpaceImage(vksift_Instance instance, const uint8_t octave, const uint8_t scale, float *blurred_image)
{
  download_plane(instance, octave, scale, false, blurred_image, "vksift_downloadScaleSpaceImage()");
}

void vksift_downloadDoGImage(vksift_Instance instance, const uint8_t octave, const uint8_t scale, float *dog_image)
{
  download_plane(instance, octave, scale, true, dog_image, "vksift_downloadDoGImage()");
}

void vksift_presentDebugFrame(vksift_Instance instance)
{
  (void)instance;
  logWarning(LOG_TAG, "vksift_presentDebugFrame() was called but instance has no external window configured.");
}

/* ------------------------------------------------------------------------------------------------ */
/* extensions                                                                                       */
/* ------------------------------------------------------------------------------------------------ */
void vksift_ext_setProfiling(vksift_Instance instance, bool enabled)
{
  instance->profiling = enabled;
  instance->prof[0].valid = instance->prof[1].valid = false;
  instance->prof[0].accounted = instance->prof[1].accounted = false;
  instance->match_timing_valid = false;
  memset(instance->acc_ms, 0, sizeof(instance->acc_ms));
  instance->acc_calls = 0;
  instance->acc_blur_launches = 0;
  instance->acc_alg_bytes = 0;
}

void vksift_ext_getAccumulatedDetectTimings(vksift_Instance instance, vksift_ext_DetectTimings *sum, uint32_t *nb_calls, bool reset)
{
  memset(sum, 0, sizeof(*sum));
  *nb_calls = 0;
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
  sum->nb_blur_launches = (uint32_t)instance->acc_blur_launches;
  sum->pyramid_algorithmic_bytes = instance->acc_alg_bytes;
  *nb_calls = instance->acc_calls;
  if (reset)
  {
    memset(instance->acc_ms, 0, sizeof(instance->acc_ms));
    instance->acc_calls = 0;
    instance->acc_blur_launches = 0;
    instance->acc_alg_bytes = 0;
  }
}

void vksift_ext_getDetectTimings(vksift_Instance instance, vksift_ext_DetectTimings *out)
{
  memset(out, 0, sizeof(*out));
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
  out->nb_blur_launches = instance->last_blur_launches;
  out->pyramid_algorithmic_bytes = instance->last_alg_bytes;
}

float vksift_ext_getMatchTime(vksift_Instance instance)
{
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
  uint32_t n = 0, max_rows = 0;
  HIP_CHECK(wait_all(inst), "stream synchronisation");
  {
    /* gather into slot 0's A scratch (norms are a by-product), then copy the rows out */
    const MatchScratch fwd = fwd_scratch(inst);
    HIP_CHECK(gather_buffers(inst, &fwd, &gpu_buffer_id, 1, 0, false, inst->d_desc_a, 2, 0u, &max_rows), "descriptor gather");
    HIP_CHECK(vksift_hip_memcpy_d2h(inst->h_match_n + 2, inst->d_match_n + 2, sizeof(uint32_t), inst->stream), "descriptor gather");
    HIP_CHECK(vksift_hip_stream_sync(inst->stream), "descriptor gather");
    n = inst->h_match_n[2];
    HIP_CHECK(vksift_hip_memcpy_d2d(d_descriptors, inst->d_desc_a, (size_t)n * 128u, inst->stream), "descriptor gather");
    HIP_CHECK(vksift_hip_stream_sync(inst->stream), "descriptor gather");
  }
  return n;
gpu_error:
  logError(LOG_TAG, "vksift_ext_exportDescriptorsDevice() error when exporting descriptors.");
  instance->error_cb(VKSIFT_VULKAN_ERROR);
  return 0;
}
