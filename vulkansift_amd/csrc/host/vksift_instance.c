/*
 * vksift_instance.c — instance creation / destruction, scale-space layout, synchronisation helpers (vulkansift.c:165-313, sift_memory.c:15-87,133-360)
 */
#include "vksift_internal.h"

/* ------------------------------------------------------------------------------------------------ */
/* layout helpers                                                                                   */
/* ------------------------------------------------------------------------------------------------ */
static uint32_t round_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }

void compute_layout(vksift_Instance inst, uint32_t w, uint32_t h, PyrLayout *L)
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
  }
  L->img_floats = off;
  uint64_t so = 0, co = 0;
  for (uint32_t o = 0; o < L->n_oct; o++)
  {
    L->seg_off[o] = so;
    so += (uint64_t)inst->S * L->h[o] * ((L->w[o] + 63) / 64);
    L->cand_off[o] = co;
    /* strict 3x3x3 maxima cannot be denser than 1/8 of the texels (no two are adjacent), nor can minima: 1/4 together is a
     * bound no image exceeds, so candidates are only ever lost through the section capacity, like in the reference */
    L->cand_cap[o] = (uint64_t)inst->S * L->w[o] * L->h[o] / 4u + 64u;
    co += L->cand_cap[o];
  }
  L->seg_total = so;
  L->cand_total = co;
}

static bool place_pyramid_buffers(vksift_Instance inst, size_t bytes, uint64_t img_stride, const PyrLayout *L, uint32_t need, float **out, bool may_search);

void set_buffer_sections(vksift_Instance inst, uint32_t buf, uint32_t n_oct, uint32_t w, uint32_t h)
{
  BufferInfo *b = &inst->bufs[buf];
  memset(b->sec_off, 0, sizeof(b->sec_off));
  memset(b->sec_cap, 0, sizeof(b->sec_cap));
  b->is_packed = false;
  b->nb_stored = 0;
  b->nb_sections = n_oct;
  b->in_w = w;
  b->in_h = h;
  vksift_hm_section_caps(inst->cfg.max_nb_sift_per_buffer, n_oct, b->sec_cap);
  uint32_t off = 0;
  for (uint32_t o = 0; o < n_oct; o++)
  {
    b->sec_off[o] = off;
    off += b->sec_cap[o];
  }
}


/* ------------------------------------------------------------------------------------------------ */
/* instance                                                                                         */
/* ------------------------------------------------------------------------------------------------ */
static vksift_Result create_instance(vksift_Instance *instance_ptr, const vksift_Config *config, uint32_t batch_cap)
{
  assert(instance_ptr != NULL);
  assert(*instance_ptr == NULL);
  assert(config != NULL);

  if (!vksift_g_loaded)
  {
    logError(LOG_TAG, "vksift_createInstance() failure: GPU runtime not available. vksift_loadVulkan() must be called before using this function.");
    return VKSIFT_VULKAN_ERROR;
  }
  if (!config_is_valid(config))
  {
    logError(LOG_TAG, "vksift_createInstance() failed: the configuration was rejected (see above).");
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
  inst->det_cap = batch_cap;
  {
    /* deferred submission of plain detect calls (vksift_internal.h): needs a second SIFT buffer to have anything to batch */
    const char *e = getenv("VKSIFT_DEFER");
    inst->defer_enabled = !(e && e[0] == '0') && config->sift_buffer_count >= 2u;
    e = getenv("VKSIFT_DEFER_MAX");
    inst->defer_max = e ? (uint32_t)strtoul(e, NULL, 10) : 128u;
    {
      const char *c = getenv("VKSIFT_DEFER_CHUNK");
      inst->defer_chunk = c ? (uint32_t)strtoul(c, NULL, 10) : 16u;
    }
    if (inst->defer_max > config->sift_buffer_count)
      inst->defer_max = config->sift_buffer_count;
    if (inst->defer_max < 2u)
      inst->defer_enabled = false;
  }

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
  /* the reference's FLOAT16 mode binds R16_SFLOAT images to shaders that declare r32f (undefined in Vulkan); this build defines it:
   * texels stored as IEEE binary16 (round to nearest even), widened exactly on every read, all arithmetic fp32 */
  inst->fp16 = config->pyramid_precision_mode == VKSIFT_PYRAMID_PRECISION_FLOAT16;
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
    /* Overlap mode: the (bandwidth-bound) scale-space construction of detection N+1 runs on its own stream, beside the matching
     * queued behind detection N's descriptors (rounds 2-3 also ran it under the descriptors themselves, out of a second scale-space
     * buffer: within 1 % in frames/s, and every stage interval measured the contention instead of the kernel — vksift_detect.c:
     * ev_desc_start). With that gate the next scale-space starts only when every reader of the previous one is done, so ONE buffer
     * serves (half the memory; and the measured placement, place_pyramid_buffers, has to find one fast range, not two).
     * Default: instances created for batches of 8 images and more (vksift_ext_createBatchInstance) — a single-image instance keeps
     * the hipGraph replay of small detections, which excludes overlapped calls. VKSIFT_PYR_PINGPONG=0 / 1 forces the mode off / on,
     * =2 is the mode with two buffers (rounds 2-5). */
    const char *e = getenv("VKSIFT_PYR_PINGPONG");
    inst->pyr_pingpong = e ? (e[0] == '1' || e[0] == '2') : (batch_cap >= 8u);
    inst->pyr_nbuf = (e && e[0] == '2') ? 2u : 1u;
    /* an instance whose detection capacity grew later (deferred submission) overlaps its batches of 8 and more only: its single
     * detections keep the forked scale-space and the graph replay */
    inst->overlap_min_count = 1u;
    inst->overlap_forced = e != NULL;
  }
  /* (the scale-space buffers themselves: below, once the stream exists — they are placed by measurement, place_pyramid_buffers) */
  ALLOC_D(inst->d_input, (size_t)inst->max_image_size * inst->det_cap);
  ALLOC_H(inst->h_input, (size_t)inst->max_image_size * inst->det_cap);
  ALLOC_D(inst->d_feats, inst->buf_stride * config->sift_buffer_count);
  ALLOC_D(inst->d_found, sizeof(uint32_t) * VKSIFT_MAX_OCTAVES * config->sift_buffer_count);
  ALLOC_H(inst->h_found, sizeof(uint32_t) * VKSIFT_MAX_OCTAVES * config->sift_buffer_count);
  ALLOC_D(inst->d_seg_mask, sizeof(uint64_t) * inst->seg_cap * inst->det_cap);
  ALLOC_D(inst->d_seg_off, sizeof(uint32_t) * inst->seg_cap * inst->det_cap);
  ALLOC_D(inst->d_cand_xy, sizeof(uint32_t) * inst->cand_cap * inst->det_cap);
  ALLOC_D(inst->d_cand_flag, sizeof(uint32_t) * inst->cand_cap * inst->det_cap);
  ALLOC_D(inst->d_cand_n, sizeof(uint32_t) * inst->det_cap * VKSIFT_MAX_OCTAVES);
  ALLOC_D(inst->d_ori_ang, sizeof(float) * VKSIFT_HIP_MAX_ORI * inst->ori_cap * inst->det_cap);
  ALLOC_D(inst->d_ori_cnt, sizeof(uint32_t) * inst->ori_cap * inst->det_cap);
  ALLOC_D(inst->d_desc_fp, sizeof(float) * DESC_FP_TAB_MAX);
  /* matching scratch: one slot per batch entry (slot 0 serves vksift_matchFeatures) */
  inst->desc_slot_stride = (((uint64_t)config->max_nb_sift_per_buffer * 128u + 256u) + 255u) & ~(uint64_t)255u;
  inst->match_slot_stride = (((uint64_t)config->max_nb_sift_per_buffer * MATCH_BYTES) + 255u) & ~(uint64_t)255u;
  inst->redo_slot_stride = (uint64_t)config->max_nb_sift_per_buffer + 32u;
  inst->cache_norm_stride = (uint64_t)config->max_nb_sift_per_buffer + 32u;
  /* the matcher's per-buffer cache (sift_buffer_count x max_nb_sift_per_buffer x 132 B: 1.7 GB for 128 buffers of 100 000) and the
   * partial lists of the single-pair kernel are allocated by the first matching / export (ensure_match_cache): detect-only
   * users never pay for them */
  ALLOC_D(inst->d_cache_n, sizeof(uint32_t) * config->sift_buffer_count);
  inst->cache_valid = (bool *)calloc(config->sift_buffer_count, sizeof(bool));
  inst->cache_queued = (bool *)calloc(config->sift_buffer_count, sizeof(bool));
  ok = ok && inst->cache_valid != NULL && inst->cache_queued != NULL;
  ALLOC_D(inst->d_matches, inst->match_slot_stride * batch_cap);
  ALLOC_D(inst->d_redo, sizeof(uint32_t) * inst->redo_slot_stride * batch_cap);
  ALLOC_D(inst->d_match_n, sizeof(uint32_t) * 4 * batch_cap);
  ALLOC_H(inst->h_match_n, sizeof(uint32_t) * 4 * batch_cap);
  inst->h_matches = NULL;
  inst->bufs = (BufferInfo *)calloc(config->sift_buffer_count, sizeof(BufferInfo));
  inst->match_busy = (bool *)calloc(config->sift_buffer_count, sizeof(bool));
  ok = ok && inst->bufs != NULL && inst->match_busy != NULL;
  /* All streams at the default priority: a high-priority instance stream with low-priority octave streams was measured
   * 20 % slower on MI355X (11.3k vs 14.1k frames/s). */
  inst->stream = vksift_hip_stream_create();
  if (ok)
  {
    /* first of the large blocks after the stream: candidates need room, and everything allocated before stays where it is */
    ok = place_pyramid_buffers(inst, pyr_texel_bytes(inst) * inst->pyr_img_stride * inst->det_cap, inst->pyr_img_stride, &L, inst->pyr_nbuf,
                               inst->d_pyr_buf, true);
    inst->d_pyr = inst->d_pyr_buf[0];
  }
  inst->pyr_stream = vksift_hip_stream_create();
  inst->dl_stream = vksift_hip_stream_create();
  inst->up_stream = vksift_hip_stream_create();
  inst->ev_pyr_done = vksift_hip_event_create();
  inst->ev_desc_start = vksift_hip_event_create();
  inst->ev_input_free = vksift_hip_event_create();
  for (int i = 0; i < 2; i++)
    inst->ev_pyr_free[i] = vksift_hip_event_create();
  {
    inst->alt_order = true; /* consecutive launches of a chain walk the batch in opposite directions (+9 % on the chain, round 3) */
    const char *e = getenv("VKSIFT_FORK_SCALES");
    inst->fork_scales = !(e && e[0] == '0');
    inst->fork_streams = 1; /* two branch streams measured no faster in stream order and 10 % slower in a replayed graph */
    inst->fork_max_pixels = (uint64_t)16 << 20;
    for (int i = 0; i < VKSIFT_MAX_OCTAVES; i++)
      inst->ev_fork[i] = vksift_hip_event_create();
    inst->ev_join[0] = vksift_hip_event_create();
    inst->ev_join[1] = vksift_hip_event_create();
    inst->side_stream = vksift_hip_stream_create();
    if (!inst->side_stream || !inst->ev_join[0] || !inst->ev_join[1])
      inst->fork_scales = false;
    for (int i = 0; i < VKSIFT_MAX_OCTAVES; i++)
      if (!inst->ev_fork[i])
        inst->fork_scales = false;
    e = getenv("VKSIFT_LDS_CHAIN");
    inst->lds_chain = !(e && e[0] == '0');
    inst->lds_chain_refuse = e && e[0] == 'r'; /* "refuse": the chain is attempted and declines (tests of the fallback) */
    e = getenv("VKSIFT_LDS_CHAIN_MAX");
    inst->lds_chain_max = e ? (uint32_t)strtoul(e, NULL, 10) : 4800u;
    if (inst->lds_chain_max > 19200u)
      inst->lds_chain_max = 19200u;
    /* hipGraph capture + replay of the detection launch sequence. Measured on MI355X / ROCm 7.2: 10 % faster for one
     * 640x480 image (0.58 vs 0.65 ms), 12 % slower from 1536x1024 up (the graph runs the per-octave branches less
     * concurrently than the streams do) -> by default only small workloads are replayed (graph_max_pixels).
     * VKSIFT_GRAPH=0 never, =1 always. */
    e = getenv("VKSIFT_GRAPH");
    inst->use_graphs = !(e && e[0] == '0');
    inst->graph_max_pixels = (e && e[0] == '1') ? ~(uint64_t)0 : (uint64_t)640 * 480;
    e = getenv("VKSIFT_POST_FEATURES");
    inst->post_enabled = inst->post_on = !(e && e[0] == '0');
  }
  for (int i = 0; i < VKSIFT_DETECT_RING; i++)
    inst->det_ring[i].ev = vksift_hip_event_create();
  inst->ev_match = vksift_hip_event_create();
  inst->ev_staging = vksift_hip_event_create();
  for (uint32_t g = 0; g < VKSIFT_UP_GROUPS; g++)
    inst->ev_up[g] = vksift_hip_event_create();
  for (int i = 0; i < 8; i++)
  {
    inst->prof[0].ev_t[i] = vksift_hip_event_create();
    inst->prof[1].ev_t[i] = vksift_hip_event_create();
  }
  for (int i = 0; i < 3; i++)
  {
    inst->prof[0].ev_pt[i] = vksift_hip_event_create();
    inst->prof[1].ev_pt[i] = vksift_hip_event_create();
  }
  inst->prof[0].ev_scan = vksift_hip_event_create();
  inst->prof[1].ev_scan = vksift_hip_event_create();
  inst->ev_m[0] = vksift_hip_event_create();
  inst->ev_m[1] = vksift_hip_event_create();
  ok = ok && inst->stream && inst->det_ring[0].ev && inst->det_ring[VKSIFT_DETECT_RING - 1].ev && inst->ev_match && inst->pyr_stream && inst->dl_stream && inst->up_stream;
  if (!ok)
  {
    logError(LOG_TAG, "vksift_createInstance() failed: device / pinned memory reservation");
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
    set_buffer_sections(inst, b, L.n_oct, side, side); /* seq 0: nothing pending, the (zeroed) counters are valid */

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
  inst->pend_n = 0; /* staged, never asked for: nobody can see the result of launching them */
  if (inst->pyr_stream)
    vksift_hip_stream_sync(inst->pyr_stream);
  if (inst->side_stream)
    vksift_hip_stream_sync(inst->side_stream);
  if (inst->dl_stream)
    vksift_hip_stream_sync(inst->dl_stream);
  if (inst->up_stream)
    vksift_hip_stream_sync(inst->up_stream);
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
  vksift_hip_free(inst->d_cache_desc);
  vksift_hip_free(inst->d_cache_norm);
  vksift_hip_free(inst->d_cache_n);
  free(inst->cache_valid);
  free(inst->cache_queued);
  vksift_hip_free(inst->d_matches);
  vksift_hip_free(inst->d_redo);
  vksift_hip_free(inst->d_match_n);
  vksift_hip_free(inst->d_match_partial);
  vksift_hip_free(inst->d_dl);
  vksift_hip_host_free(inst->h_dl);
  vksift_hip_host_free(inst->h_post[0]);
  vksift_hip_host_free(inst->h_post[1]);
  free(inst->dl_row);
  for (uint32_t k = 0; k < VKSIFT_DL_CHUNKS; k++)
    vksift_hip_event_destroy(inst->dl_ev[k]);
  for (int i = 0; i < VKSIFT_GRAPH_CACHE; i++)
    vksift_hip_graph_destroy(inst->graphs[i].exec);
  vksift_hip_free(inst->rev.matches);
  vksift_hip_free(inst->rev.redo);
  vksift_hip_free(inst->rev.match_n);
  vksift_hip_free(inst->d_filtered);
  vksift_hip_free(inst->d_filtered_n);
  vksift_hip_host_free(inst->h_filtered_n);
  vksift_hip_host_free(inst->h_match_n);
  vksift_hip_host_free(inst->h_matches);
  free(inst->bufs);
  free(inst->match_busy);
  for (int i = 0; i < VKSIFT_DETECT_RING; i++)
    vksift_hip_event_destroy(inst->det_ring[i].ev);
  vksift_hip_event_destroy(inst->ev_match);
  vksift_hip_event_destroy(inst->ev_staging);
  for (uint32_t g = 0; g < VKSIFT_UP_GROUPS; g++)
    vksift_hip_event_destroy(inst->ev_up[g]);
  for (int i = 0; i < 8; i++)
  {
    vksift_hip_event_destroy(inst->prof[0].ev_t[i]);
    vksift_hip_event_destroy(inst->prof[1].ev_t[i]);
  }
  vksift_hip_event_destroy(inst->ev_m[0]);
  vksift_hip_event_destroy(inst->ev_m[1]);
  for (int i = 0; i < VKSIFT_MAX_OCTAVES; i++)
    vksift_hip_event_destroy(inst->ev_fork[i]);
  vksift_hip_event_destroy(inst->ev_join[0]);
  vksift_hip_event_destroy(inst->ev_join[1]);
  if (inst->side_stream)
  {
    vksift_hip_stream_sync(inst->side_stream);
    vksift_hip_stream_destroy(inst->side_stream);
  }
  vksift_hip_stream_destroy(inst->pyr_stream);
  vksift_hip_stream_destroy(inst->dl_stream);
  vksift_hip_stream_destroy(inst->up_stream);
  vksift_hip_event_destroy(inst->ev_pyr_done);
  vksift_hip_event_destroy(inst->ev_desc_start);
  vksift_hip_event_destroy(inst->ev_input_free);
  vksift_hip_event_destroy(inst->prof[0].ev_scan);
  vksift_hip_event_destroy(inst->prof[1].ev_scan);
  for (int i = 0; i < 2; i++)
  {
    vksift_hip_event_destroy(inst->ev_pyr_free[i]);
    vksift_hip_event_destroy(inst->prof[0].ev_pt[i]);
    vksift_hip_event_destroy(inst->prof[1].ev_pt[i]);
  }
  vksift_hip_event_destroy(inst->prof[0].ev_pt[2]);
  vksift_hip_event_destroy(inst->prof[1].ev_pt[2]);
  vksift_hip_stream_destroy(inst->stream);
  free(inst);
  *instance_ptr = NULL;
}

/* ------------------------------------------------------------------------------------------------ */
/* synchronisation helpers (fences of the reference)                                                */
/* ------------------------------------------------------------------------------------------------ */
/* The stream is in-order: once a detection has completed, every earlier one has too, and the host mirrors of their counters
 * are valid (counts_valid()). */
void mark_detect_done(vksift_Instance inst) { inst->det_done = inst->det_seq; }

/* polls the detections in flight; true while the LATEST one is still running */
bool detect_running(vksift_Instance inst)
{
  if (inst->det_done >= inst->det_seq)
    return false;
  for (int i = 0; i < VKSIFT_DETECT_RING; i++)
  {
    const DetectSlot *d = &inst->det_ring[i];
    if (d->seq > inst->det_done && vksift_hip_event_busy(d->ev) != 1)
      inst->det_done = d->seq;
  }
  return inst->det_done < inst->det_seq;
}

/* blocks until detection `seq` has completed. Its ring slot may have been taken over by a later detection (more than
 * VKSIFT_DETECT_RING in flight): waiting for that one is conservative, never wrong. */
int wait_detect_seq(vksift_Instance inst, uint64_t seq)
{
  if (seq <= inst->det_done)
    return 0;
  const DetectSlot *d = &inst->det_ring[seq % VKSIFT_DETECT_RING];
  const int e = vksift_hip_event_sync(d->ev);
  if (d->seq > inst->det_done)
    inst->det_done = d->seq;
  return e;
}

bool match_running(vksift_Instance inst)
{
  if (!inst->match_pending)
    return false;
  if (vksift_hip_event_busy(inst->ev_match) == 1)
    return true;
  inst->match_pending = false;
  memset(inst->match_busy, 0, sizeof(bool) * inst->cfg.sift_buffer_count);
  return false;
}
/* ------------------------------------------------------------------------------------------------ */
/* Where in HBM the scale-space lives (round 5; DESIGN.md §8, tools/microbench/stream_patterns.hip)    */
/* ------------------------------------------------------------------------------------------------ */
/* The strip-march launches of pyramid.hip and the extrema scan — thousands of waves each streaming its own row segment — run at
 * 4.9-5.0 TB/s on some ranges of the device's memory and at 5.9-6.1 TB/s on others, for the SAME kernel, sizes and strides: measured
 * with a pure copy in that access pattern sliding over a 240 GiB allocation of an idle MI355X, the first ~40 GB of a fresh process's
 * memory and a few later windows are the slow ones, ~75-170 GB the fast plateau; a linear copy runs at 6.2 TB/s everywhere. A fresh
 * process gets the low range first, so an instance that simply allocates its scale-space takes the slow memory. The two
 * scale-space buffers of a batch instance are therefore chosen by measurement: allocate a candidate, time one whole-batch blur
 * launch of octave 0 on it (the pattern that matters, 1 warm-up + 3 runs of ~1 ms), keep it, allocate the next — rejected candidates
 * stay allocated while the search runs, so that the allocator has to hand out new ranges — until `need` candidates run within
 * VKSIFT_PLACE_SPREAD of the best AND a slower range has been seen (the fast mode is identified), or everything looks alike, or the
 * candidate / memory budget is used up; then every candidate but the best `need` is freed. Nothing depends on it but speed.
 * VKSIFT_PYR_PLACEMENT=<max candidates> (default 7; 0 or 1: plain allocation). */
#define VKSIFT_PLACE_MAX 8
#define VKSIFT_PLACE_SPREAD 1.04f
static float placement_probe_ms(vksift_Instance inst, void *buf, uint64_t img_stride, const PyrLayout *L, vksift_hip_event e0, vksift_hip_event e1)
{
  vksift_hip_Plane src, dst;
  src.base = (float *)((uint8_t *)buf + L->gauss_off[0] * pyr_texel_bytes(inst));
  src.fp16 = inst->fp16 ? 1u : 0u, src.reverse = 0;
  /* (a width the strip-march kernels take — the reservation's square layout may have one they leave to the generic tile kernel) */
  src.w = L->w[0] >= 512u ? (L->w[0] & ~255u) : (L->w[0] & ~3u);
  src.h = L->h[0], src.pitch = L->pitch[0], src.img_stride = img_stride;
  dst = src;
  dst.base = (float *)((uint8_t *)buf + (L->gauss_off[0] + L->plane_stride[0]) * pyr_texel_bytes(inst));
  float best = -1.f;
  for (int r = 0; r < 4; r++)
  {
    dst.reverse = (uint32_t)(r & 1);
    if (vksift_hip_event_record(e0, inst->stream) != 0 ||
        vksift_hip_blur(src, dst, &inst->taps[1 * VKSIFT_MAX_TAPS], inst->ntaps[1], inst->det_cap, inst->stream) != 0 ||
        vksift_hip_event_record(e1, inst->stream) != 0 || vksift_hip_event_sync(e1) != 0)
      return -1.f;
    const float ms = vksift_hip_event_elapsed_ms(e0, e1);
    if (r > 0 && ms > 0.f && (best < 0.f || ms < best))
      best = ms;
  }
  return best;
}

/* out[0 .. need): device blocks of `bytes` each for a scale-space of layout L (octave 0 is what gets timed); false: out of memory
 * (nothing is left allocated). may_search = false: plain allocation (re-allocations in the middle of a caller's detect call).
 * The rejected candidates of a search stay allocated while it runs — freed, the allocator would hand the same range out again —
 * so the search is bounded: the candidates together never hold more than VKSIFT_PLACE_MEM_FRACTION (55 %) of the memory that was
 * free when it started, and 24 GB stay free for the rest of the instance and for whoever else uses the device. */
static bool place_pyramid_buffers(vksift_Instance inst, size_t bytes, uint64_t img_stride, const PyrLayout *L, uint32_t need, float **out, bool may_search)
{
  int max_cand = 7;
  {
    const char *e = getenv("VKSIFT_PYR_PLACEMENT");
    if (e)
      max_cand = atoi(e);
    if (max_cand > VKSIFT_PLACE_MAX)
      max_cand = VKSIFT_PLACE_MAX;
  }
  inst->place_n = 0;
  vksift_hip_event e0 = NULL, e1 = NULL;
  const bool search = may_search && max_cand > (int)need && inst->det_cap >= 8u && bytes >= ((size_t)256 << 20) && L->n_oct > 0 && inst->stream != NULL &&
                      (e0 = vksift_hip_event_create()) != NULL && (e1 = vksift_hip_event_create()) != NULL;
  const size_t budget = search ? (size_t)((double)vksift_hip_device_free_mem() * 0.55) : 0;
  void *cand[VKSIFT_PLACE_MAX] = {NULL};
  float ms[VKSIFT_PLACE_MAX];
  uint32_t n = 0;
  bool ok = true;
  while (n < need || (search && n < (uint32_t)max_cand))
  {
    if (n >= need)
    {
      /* another candidate only within the budget, and while the device still has room for it and the rest of an instance */
      if ((size_t)(n + 1u) * bytes > budget || vksift_hip_device_free_mem() < bytes + ((size_t)24 << 30))
        break;
      /* stop rules (sorted view of what has been timed) */
      float lo = ms[0], hi = ms[0];
      uint32_t near_best = 0;
      for (uint32_t i = 0; i < n; i++)
        lo = ms[i] < lo ? ms[i] : lo, hi = ms[i] > hi ? ms[i] : hi;
      for (uint32_t i = 0; i < n; i++)
        near_best += ms[i] <= lo * VKSIFT_PLACE_SPREAD ? 1u : 0u;
      if (near_best >= need && hi > lo * 1.08f)
        break; /* the fast mode has been seen `need` times, and a slow one beside it */
      if (n >= need + 5u && hi <= lo * VKSIFT_PLACE_SPREAD)
        break; /* this memory is all alike */
    }
    void *p = vksift_hip_malloc(bytes);
    if (!p)
    {
      ok = n >= need;
      break;
    }
    cand[n] = p;
    ms[n] = search ? placement_probe_ms(inst, p, img_stride, L, e0, e1) : 0.f;
    if (search && ms[n] <= 0.f)
      ms[n] = 1e9f; /* the probe failed: last choice */
    n++;
  }
  if (ok && n >= need)
  {
    /* the `need` fastest, the rest goes back */
    for (uint32_t k = 0; k < need; k++)
    {
      uint32_t b = 0;
      for (uint32_t i = 0; i < n; i++)
        if (cand[i] && (!cand[b] || ms[i] < ms[b]))
          b = i;
      out[k] = (float *)cand[b];
      inst->place_chosen[k] = b;
      cand[b] = NULL;
    }
    if (need == 1u)
      inst->place_chosen[1] = inst->place_chosen[0];
    float slowest = 0.f;
    for (uint32_t i = 0; i < n; i++)
    {
      const double px = (double)(L->w[0] >= 512u ? (L->w[0] & ~255u) : (L->w[0] & ~3u)) * L->h[0] * inst->det_cap * 2.0 * (double)pyr_texel_bytes(inst);
      inst->place_gbps[i] = (search && ms[i] < 1e8f) ? (float)(px / (ms[i] * 1e-3) / 1e9) : 0.f;
      if (inst->place_gbps[i] > 0.f && (slowest == 0.f || inst->place_gbps[i] < slowest))
        slowest = inst->place_gbps[i];
    }
    inst->place_n = search ? n : 0;
    if (search)
      logInfo(LOG_TAG, "scale-space placement: %u candidate range(s) of %.1f GB timed, chosen %.0f GB/s, slowest %.0f GB/s", n, bytes / 1e9,
              inst->place_gbps[inst->place_chosen[0]], slowest);
  }
  else
    ok = false;
  for (uint32_t i = 0; i < n; i++)
    vksift_hip_free(cand[i]); /* NULL for the chosen ones */
  vksift_hip_event_destroy(e0);
  vksift_hip_event_destroy(e1);
  if (!ok)
    for (uint32_t k = 0; k < need; k++)
      out[k] = NULL;
  return ok;
}

/* The reservation made at creation covers `det_cap` square images of input_image_max_size pixels plus 25 %. Two things outgrow it:
 * a narrow image of the same area (every row is padded to 64 floats on every octave: 139x356 needs 1.5x; the reference re-creates its
 * images for every new input resolution, sift_memory.c:362-452) — L != NULL, the per-image strides grow to what the layout needs —
 * and a caller of the plain API who batches (deferred submission) — new_cap > det_cap, the blocks grow to new_cap images.
 * All work of the instance is drained first; captured launch graphs hold the old addresses and are dropped.
 * 0: done. 1: the larger capacity did not fit, the instance is as it was (capacity growth only). -1: out of device memory, the instance
 * holds NO detection scratch (capacities 0): the next detection retries the allocation or fails cleanly with VKSIFT_VULKAN_ERROR. */
static void free_detect_scratch(vksift_Instance inst, bool cap_blocks)
{
  vksift_hip_free(inst->d_pyr_buf[0]);
  vksift_hip_free(inst->d_pyr_buf[1]);
  vksift_hip_free(inst->d_seg_mask);
  vksift_hip_free(inst->d_seg_off);
  vksift_hip_free(inst->d_cand_xy);
  vksift_hip_free(inst->d_cand_flag);
  inst->d_pyr_buf[0] = inst->d_pyr_buf[1] = inst->d_pyr = NULL;
  inst->d_seg_mask = NULL, inst->d_seg_off = NULL, inst->d_cand_xy = NULL, inst->d_cand_flag = NULL;
  if (cap_blocks)
  {
    vksift_hip_free(inst->d_input);
    vksift_hip_host_free(inst->h_input);
    vksift_hip_free(inst->d_cand_n);
    vksift_hip_free(inst->d_ori_ang);
    vksift_hip_free(inst->d_ori_cnt);
    inst->d_input = inst->h_input = NULL;
    inst->d_cand_n = NULL, inst->d_ori_ang = NULL, inst->d_ori_cnt = NULL;
  }
}

static bool alloc_detect_scratch(vksift_Instance inst, const PyrLayout *L, uint32_t cap, uint64_t pyr, uint64_t seg, uint64_t cand, bool cap_blocks, bool may_search)
{
  const uint64_t n = cap;
  const uint32_t old_cap = inst->det_cap;
  inst->det_cap = cap; /* the placement probe launches on `det_cap` images */
  bool ok = place_pyramid_buffers(inst, pyr_texel_bytes(inst) * pyr * n, pyr, L, inst->pyr_nbuf, inst->d_pyr_buf, may_search);
  inst->det_cap = old_cap;
#define GROW_D(ptr, bytes) ok = ok && ((ptr = vksift_hip_malloc(bytes)) != NULL)
  GROW_D(inst->d_seg_mask, sizeof(uint64_t) * seg * n);
  GROW_D(inst->d_seg_off, sizeof(uint32_t) * seg * n);
  GROW_D(inst->d_cand_xy, sizeof(uint32_t) * cand * n);
  GROW_D(inst->d_cand_flag, sizeof(uint32_t) * cand * n);
  if (cap_blocks)
  {
    GROW_D(inst->d_input, (size_t)inst->max_image_size * n);
    ok = ok && (inst->h_input = vksift_hip_host_malloc((size_t)inst->max_image_size * n)) != NULL;
    GROW_D(inst->d_cand_n, sizeof(uint32_t) * n * VKSIFT_MAX_OCTAVES);
    GROW_D(inst->d_ori_ang, sizeof(float) * VKSIFT_HIP_MAX_ORI * inst->ori_cap * n);
    GROW_D(inst->d_ori_cnt, sizeof(uint32_t) * inst->ori_cap * n);
  }
#undef GROW_D
  return ok;
}

int resize_detect_scratch(vksift_Instance inst, const PyrLayout *L, uint32_t new_cap)
{
  assert(inst->pend_n == 0);
  if (wait_all(inst) != 0)
    return -1;
  if (inst->pyr_stream)
    vksift_hip_stream_sync(inst->pyr_stream);
  if (inst->side_stream)
    vksift_hip_stream_sync(inst->side_stream);
  if (inst->up_stream)
    vksift_hip_stream_sync(inst->up_stream);
  inst->staging_pending = false;
  inst->input_free_valid = false;
  for (int i = 0; i < VKSIFT_GRAPH_CACHE; i++)
  {
    vksift_hip_graph_destroy(inst->graphs[i].exec);
    memset(&inst->graphs[i], 0, sizeof(inst->graphs[i]));
  }
  PyrLayout cur;
  if (!L)
  {
    /* capacity growth alone: the layout the probe launch runs on is the reservation's */
    compute_layout(inst, inst->cur_w ? inst->cur_w : (uint32_t)ceilf(sqrtf((float)inst->cfg.input_image_max_size)),
                   inst->cur_h ? inst->cur_h : (uint32_t)ceilf(sqrtf((float)inst->cfg.input_image_max_size)), &cur);
  }
  const PyrLayout *PL = L ? L : &cur;
  const uint64_t pyr = L ? L->img_floats + L->img_floats / 4 + 4096 : 0, seg = L ? L->seg_total + L->seg_total / 4 + 1024 : 0,
                 cand = L ? L->cand_total + L->cand_total / 4 + 4096u : 0;
  const uint64_t new_pyr = pyr > inst->pyr_img_stride ? pyr : inst->pyr_img_stride;
  const uint64_t new_seg = seg > inst->seg_cap ? seg : inst->seg_cap, new_cand = cand > inst->cand_cap ? cand : inst->cand_cap;
  const uint32_t old_cap = inst->det_cap;
  const bool cap_blocks = new_cap != old_cap || inst->d_input == NULL; /* (NULL: lost by an earlier attempt that ran out of memory) */
  /* old blocks first: the pyramid is the largest allocation of the instance, two generations of it may not fit */
  free_detect_scratch(inst, cap_blocks);
  inst->pyr_free_valid[0] = inst->pyr_free_valid[1] = false;
  inst->cur_w = inst->cur_h = 0; /* no scale-space to download until the next detection */
  int rc = 0;
  /* (a capacity growth happens once per size, outside any detection that runs: it may search for fast memory; a stride growth sits in
   * the middle of a detect call of whatever the caller is doing and takes plain allocations) */
  bool ok = alloc_detect_scratch(inst, PL, new_cap, new_pyr, new_seg, new_cand, cap_blocks, cap_blocks);
  uint32_t cap = new_cap;
  if (!ok && cap_blocks)
  {
    /* the larger capacity does not fit: back to the one the instance had */
    free_detect_scratch(inst, true);
    ok = alloc_detect_scratch(inst, PL, old_cap, new_pyr, new_seg, new_cand, true, false);
    cap = old_cap;
    rc = 1;
  }
  if (!ok)
  {
    free_detect_scratch(inst, cap_blocks);
    inst->pyr_img_stride = 0, inst->seg_cap = 0, inst->cand_cap = 0;
    return -1;
  }
  inst->pyr_img_stride = new_pyr, inst->seg_cap = new_seg, inst->cand_cap = new_cand;
  inst->det_cap = cap;
  inst->d_pyr = inst->d_pyr_buf[inst->pyr_nbuf == 2u ? inst->pyr_cur : 0];
  if (!inst->overlap_forced && inst->batch_cap < 8u && cap >= 8u && !inst->pyr_pingpong)
  {
    /* a plain instance that now takes batches: they overlap like a batch instance's, its small detections stay as they were */
    inst->pyr_pingpong = true;
    inst->overlap_min_count = 8u;
  }
  return rc;
}

int grow_image_scratch(vksift_Instance inst, const PyrLayout *L) { return resize_detect_scratch(inst, L, inst->det_cap) == 0 ? 0 : -1; }

int wait_all(vksift_Instance inst)
{
  vksift_hip_set_device(inst->device);
  int e = vksift_hip_stream_sync(inst->stream);
  mark_detect_done(inst);
  inst->match_pending = false;
  memset(inst->match_busy, 0, sizeof(bool) * inst->cfg.sift_buffer_count);
  return e;
}

bool vksift_isBufferAvailable(vksift_Instance instance, const uint32_t gpu_buffer_id)
{
  vksift_hip_set_device(instance->device);
  defer_sync(instance);
  if (gpu_buffer_id >= instance->cfg.sift_buffer_count)
    return true;
  (void)detect_running(instance);
  if (!counts_valid(instance, gpu_buffer_id))
    return false;
  if (match_running(instance) && instance->match_busy[gpu_buffer_id])
    return false;
  return true;
}

