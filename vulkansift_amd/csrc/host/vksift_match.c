/*
 * vksift_match.c — the matching pipeline (vulkansift.c:417-462, sift_memory.c:957-1058, sift_matcher.c:246-279) and GPU-side match filtering
 */
#include "vksift_internal.h"

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

/* Launch bound on the number of stored rows of a buffer: the real total if its counters have reached the host, else the
 * sum of its section capacities. */
static uint32_t rows_bound(vksift_Instance inst, uint32_t id)
{
  const BufferInfo *b = &inst->bufs[id];
  if (b->nb_sections == 0)
    return b->nb_stored;
  uint32_t known = 0, cap_sum = 0;
  const uint32_t *found = inst->h_found + (size_t)id * VKSIFT_MAX_OCTAVES;
  for (uint32_t o = 0; o < b->nb_sections; o++)
  {
    known += found[o] < b->sec_cap[o] ? found[o] : b->sec_cap[o];
    cap_sum += b->sec_cap[o];
  }
  return counts_valid(inst, id) ? known : cap_sum;
}

/* The matcher's input of a SIFT buffer — dense 128-byte descriptor rows in download order, their shifted norms and the row
 * count — lives in a per-buffer cache entry, filled by ONE device-side gather the first time the buffer is matched after it
 * changed (detection, upload) and reused by every later matching: a self-match gathers once instead of twice, the two
 * directions of a pair matching gather nothing the second time. The reference physically packs the octave sections
 * (pack_BufferMemory, sift_memory.c:957-1047) after reading the counts on the host; here the gather kernel reads the
 * counters in HBM and walks the sections in the same order, so nothing waits on the host. Rows below 2 are zero-filled:
 * Get2NearestNeighbors.comp:66-67 reads b[0] and b[1] unconditionally (stale memory in the reference when B holds fewer
 * than two features); here the missing rows are defined as all-zero descriptors. */
/* first matching / descriptor export of the instance: the cache blocks (see create_instance). A failed allocation is retried by
 * the next call; nothing is leaked in between. */
static int ensure_match_cache(vksift_Instance inst)
{
  if (!inst->d_cache_desc)
    inst->d_cache_desc = (uint8_t *)vksift_hip_malloc(inst->desc_slot_stride * inst->cfg.sift_buffer_count);
  if (!inst->d_cache_norm)
    inst->d_cache_norm = (uint32_t *)vksift_hip_malloc(sizeof(uint32_t) * inst->cache_norm_stride * inst->cfg.sift_buffer_count);
  if (!inst->d_match_partial && inst->cfg.max_nb_sift_per_buffer > VKSIFT_HIP_MATCH_SMALL_NA)
  {
    inst->d_match_partial = (uint32_t *)vksift_hip_malloc(sizeof(uint32_t) * (size_t)inst->cfg.max_nb_sift_per_buffer * 5u * VKSIFT_HIP_MATCH_CHUNKS);
    if (!inst->d_match_partial)
      return 2; /* hipErrorOutOfMemory */
  }
  return (inst->d_cache_desc && inst->d_cache_norm) ? 0 : 2;
}

int refresh_match_cache(vksift_Instance inst, const uint32_t *ids, uint32_t count)
{
  const int ce = ensure_match_cache(inst);
  if (ce)
    return ce;
  (void)detect_running(inst); /* polls the detections in flight: counts_valid() is up to date */
  /* Passes of at most VKSIFT_HIP_GATHER_SLOTS not-yet-cached buffers until none is left: a run of VKSIFT_HIP_MATCH_SLOTS (256) pairs may
   * name that many distinct buffers per side, and every one of them must be gathered before the matching kernels read its cache entry. */
  for (;;)
  {
    uint32_t todo[VKSIFT_HIP_GATHER_SLOTS];
    uint32_t n = 0;
    bool more = false;
    for (uint32_t i = 0; i < count; i++)
    {
      if (inst->cache_valid[ids[i]] || inst->cache_queued[ids[i]])
        continue;
      if (n == VKSIFT_HIP_GATHER_SLOTS)
      {
        more = true; /* taken by the next pass */
        break;
      }
      inst->cache_queued[ids[i]] = true;
      todo[n++] = ids[i];
    }
    for (uint32_t k = 0; k < n; k++)
      inst->cache_queued[todo[k]] = false;
    uint32_t i0 = 0;
    while (i0 < n)
    {
      /* one launch per run of buffers that share a section layout (always all of them after a batched detection) */
      const BufferInfo *b = &inst->bufs[todo[i0]];
      uint32_t i1 = i0 + 1, max_rows = rows_bound(inst, todo[i0]);
      while (i1 < n && same_layout(b, &inst->bufs[todo[i1]]))
      {
        const uint32_t r = rows_bound(inst, todo[i1]);
        max_rows = r > max_rows ? r : max_rows;
        i1++;
      }
      int e;
      if (b->nb_sections == 0)
      {
        uint32_t zero_off = 0, cap1 = b->nb_stored, fixed1 = b->nb_stored;
        e = vksift_hip_gather_sections(inst->d_feats, inst->buf_stride, todo + i0, i1 - i0, 1, &zero_off, &cap1, &fixed1, NULL, 0, max_rows, 2u,
                                       inst->d_cache_desc, inst->desc_slot_stride, inst->d_cache_norm, inst->cache_norm_stride, inst->d_cache_n, 1, inst->stream);
      }
      else
        e = vksift_hip_gather_sections(inst->d_feats, inst->buf_stride, todo + i0, i1 - i0, b->nb_sections, b->sec_off, b->sec_cap, NULL, inst->d_found,
                                       VKSIFT_MAX_OCTAVES, max_rows, 2u, inst->d_cache_desc, inst->desc_slot_stride, inst->d_cache_norm, inst->cache_norm_stride,
                                       inst->d_cache_n, 1, inst->stream);
      if (e)
        return e;
      for (uint32_t k = i0; k < i1; k++)
        inst->cache_valid[todo[k]] = true;
      i0 = i1;
    }
    if (!more)
      return 0;
  }
}

MatchScratch fwd_scratch(vksift_Instance inst)
{
  MatchScratch ms = {inst->d_matches, inst->d_redo, inst->d_match_n};
  return ms;
}

static int match_slots(vksift_Instance inst, const MatchScratch *ms, const uint32_t *ids_a, const uint32_t *ids_b, uint32_t count, uint32_t first_slot)
{
  int e = refresh_match_cache(inst, ids_a, count);
  if (e == 0)
    e = refresh_match_cache(inst, ids_b, count);
  if (e)
    return e;
  uint32_t max_na = 0, max_nb = 0;
  bool nb_exact = true; /* every reference buffer's count is known on the host (uploaded, or its detection has completed) */
  for (uint32_t i = 0; i < count; i++)
  {
    const uint32_t r = rows_bound(inst, ids_a[i]), rb = rows_bound(inst, ids_b[i]);
    max_na = r > max_na ? r : max_na;
    max_nb = rb > max_nb ? rb : max_nb;
    nb_exact = nb_exact && (inst->bufs[ids_b[i]].nb_sections == 0 || counts_valid(inst, ids_b[i]));
  }
  return vksift_hip_match_2nn_async(inst->d_cache_desc, inst->d_cache_norm, inst->d_cache_n, ids_a, ids_b, max_na, max_nb, nb_exact ? 1u : 0u,
                                    ms->redo + (uint64_t)first_slot * inst->redo_slot_stride, ms->match_n + (size_t)first_slot * 4,
                                    ms->matches + (uint64_t)first_slot * inst->match_slot_stride, count, inst->desc_slot_stride, inst->cache_norm_stride,
                                    inst->redo_slot_stride, inst->match_slot_stride, 4, inst->d_match_partial, inst->stream);
}

/* reverse-matching scratch + survivor lists of vksift_ext_matchFeaturesFiltered, allocated on first use */
static bool ensure_filter_scratch(vksift_Instance inst)
{
  const uint32_t bc = inst->batch_cap;
  inst->filtered_slot_stride = (((uint64_t)inst->cfg.max_nb_sift_per_buffer * 16u) + 255u) & ~(uint64_t)255u;
  /* each block only if it does not exist yet: a call that ran out of memory half-way is retried without leaking */
  bool ok = true;
#define ENSURE_D(ptr, bytes) ok = ok && ((ptr) != NULL || ((ptr) = vksift_hip_malloc(bytes)) != NULL)
  ENSURE_D(inst->rev.matches, inst->match_slot_stride * bc);
  ENSURE_D(inst->rev.redo, sizeof(uint32_t) * inst->redo_slot_stride * bc);
  ENSURE_D(inst->rev.match_n, sizeof(uint32_t) * 4 * bc);
  ENSURE_D(inst->d_filtered_n, sizeof(uint32_t) * bc);
  ENSURE_D(inst->d_filtered, inst->filtered_slot_stride * bc);
#undef ENSURE_D
  ok = ok && (inst->h_filtered_n != NULL || (inst->h_filtered_n = vksift_hip_host_malloc(sizeof(uint32_t) * bc)) != NULL);
  return ok;
}

static void match_impl(vksift_Instance inst, const uint32_t *ids_a, const uint32_t *ids_b, uint32_t count, const char *fn, bool filter, float ratio,
                       bool cross_check)
{
  bool range_open = false;
  vksift_hip_set_device(inst->device);
  defer_sync(inst);
  bool valid = count >= 1 && count <= inst->batch_cap;
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
  range_open = true;
  const MatchScratch fwd = fwd_scratch(inst);
  /* the matcher's view of every buffer of the call first, in one gather launch per 512 buffers (each run below would gather its own) */
  HIP_CHECK(refresh_match_cache(inst, ids_a, count), "descriptor gather");
  HIP_CHECK(refresh_match_cache(inst, ids_b, count), "descriptor gather");
  /* one launch sequence serves up to VKSIFT_HIP_MATCH_SLOTS pairs; a longer list goes in runs of that many, run r into the slots from r on */
  for (uint32_t r = 0; r < count; r += VKSIFT_HIP_MATCH_SLOTS)
    HIP_CHECK(match_slots(inst, &fwd, ids_a + r, ids_b + r, count - r < VKSIFT_HIP_MATCH_SLOTS ? count - r : VKSIFT_HIP_MATCH_SLOTS, r), "2-NN matching");
  HIP_CHECK(vksift_hip_post_words(inst->h_match_n, inst->d_match_n, (size_t)4 * count, inst->stream), "match count read-back");
  if (inst->desc_start_valid && vksift_hip_tune_get(VKSIFT_TUNE_PYR_GATE) == 1)
    vksift_hip_event_record(inst->ev_desc_start, inst->stream); /* experiment: the next scale-space behind this matching, not beside it */
  inst->filtered_slots_used = 0;
  inst->md_valid = false, inst->md_hits = 0, inst->md_direct = false, inst->md_asked = false;
  if (filter)
  {
    /* the reverse matching, then cross-check + ratio test on the device; only the survivors are read back */
    if (!ensure_filter_scratch(inst))
    {
      logError(LOG_TAG, "%s error: out of device memory for the filtered-matching scratch.", fn);
      goto gpu_error;
    }
    for (uint32_t r = 0; r < count; r += VKSIFT_HIP_MATCH_SLOTS)
    {
      const uint32_t n = count - r < VKSIFT_HIP_MATCH_SLOTS ? count - r : VKSIFT_HIP_MATCH_SLOTS;
      if (cross_check)
        HIP_CHECK(match_slots(inst, &inst->rev, ids_b + r, ids_a + r, n, r), "reverse 2-NN matching");
      HIP_CHECK(vksift_hip_filter_matches(inst->d_matches + (uint64_t)r * inst->match_slot_stride, inst->match_slot_stride,
                                          cross_check ? inst->rev.matches + (uint64_t)r * inst->match_slot_stride : NULL, inst->match_slot_stride,
                                          inst->d_match_n + (size_t)r * 4, 4, ratio, n, inst->d_filtered + (uint64_t)r * inst->filtered_slot_stride,
                                          inst->filtered_slot_stride, inst->d_filtered_n + r, inst->stream),
                "match filtering");
    }
    HIP_CHECK(vksift_hip_post_words(inst->h_filtered_n, inst->d_filtered_n, count, inst->stream), "filtered count read-back");
    inst->filtered_slots_used = count;
  }
  vksift_hip_range_pop();
  range_open = false;
  if (inst->profiling)
  {
    vksift_hip_event_record(inst->ev_m[1], inst->stream);
    inst->match_timing_valid = true;
  }
  (void)match_running(inst); /* an earlier matching that has completed releases its buffers; one still in flight keeps them */
  HIP_CHECK(vksift_hip_event_record(inst->ev_match, inst->stream), "event record");
  inst->match_pending = true;
  inst->match_slots_used = count;
  for (uint32_t i = 0; i < count; i++)
    inst->match_busy[ids_a[i]] = inst->match_busy[ids_b[i]] = true; /* every pair of a batched call (vksift_isBufferAvailable) */
  return;
gpu_error:
  if (range_open)
    vksift_hip_range_pop();
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
  defer_sync(inst);
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

/* all pairs of the last matching -> pinned memory (row pitch = the longest list), then `pair` out of it; false: not available */
static bool packed_match_download(vksift_Instance inst, uint32_t pair, vksift_Match_2NN *matches, uint32_t n)
{
  if (!inst->md_valid)
  {
    uint32_t max_n = 0;
    for (uint32_t i = 0; i < inst->match_slots_used; i++)
      max_n = inst->h_match_n[(size_t)i * 4] > max_n ? inst->h_match_n[(size_t)i * 4] : max_n;
    const size_t pitch = (size_t)max_n * MATCH_BYTES, bytes = pitch * inst->match_slots_used;
    if (bytes > inst->md_cap)
    {
      vksift_hip_host_free(inst->h_matches);
      inst->h_matches = (uint8_t *)vksift_hip_host_malloc(bytes + bytes / 4u + 4096u);
      inst->md_cap = inst->h_matches ? bytes + bytes / 4u + 4096u : 0;
      if (!inst->h_matches)
        return false;
    }
    if (vksift_hip_memcpy2d_d2h(inst->h_matches, pitch, inst->d_matches, inst->match_slot_stride, pitch, inst->match_slots_used, inst->dl_stream) != 0 ||
        vksift_hip_stream_sync(inst->dl_stream) != 0)
      return false;
    inst->md_pitch = pitch;
    inst->md_valid = true;
  }
  memcpy(matches, inst->h_matches + (size_t)pair * inst->md_pitch, (size_t)n * MATCH_BYTES);
  return true;
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
    /* A caller that walks the pairs of a batched matching gets them out of ONE strided copy into pinned memory instead of one
     * copy + synchronisation per pair (12 us each: 1.5 ms per 128 pairs). Like the packed feature download it starts with the
     * second download after a matching: a caller that samples one pair must not pay for all of them. */
    /* (a page-locked destination takes the records by DMA straight from the slot: vksift_ext_pinHostMemory) */
    if (inst->match_slots_used >= VKSIFT_DL_BATCH_MIN && !inst->md_asked)
      inst->md_direct = !detect_running(inst) && vksift_hip_is_pinned(matches) == 1, inst->md_asked = true; /* once per matching: the query costs a microsecond; not behind a queued detection (vksift_buffers.c) */
    if (inst->match_slots_used >= VKSIFT_DL_BATCH_MIN && !inst->md_direct && (inst->md_valid || inst->md_hits++ > 0) &&
        packed_match_download(inst, pair, matches, n))
      return;
    HIP_CHECK(vksift_hip_memcpy_d2h(matches, inst->d_matches + (uint64_t)pair * inst->match_slot_stride, (size_t)n * MATCH_BYTES, inst->dl_stream),
              "match read-back");
    HIP_CHECK(vksift_hip_stream_sync(inst->dl_stream), "match read-back");
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
                                    inst->dl_stream),
              "filtered match read-back");
    HIP_CHECK(vksift_hip_stream_sync(inst->dl_stream), "filtered match read-back");
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

