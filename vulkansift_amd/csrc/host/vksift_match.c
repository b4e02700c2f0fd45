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

/* Device-side gather of `count` buffers (all with the layout of bufs[ids[0]]) into match slots first_slot.. .
 * The reference physically packs the octave sections (pack_BufferMemory, sift_memory.c:957-1047) after reading the
 * counts on the host; here the gather kernel reads the counters in HBM and walks the sections in the same order, so
 * nothing waits on the host. Returns the launch bound on the row count through *max_rows_out. */
int gather_buffers(vksift_Instance inst, const MatchScratch *ms, const uint32_t *ids, uint32_t count, uint32_t first_slot, bool side_b, uint8_t *d_desc_base,
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

MatchScratch fwd_scratch(vksift_Instance inst)
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
  const uint32_t bc = inst->batch_cap;
  inst->filtered_slot_stride = (((uint64_t)inst->cfg.max_nb_sift_per_buffer * 16u) + 255u) & ~(uint64_t)255u;
  /* each block only if it does not exist yet: a call that ran out of memory half-way is retried without leaking */
  bool ok = true;
#define ENSURE_D(ptr, bytes) ok = ok && ((ptr) != NULL || ((ptr) = vksift_hip_malloc(bytes)) != NULL)
  ENSURE_D(inst->rev.desc_a, inst->desc_slot_stride * bc);
  ENSURE_D(inst->rev.desc_b, inst->desc_slot_stride * bc);
  ENSURE_D(inst->rev.matches, inst->match_slot_stride * bc);
  ENSURE_D(inst->rev.norms, sizeof(uint32_t) * inst->norm_slot_stride * bc);
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
  range_open = true;
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

