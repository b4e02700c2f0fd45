/*
 * vksift_buffers.c — SIFT-buffer accessors and scale-space inspection (sift_memory.c:1060-1383, vulkansift.c:346-415,464-518)
 */
#include "vksift_internal.h"

/* ------------------------------------------------------------------------------------------------ */
/* feature count / download / upload (sift_memory.c:1060-1272)                                      */
/* ------------------------------------------------------------------------------------------------ */
void wait_for_buffer(vksift_Instance inst, uint32_t buf)
{
  vksift_hip_set_device(inst->device);
  if (!counts_valid(inst, buf))
    wait_detect_seq(inst, inst->bufs[buf].seq); /* the detection that filled THIS buffer, not the latest one */
  if (inst->match_pending && inst->match_busy[buf])
  {
    vksift_hip_event_sync(inst->ev_match);
    inst->match_pending = false;
    memset(inst->match_busy, 0, sizeof(bool) * inst->cfg.sift_buffer_count);
  }
}

/* per-section stored counts, clamped to the section capacity (sift_memory.c:1080-1095) */
uint32_t buffer_counts(vksift_Instance inst, uint32_t buf, uint32_t *cnt, bool log_lost)
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
  defer_sync(instance);
  if (!buffer_idx_valid(instance, gpu_buffer_id))
  {
    logError(LOG_TAG, "vksift_getFeaturesNumber(): bad argument.");
    instance->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return 0;
  }
  wait_for_buffer(instance, gpu_buffer_id);
  return buffer_counts(instance, gpu_buffer_id, NULL, true);
}

/* Batched download (see vksift_internal.h: d_dl): true when feats_ptr was filled from the batch cache. Every failure falls
 * back to the per-buffer copies below. The caller has waited for the detection that filled the buffer. */
static bool download_from_batch(vksift_Instance inst, vksift_Feature *feats_ptr, uint32_t buf)
{
  const BufferInfo *b = &inst->bufs[buf];
  const DetectSlot *det = &inst->det_ring[b->seq % VKSIFT_DETECT_RING];
  if (b->seq == 0 || det->seq != b->seq)
    return false; /* filled from the host, or by a detection whose record has left the ring */
  const uint32_t first = det->first, count = det->count;
  if (count < VKSIFT_DL_BATCH_MIN || buf < first || buf >= first + count || b->nb_sections == 0 || b->is_packed)
    return false;
  /* The packed copy moves EVERY buffer of the detection: a caller that samples one frame of 128 must not pay for the other 127
   * (nor for the staging memory). The first download after a detection therefore takes the per-section copies; a second one
   * says the caller walks the batch, and the rest of it comes out of one packed copy. */
  const bool cached = inst->dl_valid && inst->dl_seq == b->seq;
  if (inst->dl_hits_seq != b->seq)
    inst->dl_hits_seq = b->seq, inst->dl_hits = 0;
  if (!cached && inst->dl_hits++ == 0)
    return false;
  if (!cached)
  {
    /* the rebuild below overwrites dl_row[] and may reallocate the staging pair before any of its early exits: whatever was
     * cached (for another detection) is gone from here on, and the cache is valid again only once every copy and event is queued */
    inst->dl_valid = false;
    /* every buffer of the batch shares the section table of `b` (one resolution per batched detection) */
    if (!inst->dl_row)
      inst->dl_row = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)inst->cfg.sift_buffer_count + 1u));
    if (!inst->dl_row)
      return false;
    uint32_t rows = 0, max_rows = 0;
    for (uint32_t i = 0; i < count; i++)
    {
      const BufferInfo *bi = &inst->bufs[first + i];
      if (bi->seq != b->seq)
        return false; /* a buffer of the range was refilled by something else since */
      const uint32_t n = buffer_counts(inst, first + i, NULL, false);
      inst->dl_row[i] = rows;
      if (n > 0x7FFFFFFFu - rows)
        return false; /* more records than the 32-bit row offsets of the packed copy address: per-buffer copies */
      rows += n;
      max_rows = n > max_rows ? n : max_rows;
    }
    inst->dl_row[count] = rows;
    const size_t bytes = (size_t)rows * FEAT_BYTES;
    /* the staging pair follows the workload down as well as up: a block more than four times what this detection needs (and
     * beyond 64 MB) is released instead of being kept for the life of the instance */
    if (bytes > inst->dl_cap || (inst->dl_cap > ((size_t)64 << 20) && inst->dl_cap / 4u > bytes + 4096u))
    {
      vksift_hip_free(inst->d_dl);
      vksift_hip_host_free(inst->h_dl);
      const size_t cap = bytes + bytes / 4u + 4096u;
      inst->d_dl = (uint8_t *)vksift_hip_malloc(cap);
      inst->h_dl = (uint8_t *)vksift_hip_host_malloc(cap);
      inst->dl_cap = (inst->d_dl && inst->h_dl) ? cap : 0;
      if (!inst->dl_cap)
      {
        vksift_hip_free(inst->d_dl);
        vksift_hip_host_free(inst->h_dl);
        inst->d_dl = inst->h_dl = NULL;
        return false;
      }
    }
    uint32_t ids[64];
    for (uint32_t i0 = 0; i0 < count; i0 += 64)
    {
      const uint32_t n = count - i0 < 64u ? count - i0 : 64u;
      for (uint32_t k = 0; k < n; k++)
        ids[k] = first + i0 + k;
      if (vksift_hip_pack_features(inst->d_feats, inst->buf_stride, ids, inst->dl_row + i0, n, b->nb_sections, b->sec_off, b->sec_cap, inst->d_found,
                                   VKSIFT_MAX_OCTAVES, inst->d_dl, max_rows, NULL, inst->dl_stream) != 0)
        return false;
    }
    /* A page-locked destination (vksift_ext_pinHostMemory, hipHostMalloc) takes its records by DMA straight out of the packed device
     * copy: no transfer into the pinned staging block for this detection at all. (A later pageable destination of the same detection
     * falls back to a plain copy out of the device block.) */
    /* Only on an otherwise idle GPU: behind a queued detection each such transfer takes ~77 us against the staged path's ~7.5 us per
     * buffer, so a pipelined caller with pinned destinations keeps the staged path. */
    inst->dl_direct = !detect_running(inst) && vksift_hip_is_pinned(feats_ptr) == 1;
    if (inst->dl_direct)
    {
      if (!inst->dl_ev[0])
        inst->dl_ev[0] = vksift_hip_event_create();
      if (!inst->dl_ev[0] || vksift_hip_event_record(inst->dl_ev[0], inst->dl_stream) != 0)
        return false;
      inst->dl_chunks = 0, inst->dl_chunks_done = 0;
      inst->dl_first = first, inst->dl_count = count, inst->dl_seq = b->seq, inst->dl_valid = true;
    }
    /* the copy goes in a few pieces with an event each: a caller that walks the buffers in order copies buffer i out of pinned
     * memory while the pieces behind it are still on the bus */
    uint32_t nch = inst->dl_direct ? 0u : (uint32_t)(bytes / ((size_t)4 << 20)) + 1u;
    if (nch > VKSIFT_DL_CHUNKS)
      nch = VKSIFT_DL_CHUNKS;
    for (uint32_t k = 0; k < nch; k++)
    {
      const size_t lo = bytes * k / nch, hi = bytes * (k + 1) / nch;
      if (hi > lo && vksift_hip_memcpy_d2h(inst->h_dl + lo, inst->d_dl + lo, hi - lo, inst->dl_stream) != 0)
        return false;
      if (!inst->dl_ev[k])
        inst->dl_ev[k] = vksift_hip_event_create();
      if (!inst->dl_ev[k] || vksift_hip_event_record(inst->dl_ev[k], inst->dl_stream) != 0)
        return false;
      inst->dl_chunk_end[k] = hi;
    }
    if (!inst->dl_direct)
    {
      inst->dl_chunks = nch, inst->dl_chunks_done = 0;
      inst->dl_first = first, inst->dl_count = count, inst->dl_seq = b->seq, inst->dl_valid = true;
    }
  }
  const uint32_t i = buf - first;
  if (inst->dl_direct)
  {
    /* device block -> the caller's memory: one transfer on the download stream (behind the pack kernels), one wait */
    const size_t n_bytes = (size_t)(inst->dl_row[i + 1] - inst->dl_row[i]) * FEAT_BYTES;
    if (n_bytes == 0)
      return true;
    if (vksift_hip_memcpy_d2h(feats_ptr, inst->d_dl + (size_t)inst->dl_row[i] * FEAT_BYTES, n_bytes, inst->dl_stream) != 0 ||
        vksift_hip_stream_sync(inst->dl_stream) != 0)
    {
      inst->dl_valid = false;
      return false;
    }
    return true;
  }
  {
    /* wait for the pieces that hold this buffer's records */
    const size_t need = (size_t)inst->dl_row[i + 1] * FEAT_BYTES;
    while (inst->dl_chunks_done < inst->dl_chunks && (inst->dl_chunks_done == 0 || inst->dl_chunk_end[inst->dl_chunks_done - 1] < need))
    {
      if (vksift_hip_event_sync(inst->dl_ev[inst->dl_chunks_done]) != 0)
      {
        inst->dl_valid = false;
        return false;
      }
      inst->dl_chunks_done++;
    }
  }
  memcpy(feats_ptr, inst->h_dl + (size_t)inst->dl_row[i] * FEAT_BYTES, (size_t)(inst->dl_row[i + 1] - inst->dl_row[i]) * FEAT_BYTES);
  return true;
}

/* Posted features (vksift_internal.h: h_post): the detection itself left the dense records of this buffer in pinned memory */
static bool download_posted(vksift_Instance inst, vksift_Feature *feats_ptr, uint32_t buf)
{
  const uint32_t slot = buf & 1u;
  const BufferInfo *b = &inst->bufs[buf];
  if (inst->post_seq[slot] == 0 || inst->post_seq[slot] != b->seq || inst->post_buf[slot] != buf)
    return false;
  const uint32_t n = buffer_counts(inst, buf, NULL, false);
  memcpy(feats_ptr, inst->h_post[slot], (size_t)n * FEAT_BYTES);
  inst->post_fetched[slot] = true;
  inst->post_idle = 0;
  return true;
}

/* One sectioned buffer (a single detection): the sections are packed on the device into the pinned staging pair and fetched with ONE
 * copy, then copied out — instead of one device-to-pageable-host copy per octave section, each of which the runtime stages and
 * synchronises on its own (five copies of a 640x480 detection: ~90 us of the 0.5 ms single-image latency; this path: ~45 us).
 * false: nothing was done (a failure here falls back to the per-section copies). */
static bool download_one_packed(vksift_Instance inst, vksift_Feature *feats_ptr, uint32_t buf)
{
  const BufferInfo *b = &inst->bufs[buf];
  if (b->nb_sections < 2 || b->is_packed)
    return false;
  const uint32_t n = buffer_counts(inst, buf, NULL, false);
  if (n == 0)
    return true;
  const size_t bytes = (size_t)n * FEAT_BYTES;
  if (bytes > ((size_t)64 << 20))
    return false; /* large single detections: the section copies stream at the bus rate anyway */
  if (bytes > inst->dl_cap)
  {
    inst->dl_valid = false; /* the batch cache lives in the same staging pair */
    vksift_hip_free(inst->d_dl);
    vksift_hip_host_free(inst->h_dl);
    const size_t cap = bytes + bytes / 4u + 4096u;
    inst->d_dl = (uint8_t *)vksift_hip_malloc(cap);
    inst->h_dl = (uint8_t *)vksift_hip_host_malloc(cap);
    inst->dl_cap = (inst->d_dl && inst->h_dl) ? cap : 0;
    if (!inst->dl_cap)
    {
      vksift_hip_free(inst->d_dl);
      vksift_hip_host_free(inst->h_dl);
      inst->d_dl = inst->h_dl = NULL;
      return false;
    }
  }
  inst->dl_valid = false;
  const uint32_t zero = 0;
  if (vksift_hip_pack_features(inst->d_feats, inst->buf_stride, &buf, &zero, 1, b->nb_sections, b->sec_off, b->sec_cap, inst->d_found, VKSIFT_MAX_OCTAVES,
                               inst->d_dl, n, NULL, inst->dl_stream) != 0)
    return false;
  if (vksift_hip_memcpy_d2h(inst->h_dl, inst->d_dl, bytes, inst->dl_stream) != 0 || vksift_hip_stream_sync(inst->dl_stream) != 0)
    return false;
  memcpy(feats_ptr, inst->h_dl, bytes);
  {
    /* the caller does fetch single detections: post the next ones. (Not when this buffer came out of a batched detection — the first
     * download of a batch lands here too, and says nothing about single-image use.) */
    const DetectSlot *det = &inst->det_ring[b->seq % VKSIFT_DETECT_RING];
    const bool single = b->seq != 0 && det->seq == b->seq && det->count == 1u;
    if (single && inst->post_enabled && !inst->post_on)
      inst->post_on = true, inst->post_idle = 0;
  }
  return true;
}

void vksift_downloadFeatures(vksift_Instance instance, vksift_Feature *feats_ptr, uint32_t gpu_buffer_id)
{
  defer_sync(instance);
  if (!buffer_idx_valid(instance, gpu_buffer_id))
  {
    logError(LOG_TAG, "vksift_downloadFeatures(): bad argument.");
    instance->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return;
  }
  vksift_Instance inst = instance;
  wait_for_buffer(inst, gpu_buffer_id);
  if (download_posted(inst, feats_ptr, gpu_buffer_id) || download_from_batch(inst, feats_ptr, gpu_buffer_id))
    return;
  const BufferInfo *b = &inst->bufs[gpu_buffer_id];
  const uint8_t *base = inst->d_feats + (uint64_t)gpu_buffer_id * inst->buf_stride;
  if (download_one_packed(inst, feats_ptr, gpu_buffer_id))
    return;
  if (b->nb_sections == 0)
  {
    HIP_CHECK(vksift_hip_memcpy_d2h(feats_ptr, base, (size_t)b->nb_stored * FEAT_BYTES, inst->dl_stream), "feature download");
  }
  else
  {
    uint32_t cnt[VKSIFT_MAX_OCTAVES] = {0};
    buffer_counts(inst, gpu_buffer_id, cnt, false);
    uint32_t out = 0;
    for (uint32_t o = 0; o < b->nb_sections; o++)
    {
      HIP_CHECK(vksift_hip_memcpy_d2h((uint8_t *)feats_ptr + (size_t)out * FEAT_BYTES, base + (size_t)b->sec_off[o] * FEAT_BYTES, (size_t)cnt[o] * FEAT_BYTES,
                                      inst->dl_stream),
                "feature download");
      out += cnt[o];
    }
  }
  HIP_CHECK(vksift_hip_stream_sync(inst->dl_stream), "feature download");
  return;
gpu_error:
  logError(LOG_TAG, "vksift_downloadFeatures(): the device-to-host copy of the features failed.");
  instance->error_cb(VKSIFT_VULKAN_ERROR);
}

void vksift_uploadFeatures(vksift_Instance instance, const vksift_Feature *feats_ptr, const uint32_t nb_feats, const uint32_t gpu_buffer_id)
{
  defer_sync(instance);
  if (!buffer_idx_valid(instance, gpu_buffer_id) || nb_feats > instance->cfg.max_nb_sift_per_buffer)
  {
    if (nb_feats > instance->cfg.max_nb_sift_per_buffer)
      logError(LOG_TAG, "%d features do not fit a SIFT buffer of %d (max_nb_sift_per_buffer).", nb_feats,
               instance->cfg.max_nb_sift_per_buffer);
    logError(LOG_TAG, "vksift_uploadFeatures(): bad argument.");
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
  inst->cache_valid[gpu_buffer_id] = false;
  b->nb_sections = 0;
  b->seq = 0; /* host-filled: nothing to wait for, and no longer part of a detection's packed download */
  return;
gpu_error:
  logError(LOG_TAG, "vksift_uploadFeatures(): the host-to-device copy of the features failed.");
  instance->error_cb(VKSIFT_VULKAN_ERROR);
}

/* ------------------------------------------------------------------------------------------------ */
/* scale-space inspection (vulkansift.c:464-518, sift_memory.c:1303-1383)                           */
/* ------------------------------------------------------------------------------------------------ */
uint8_t vksift_getScaleSpaceNbOctaves(vksift_Instance instance)
{
  defer_sync(instance);
  return (uint8_t)instance->lay.n_oct;
}

void vksift_getScaleSpaceOctaveResolution(vksift_Instance instance, const uint8_t octave, uint32_t *octave_images_width, uint32_t *octave_images_height)
{
  defer_sync(instance);
  if (octave >= instance->lay.n_oct)
  {
    logError(LOG_TAG, "vksift_getScaleSpaceOctaveResolution(): octave %d requested, the current scale-space has %d",
             octave, instance->lay.n_oct);
    instance->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return;
  }
  *octave_images_width = instance->lay.w[octave];
  *octave_images_height = instance->lay.h[octave];
}

static void download_plane(vksift_Instance inst, uint8_t octave, uint8_t scale, bool is_dog, float *dst, const char *fn)
{
  float *tmp = NULL;
  defer_sync(inst);
  uint32_t nscales = inst->S + (is_dog ? 2 : 3);
  if (octave >= inst->lay.n_oct || scale >= nscales)
  {
    if (octave >= inst->lay.n_oct)
      logError(LOG_TAG, "octave %d requested, the current scale-space has %d", octave, inst->lay.n_oct);
    else
      logError(LOG_TAG, "scale %d requested, an octave has %d %s layers", scale, nscales, is_dog ? "difference-of-Gaussian" : "Gaussian");
    logError(LOG_TAG, "%s error: invalid input.", fn);
    inst->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return;
  }
  vksift_hip_set_device(inst->device);
  /* images cannot be read while a detection runs (vulkansift.c:490-491) */
  HIP_CHECK(wait_all(inst), "stream synchronisation");
  const PyrLayout *L = &inst->lay;
  if (inst->d_pyr == NULL || inst->cur_w == 0)
  {
    logError(LOG_TAG, "%s error: no scale-space available (no detection since the last resize).", fn);
    inst->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return;
  }
  const float *src = pyr_at(inst, L->gauss_off[octave] + (uint64_t)scale * L->plane_stride[octave]);
  if (is_dog || inst->fp16)
  {
    /* DifferenceOfGaussian.comp layer z = G[z+1] - G[z]: the detection path forms these in registers and never stores them;
     * this (debug) accessor materialises the requested layer of image 0. A binary16 pyramid is widened to fp32 the same way
     * (the reference blits R16 -> R32 for the download, sift_memory.c:1313-1325). */
    tmp = (float *)vksift_hip_malloc(sizeof(float) * (size_t)L->w[octave] * L->h[octave]);
    if (!tmp)
      goto gpu_error;
    HIP_CHECK(vksift_hip_dog_plane(src, is_dog ? pyr_at(inst, L->gauss_off[octave] + (uint64_t)(scale + 1) * L->plane_stride[octave]) : NULL, L->w[octave],
                                   L->h[octave], L->pitch[octave], inst->fp16 ? 1u : 0u, tmp, inst->stream),
              "layer conversion");
    HIP_CHECK(vksift_hip_memcpy_d2h(dst, tmp, sizeof(float) * (size_t)L->w[octave] * L->h[octave], inst->stream), "plane download");
  }
  else
    HIP_CHECK(vksift_hip_memcpy2d_d2h(dst, sizeof(float) * L->w[octave], src, sizeof(float) * L->pitch[octave], sizeof(float) * L->w[octave], L->h[octave],
                                      inst->stream),
              "plane download");
  HIP_CHECK(vksift_hip_stream_sync(inst->stream), "plane download");
  vksift_hip_free(tmp);
  return;
gpu_error:
  vksift_hip_free(tmp);
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
  logWarning(LOG_TAG, "vksift_presentDebugFrame(): this build has no frame presenter, the call does nothing.");
}

