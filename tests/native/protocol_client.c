/* The host-protocol legs of bench.py as a C caller runs them (bench.py's Python loops cost 4-8 us of interpreter time per frame —
 * a fifth of a 26 ms batch — which a C application of the reference does not pay). Public API only: the reference's per-frame
 * accessors (vulkansift.h) and the batched detect / match entries of vksift_ext.h. Built by bench.py with gcc into a small shared
 * object next to libvulkansift.so and called through ctypes; returns seconds for `steps` batches of `n` frames.
 *
 * proto_serial     src/perf/wrappers/vulkansift_wrapper.cpp:30-33 per frame = detect(host image) + getFeaturesNumber +
 *                  downloadFeatures; here per batch of n frames, then the self-match of every frame and the download of its
 *                  records. Strictly serial: nothing is queued while the host waits or copies.
 * proto_single     one image per call, everything downloaded (BASELINE config 2 read literally)
 * proto_pipelined  the same inputs and outputs with two sets of n SIFT buffers: the detection of the next batch is queued before the
 *                  results of the current one are fetched (vulkansift.h:43-47: detection and matching calls are asynchronous).
 * proto_plain      the 20 entry points of the reference ONLY (vulkansift.h; an instance from vksift_createInstance): one host image
 *                  per vksift_detectFeatures call, vksift_getFeaturesNumber + vksift_downloadFeatures per buffer — what an application
 *                  written against the reference does when it has more than one image at hand. */
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

#include <vulkansift/vulkansift.h>

#include "vksift_ext.h"

static double now_s(void)
{
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* PROTO_TRACE=1: where the host's time goes in the pipelined leg, per phase (stderr) */
#include <stdio.h>
static double g_t[8];
static void collect(vksift_Instance inst, uint32_t first, uint32_t n, int do_match, vksift_Feature *feat_buf, vksift_Match_2NN *match_buf)
{
  double t = now_s();
  for (uint32_t i = 0; i < n; i++)
  {
    const uint32_t nf = vksift_getFeaturesNumber(inst, first + i);
    if (i == 0)
      g_t[1] += now_s() - t, t = now_s(); /* the wait for the detection */
    if (nf)
      vksift_downloadFeatures(inst, feat_buf, first + i);
    if (i == 0)
      g_t[2] += now_s() - t, t = now_s(); /* the first download (plain copies to the caller's pageable buffer) */
    if (i == 1)
      g_t[3] += now_s() - t, t = now_s(); /* the second: starts the packed copy */
  }
  g_t[4] += now_s() - t, t = now_s();     /* the other downloads */
  if (do_match)
    for (uint32_t k = 0; k < n; k++)
    {
      const uint32_t nm = vksift_ext_getMatchesNumberBatch(inst, k);
      if (k == 0)
        g_t[5] += now_s() - t, t = now_s(); /* the wait for the matching */
      if (nm)
        vksift_ext_downloadMatchesBatch(inst, k, match_buf);
    }
  g_t[6] += now_s() - t;
}

double proto_serial(vksift_Instance inst, const uint8_t *const *images, uint32_t n, uint32_t w, uint32_t h, int do_match, uint32_t steps,
                    vksift_Feature *feat_buf, vksift_Match_2NN *match_buf)
{
  uint32_t *ids = (uint32_t *)malloc(sizeof(uint32_t) * n);
  for (uint32_t i = 0; i < n; i++)
    ids[i] = i;
  double t0 = 0.0;
  for (uint32_t s = 0; s <= steps; s++) /* step 0 is the warm-up */
  {
    if (s == 1)
      t0 = now_s();
    vksift_ext_detectFeaturesBatch(inst, images, n, w, h, 0u);
    for (uint32_t i = 0; i < n; i++)
      if (vksift_getFeaturesNumber(inst, i))
        vksift_downloadFeatures(inst, feat_buf, i);
    if (do_match)
    {
      vksift_ext_matchFeaturesBatch(inst, n, ids, ids);
      for (uint32_t k = 0; k < n; k++)
        if (vksift_ext_getMatchesNumberBatch(inst, k))
          vksift_ext_downloadMatchesBatch(inst, k, match_buf);
    }
  }
  const double dt = now_s() - t0;
  free(ids);
  return dt;
}

/* BASELINE config 2 literally (src/perf/perf_runtime.cpp:63-81 through vulkansift_wrapper.cpp:30-33): ONE host image per call,
 * vksift_detectFeatures + vksift_getFeaturesNumber + vksift_downloadFeatures (+ vksift_matchFeatures(0, 0) + vksift_getMatchesNumber +
 * vksift_downloadMatches), `warm` untimed runs, then the mean of `runs`. Returns seconds per run. */
double proto_single(vksift_Instance inst, const uint8_t *image, uint32_t w, uint32_t h, int do_match, uint32_t warm, uint32_t runs, vksift_Feature *feat_buf,
                    vksift_Match_2NN *match_buf, uint32_t *nb_feats)
{
  double t0 = 0.0, ph[4] = {0.0, 0.0, 0.0, 0.0};
  uint32_t n = 0;
  for (uint32_t i = 0; i < warm + runs; i++)
  {
    if (i == warm)
      t0 = now_s();
    const double ta = now_s();
    vksift_detectFeatures(inst, image, w, h, 0u);
    const double tb = now_s();
    n = vksift_getFeaturesNumber(inst, 0u);
    const double tc = now_s();
    vksift_downloadFeatures(inst, feat_buf, 0u);
    const double td = now_s();
    if (do_match)
    {
      vksift_matchFeatures(inst, 0u, 0u);
      if (vksift_getMatchesNumber(inst))
        vksift_downloadMatches(inst, match_buf);
    }
    if (i >= warm)
      ph[0] += tb - ta, ph[1] += tc - tb, ph[2] += td - tc, ph[3] += now_s() - td;
  }
  const double dt = (now_s() - t0) / (double)runs;
  if (getenv("PROTO_TRACE") && atoi(getenv("PROTO_TRACE")))
    fprintf(stderr, "PROTO_TRACE single image, us per run: detect call %.1f | wait (getFeaturesNumber) %.1f | downloadFeatures %.1f | match + download %.1f | period %.1f\n",
            1e6 * ph[0] / runs, 1e6 * ph[1] / runs, 1e6 * ph[2] / runs, 1e6 * ph[3] / runs, 1e6 * dt);
  if (nb_feats)
    *nb_feats = n;
  return dt;
}

static void run_pipelined(vksift_Instance inst, const uint8_t *const *images, uint32_t n, uint32_t w, uint32_t h, int do_match, uint32_t iters,
                          uint32_t *ids0, uint32_t *ids1, vksift_Feature *feat_buf, vksift_Match_2NN *match_buf)
{
  vksift_ext_detectFeaturesBatch(inst, images, n, w, h, 0u);
  if (do_match)
    vksift_ext_matchFeaturesBatch(inst, n, ids0, ids0);
  for (uint32_t it = 0; it < iters; it++)
  {
    const uint32_t cur = it & 1u, nxt = cur ^ 1u;
    const double t = now_s();
    if (it + 1 < iters)
      vksift_ext_detectFeaturesBatch(inst, images, n, w, h, nxt * n); /* queued behind the matching of `cur`; staged while the GPU works */
    g_t[0] += now_s() - t;
    collect(inst, cur * n, n, do_match, feat_buf, match_buf);
    if (it + 1 < iters && do_match)
      vksift_ext_matchFeaturesBatch(inst, n, nxt ? ids1 : ids0, nxt ? ids1 : ids0); /* the match slots are free again once `cur`'s records are out */
  }
}

/* inst must have been created with sift_buffer_count >= 2 n and a batch capacity of n */
double proto_pipelined(vksift_Instance inst, const uint8_t *const *images, uint32_t n, uint32_t w, uint32_t h, int do_match, uint32_t steps,
                       vksift_Feature *feat_buf, vksift_Match_2NN *match_buf)
{
  uint32_t *ids0 = (uint32_t *)malloc(sizeof(uint32_t) * n), *ids1 = (uint32_t *)malloc(sizeof(uint32_t) * n);
  for (uint32_t i = 0; i < n; i++)
    ids0[i] = i, ids1[i] = n + i;
  /* PROTO_PIN=1: page-locked result buffers (vksift_ext_pinHostMemory) — every download becomes a DMA transfer of its own out of device
   * memory. Measured on MI355X with the next detection already queued: 77 us per transfer (the copy engine's ring behind a busy GPU,
   * DESIGN.md section 4) against 7.5 us for the host copy out of the library's staged block: 6.9 k instead of 22.3 k frames/s. Off by
   * default; the entry is for callers that fetch while the GPU is idle. */
  const int pin = getenv("PROTO_PIN") && atoi(getenv("PROTO_PIN")) == 1;
  const vksift_Config cfg_sz = vksift_getDefaultConfig();
  const int pinned_f = pin && vksift_ext_pinHostMemory(feat_buf, (size_t)cfg_sz.max_nb_sift_per_buffer * sizeof(vksift_Feature)) == VKSIFT_SUCCESS;
  const int pinned_m = pin && vksift_ext_pinHostMemory(match_buf, (size_t)cfg_sz.max_nb_sift_per_buffer * sizeof(vksift_Match_2NN)) == VKSIFT_SUCCESS;
  run_pipelined(inst, images, n, w, h, do_match, 2, ids0, ids1, feat_buf, match_buf);
  const double t0 = now_s();
  run_pipelined(inst, images, n, w, h, do_match, steps, ids0, ids1, feat_buf, match_buf);
  const double dt = now_s() - t0;
  if (getenv("PROTO_TRACE") && atoi(getenv("PROTO_TRACE")))
    fprintf(stderr, "PROTO_TRACE ms per iteration (2 warm-up iterations included in the sums): detect call %.2f | wait detection %.2f | first download %.2f | "
                    "second %.2f | other %u downloads %.2f | wait matching %.2f | match downloads %.2f | measured period %.2f\n",
            g_t[0] * 1e3 / (steps + 2), g_t[1] * 1e3 / (steps + 2), g_t[2] * 1e3 / (steps + 2), g_t[3] * 1e3 / (steps + 2), n - 2, g_t[4] * 1e3 / (steps + 2),
            g_t[5] * 1e3 / (steps + 2), g_t[6] * 1e3 / (steps + 2), dt * 1e3 / steps);
  if (pinned_f)
    vksift_ext_unpinHostMemory(feat_buf);
  if (pinned_m)
    vksift_ext_unpinHostMemory(match_buf);
  free(ids0);
  free(ids1);
  return dt;
}

/* The reference's API and nothing else. `inst` comes from vksift_createInstance with sift_buffer_count >= n (mode 0), 2 n (mode 1) or 2
 * (mode 2); images[k % n_images] is frame k.
 *   mode 0  a run of n detect calls into buffers 0 .. n-1, then count + features of each (the detection cannot overlap the reads)
 *   mode 1  two sets of n buffers: the run of detect calls for the next set is issued BEFORE the current set is read, so the GPU works
 *           on one set while the host copies the other out (vulkansift.h:43-47)
 *   mode 2  the two-buffer ping-pong of a video loop: detect(frame k + 1) into the other buffer, then read frame k
 * do_match: every frame is also self-matched through vksift_matchFeatures + vksift_getMatchesNumber + vksift_downloadMatches (one
 * pair per call is all the reference's matching interface offers: it has one result slot).
 * Returns seconds for steps * n frames (mode 2: steps frames), after an untimed warm-up pass of min(steps, 10) iterations (16 in mode 2). */
static void plain_read(vksift_Instance inst, uint32_t first, uint32_t n, int do_match, vksift_Feature *feat_buf, vksift_Match_2NN *match_buf)
{
  for (uint32_t i = 0; i < n; i++)
  {
    if (vksift_getFeaturesNumber(inst, first + i))
      vksift_downloadFeatures(inst, feat_buf, first + i);
    if (do_match)
    {
      vksift_matchFeatures(inst, first + i, first + i);
      if (vksift_getMatchesNumber(inst))
        vksift_downloadMatches(inst, match_buf);
    }
  }
}

static void plain_run(vksift_Instance inst, const uint8_t *const *images, uint32_t n_images, uint32_t n, uint32_t w, uint32_t h, int mode, int do_match,
                      uint32_t iters, vksift_Feature *feat_buf, vksift_Match_2NN *match_buf)
{
  uint32_t k = 0;
  if (mode == 0)
  {
    for (uint32_t it = 0; it < iters; it++)
    {
      for (uint32_t i = 0; i < n; i++)
        vksift_detectFeatures(inst, images[k++ % n_images], w, h, i);
      plain_read(inst, 0u, n, do_match, feat_buf, match_buf);
    }
    return;
  }
  if (mode == 1)
  {
    for (uint32_t i = 0; i < n; i++)
      vksift_detectFeatures(inst, images[k++ % n_images], w, h, i);
    for (uint32_t it = 0; it < iters; it++)
    {
      const uint32_t cur = it & 1u, nxt = cur ^ 1u;
      if (it + 1 < iters)
        for (uint32_t i = 0; i < n; i++)
          vksift_detectFeatures(inst, images[k++ % n_images], w, h, nxt * n + i);
      plain_read(inst, cur * n, n, do_match, feat_buf, match_buf);
    }
    return;
  }
  vksift_detectFeatures(inst, images[k++ % n_images], w, h, 0u);
  for (uint32_t it = 0; it < iters; it++)
  {
    if (it + 1 < iters)
      vksift_detectFeatures(inst, images[k++ % n_images], w, h, (it + 1u) & 1u);
    plain_read(inst, it & 1u, 1u, do_match, feat_buf, match_buf);
  }
}

double proto_plain(vksift_Instance inst, const uint8_t *const *images, uint32_t n_images, uint32_t n, uint32_t w, uint32_t h, int mode, int do_match,
                   uint32_t steps, vksift_Feature *feat_buf, vksift_Match_2NN *match_buf)
{
  /* warm-up: long enough for the instance to have seen the calling pattern (its staging capacity doubles batch by batch) */
  uint32_t warm = mode == 2 ? 16u : 10u;
  plain_run(inst, images, n_images, n, w, h, mode, do_match, warm < steps ? warm : steps, feat_buf, match_buf);
  const double t0 = now_s();
  plain_run(inst, images, n_images, n, w, h, mode, do_match, steps, feat_buf, match_buf);
  return now_s() - t0;
}
