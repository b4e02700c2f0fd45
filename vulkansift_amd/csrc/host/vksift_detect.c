/*
 * vksift_detect.c — the detection pipeline (vulkansift.c:315-344 + sift_memory.c:891-955 + sift_detector.c:1313-1410,1462-1542)
 */
#include "vksift_internal.h"
#include <pthread.h>

/* Host copy of the caller's images into the pinned staging buffer (the caller may reuse its memory as soon as the call returns,
 * sift_memory.c:943). A batch is tens of megabytes: one thread moves ~10 GB/s, which made this copy as long as the detection
 * itself (39 MB per 128 VGA frames: 4 ms). A small pool of persistent workers shares it (creating threads per call cost as
 * much as a chunk's copy), and the batch goes chunk by chunk: the host-to-device copy of chunk i runs while chunk i+1 is staged. */
enum { STAGE_MAXT = 8 };
typedef struct
{
  uint8_t *dst;
  const uint8_t *const *images;
  uint32_t i0, i1;
  size_t img_bytes;
} StageJob;

static struct
{
  pthread_mutex_t user;     /* one staging operation at a time (instances on different threads share the pool) */
  pthread_mutex_t mu;
  pthread_cond_t cv_work, cv_done;
  pthread_t th[STAGE_MAXT];
  StageJob job[STAGE_MAXT];
  uint64_t gen[STAGE_MAXT]; /* generation each worker has to run (0: none yet) */
  uint64_t cur;
  uint32_t pending, nworkers;
  bool started;
} g_stage = {.user = PTHREAD_MUTEX_INITIALIZER, .mu = PTHREAD_MUTEX_INITIALIZER, .cv_work = PTHREAD_COND_INITIALIZER, .cv_done = PTHREAD_COND_INITIALIZER};

static void stage_copy(const StageJob *j)
{
  for (uint32_t i = j->i0; i < j->i1; i++)
    memcpy(j->dst + (size_t)i * j->img_bytes, j->images[i], j->img_bytes);
}

static void *stage_worker(void *p)
{
  const uint32_t id = (uint32_t)(uintptr_t)p;
  uint64_t seen = 0;
  pthread_mutex_lock(&g_stage.mu);
  for (;;)
  {
    while (g_stage.gen[id] == seen)
      pthread_cond_wait(&g_stage.cv_work, &g_stage.mu);
    seen = g_stage.gen[id];
    const StageJob j = g_stage.job[id];
    pthread_mutex_unlock(&g_stage.mu);
    stage_copy(&j);
    pthread_mutex_lock(&g_stage.mu);
    if (--g_stage.pending == 0)
      pthread_cond_signal(&g_stage.cv_done);
  }
  return NULL;
}

/* images [i0, i1) -> dst, shared by the caller and up to STAGE_MAXT - 1 workers; returns when all of it is in place */
static void stage_images(uint8_t *dst, const uint8_t *const *images, uint32_t i0, uint32_t i1, size_t img_bytes)
{
  const uint32_t n = i1 - i0;
  StageJob all = {dst, images, i0, i1, img_bytes};
  if ((size_t)n * img_bytes < ((size_t)2 << 20) || n < 2)
  {
    stage_copy(&all);
    return;
  }
  pthread_mutex_lock(&g_stage.user);
  if (!g_stage.started)
  {
    g_stage.started = true;
    for (uint32_t t = 0; t + 1 < STAGE_MAXT; t++)
      if (pthread_create(&g_stage.th[g_stage.nworkers], NULL, stage_worker, (void *)(uintptr_t)g_stage.nworkers) == 0)
      {
        pthread_detach(g_stage.th[g_stage.nworkers]);
        g_stage.nworkers++;
      }
  }
  uint32_t parts = g_stage.nworkers + 1u;
  if (parts > n)
    parts = n;
  pthread_mutex_lock(&g_stage.mu);
  g_stage.cur++;
  g_stage.pending = parts - 1u;
  for (uint32_t t = 1; t < parts; t++)
  {
    StageJob *j = &g_stage.job[t - 1];
    *j = all;
    j->i0 = i0 + (uint32_t)((uint64_t)n * t / parts), j->i1 = i0 + (uint32_t)((uint64_t)n * (t + 1) / parts);
    g_stage.gen[t - 1] = g_stage.cur;
  }
  pthread_cond_broadcast(&g_stage.cv_work);
  pthread_mutex_unlock(&g_stage.mu);
  all.i1 = i0 + (uint32_t)((uint64_t)n / parts);
  stage_copy(&all); /* the caller takes the first share */
  pthread_mutex_lock(&g_stage.mu);
  while (g_stage.pending != 0)
    pthread_cond_wait(&g_stage.cv_done, &g_stage.mu);
  pthread_mutex_unlock(&g_stage.mu);
  pthread_mutex_unlock(&g_stage.user);
}


/* ------------------------------------------------------------------------------------------------ */
/* detection (vulkansift.c:315-344 + sift_memory.c:891-955 + sift_detector.c:1313-1410,1462-1542)   */
/* ------------------------------------------------------------------------------------------------ */
vksift_hip_Plane plane_at(vksift_Instance inst, uint32_t o, uint64_t base_off, uint32_t layer)
{
  vksift_hip_Plane p;
  p.base = pyr_at(inst, base_off + (uint64_t)layer * inst->lay.plane_stride[o]);
  p.fp16 = inst->fp16 ? 1u : 0u;
  p.reverse = 0;
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
    bytes += (uint64_t)L->w[o] * L->h[o] * pyr_texel_bytes(inst) * ((inst->S + 3) + (inst->S + 2) + (inst->S + 2));
  bytes += (uint64_t)w * h + (uint64_t)L->w[0] * L->h[0] * 2u * pyr_texel_bytes(inst);
  return bytes;
}

/* fold the (completed) event timings of a detect call into the running sums */
void account_set(vksift_Instance inst, ProfSet *ps)
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
  inst->acc_ms[6] += vksift_hip_event_elapsed_ms(e[2], ps->ev_scan);
  inst->acc_ms[7] += vksift_hip_event_elapsed_ms(ps->ev_pt[0], ps->ev_pt[2]);
  inst->acc_blur_launches_all += ps->blur_launches_all;
  inst->acc_calls++;
  inst->acc_blur_launches += ps->blur_launches;
  inst->acc_alg_bytes += ps->alg_bytes;
  inst->acc_scan_bytes += ps->scan_bytes;
  ps->accounted = true;
}

/* all detections have completed (caller waited): account both event sets, oldest first */
void account_timings(vksift_Instance inst)
{
  account_set(inst, &inst->prof[inst->prof_cur ^ 1]);
  account_set(inst, &inst->prof[inst->prof_cur]);
}

/* ------------------------------------------------------------------------------------------------ */
/* the launch sequence of one detection                                                             */
/* ------------------------------------------------------------------------------------------------ */
#define TRY(expr, what)                                                        \
  do                                                                           \
  {                                                                            \
    int _e = (expr);                                                           \
    if (_e != 0)                                                               \
    {                                                                          \
      logError(LOG_TAG, "%s failed: %s", what, vksift_hip_error_string(_e));   \
      return _e;                                                               \
    }                                                                          \
  } while (0)

typedef struct
{
  vksift_Instance inst;
  const PyrLayout *L;
  ProfSet *PS;
  bool prof;      /* HIP-event stage timings requested */
  bool overlap;   /* scale-space on its own stream and buffer (ping-pong), see detect_impl */
  bool upload;    /* host images were staged in h_input and have to be copied to d_input */
  bool prestaged; /* ... they are in h_input already (deferred vksift_detectFeatures calls): nothing to stage */
  bool capturing; /* the sequence is being captured into a hipGraph: no host-visible events inside */
  bool gpu_busy;  /* an earlier detection was still running when this one was queued */
  bool post;      /* feature posting at the end of the sequence (vksift_internal.h: h_post) */
  bool fork;      /* scales S+1.. of every octave on the side stream (vksift_internal.h: ev_fork) */
  bool dense;     /* the descriptor launch also writes the buffers' matcher cache entries (vksift_hip_DenseRows) */
  const uint8_t *const *images; /* upload: the caller's images, staged chunk by chunk while the sequence is enqueued */
  const uint8_t *d_src;
  uint32_t w, h, count, first_buf;
  size_t img_bytes;
  uint32_t nblur; /* blur launches of octave 0 (the profiled scale-space interval) */
  uint32_t nblur_all; /* ... of every octave */
  bool zero_copy;     /* one host image: the seed launch reads it straight out of the pinned staging buffer (no copy to d_input in front of it) */
  bool tail_batch;    /* a batch queues scales S+1, S+2 of its coarser octaves per SCALE (enqueue_tail), like a forked detection does */
  bool tail[VKSIFT_MAX_OCTAVES]; /* octave o was queued up to scale S only: its last two scales go with enqueue_tail */
  vksift_hip_OctaveJob jobs[VKSIFT_MAX_OCTAVES];
} DetectCtx;

static void build_jobs(DetectCtx *c)
{
  vksift_Instance inst = c->inst;
  const PyrLayout *L = c->L;
  const BufferInfo *b0 = &inst->bufs[c->first_buf];
  for (uint32_t o = 0; o < L->n_oct; o++)
  {
    vksift_hip_OctaveJob *j = &c->jobs[o];
    memset(j, 0, sizeof(*j));
    j->gauss = pyr_at(inst, L->gauss_off[o]);
    j->fp16 = inst->fp16 ? 1u : 0u;
    j->w = L->w[o], j->h = L->h[o], j->pitch = L->pitch[o];
    j->plane_stride = L->plane_stride[o];
    j->img_stride = inst->pyr_img_stride;
    j->S = inst->S;
    j->octave_idx = (int32_t)o - (inst->cfg.use_input_upsampling ? 1 : 0);
    j->seed_sigma = inst->cfg.seed_scale_sigma;
    j->dog_threshold = inst->cfg.intensity_threshold / (float)inst->S;
    j->edge_limit = ((inst->cfg.edge_threshold + 1.f) * (inst->cfg.edge_threshold + 1.f)) / inst->cfg.edge_threshold;
    j->feats = inst->d_feats + (uint64_t)c->first_buf * inst->buf_stride + (uint64_t)b0->sec_off[o] * FEAT_BYTES;
    j->feat_img_stride = inst->buf_stride;
    j->cap = b0->sec_cap[o];
    j->found = inst->d_found + (size_t)c->first_buf * VKSIFT_MAX_OCTAVES + o;
    j->found_img_stride = VKSIFT_MAX_OCTAVES;
    /* segment scratch is octave-major: [octave][image][segment], so one octave's masks of the whole batch are contiguous */
    const uint64_t nsegs_o = (uint64_t)inst->S * L->h[o] * ((L->w[o] + 63) / 64);
    j->seg_mask = inst->d_seg_mask + L->seg_off[o] * c->count;
    j->seg_off = inst->d_seg_off + L->seg_off[o] * c->count;
    j->seg_img_stride = nsegs_o;
    j->cand_xy = inst->d_cand_xy + L->cand_off[o];
    j->cand_flag = inst->d_cand_flag + L->cand_off[o];
    j->cand_n = inst->d_cand_n + (size_t)o * inst->det_cap;
    j->cand_img_stride = inst->cand_cap;
    j->cand_cap = (uint32_t)L->cand_cap[o];
    j->ori_ang = inst->d_ori_ang + (size_t)b0->sec_off[o] * VKSIFT_HIP_MAX_ORI;
    j->ori_cnt = inst->d_ori_cnt + b0->sec_off[o];
    j->ori_img_stride = inst->ori_cap;
    j->max_ori = inst->cfg.max_nb_orientation_per_keypoint;
    j->use_vlfeat = inst->cfg.descriptor_format == VKSIFT_DESCRIPTOR_FORMAT_VLFEAT ? 1u : 0u;
    j->desc_fp_tab = inst->d_desc_fp;
    j->desc_fp_tab_len = inst->desc_fp_len;
    j->masks_cleared = 0;
    j->sec_index = o;
    j->scan_reverse = 0; /* set by enqueue_pyramid: the direction opposite to the octave's last blur launch */
  }
}

/* Scale-space construction of octave o on stream sp (sift_detector.c:881-1037). The DoG pass of the reference
 * (sift_detector.c:1039-1079) has no counterpart: the extrema stage forms D[s] = G[s+1] - G[s] from the Gaussian planes.
 * *g0_done: plane 0 of this octave was already written by the previous octave's scale-S pass; on return it tells the same
 * for the next octave. */
/* images [first, first + count) of the batch only: the same launches on a slice of the planes */
static vksift_hip_Plane plane_sub(vksift_Instance inst, vksift_hip_Plane p, uint32_t first)
{
  p.base = (float *)((uint8_t *)p.base + (uint64_t)first * p.img_stride * pyr_texel_bytes(inst));
  return p;
}

enum { PYR_FIRST_GROUP = 1, PYR_LAST_GROUP = 2, PYR_TRUNK = 4 };
static int enqueue_pyramid(DetectCtx *c, uint32_t o, vksift_hip_stream sp, uint32_t first, uint32_t count, int group_flags, bool *g0_done)
{
  vksift_Instance inst = c->inst;
  const PyrLayout *L = c->L;
  uint32_t nb_o = 0;
#define PL(oct, layer) plane_sub(inst, plane_at(inst, (oct), L->gauss_off[(oct)], (layer)), first)
  vksift_hip_range_push("Scale space construction");
  if (o == 0)
  {
    if (c->prof && (group_flags & PYR_FIRST_GROUP))
      vksift_hip_event_record(c->PS->ev_pt[0], sp);
    /* u8 -> fp32 (2x LINEAR blit when up-sampling) + seed blur: one fused pass when the shape allows it, else the blit goes
     * into the (still unused) layer-1 slot and is seed-blurred into layer 0 */
    const uint8_t *src = c->d_src + (size_t)first * c->img_bytes;
    int fused = -1;
    if (L->w[0] == 2 * c->w && L->h[0] == 2 * c->h)
    {
      fused = vksift_hip_seed_upsampled(src, c->w, c->h, c->img_bytes, PL(0, 0), &inst->taps[0], inst->ntaps[0], count, sp);
      if (fused > 0)
        TRY(fused, "fused up-sampling + seed blur");
    }
    else if (L->w[0] == c->w && L->h[0] == c->h)
    {
      fused = vksift_hip_seed_direct(src, c->w, c->h, c->img_bytes, PL(0, 0), &inst->taps[0], inst->ntaps[0], count, sp);
      if (fused > 0)
        TRY(fused, "fused input conversion + seed blur");
    }
    if (fused < 0)
    {
      vksift_hip_Plane tmp = PL(0, 1);
      TRY(vksift_hip_input_blit(src, c->w, c->h, c->img_bytes, tmp, count, sp), "input blit");
      TRY(vksift_hip_blur(tmp, PL(0, 0), &inst->taps[0], inst->ntaps[0], count, sp), "seed blur");
    }
    /* d_input has been consumed: the next upload (possibly on another stream) may overwrite it after this point */
    if (!c->capturing && (group_flags & PYR_LAST_GROUP))
    {
      TRY(vksift_hip_event_record(inst->ev_input_free, sp), "event record");
      inst->input_free_valid = true;
    }
    nb_o++;
  }
  else if (!*g0_done)
    TRY(vksift_hip_downsample(PL(o - 1, inst->S), PL(o, 0), count, sp), "downsample");
  *g0_done = false;
  /* consecutive launches of the chain walk the batch in opposite directions: a launch starts on the planes its predecessor wrote
   * last, which are still in the Infinity Cache (a whole-batch plane is 2.5x the cache: in the same direction every read misses);
   * the extrema scan continues the alternation */
  uint32_t li = 0;
  for (uint32_t s = 1; s < inst->S + 3; s++)
  {
    /* PYR_TRUNK (forked detections, the coarser octaves of a batch): the launches up to scale S — the path to the next octave — only;
     * scales S+1 and S+2 are queued per SCALE afterwards (enqueue_tail) */
    if ((group_flags & PYR_TRUNK) && s == inst->S + 1u)
      break;
    const vksift_hip_Plane srcp = PL(o, s - 1);
    vksift_hip_Plane dstp = PL(o, s);
    li++;
    dstp.reverse = inst->alt_order ? (li & 1u) : 0u;
    /* two scales in one launch where the kernels cover the tap counts (the source plane is read once, scale s never re-read):
     * not across scale S, which also seeds the next octave */
    if (s + 1 < inst->S + 3 && s != inst->S && s + 1 != inst->S)
    {
      vksift_hip_Plane dst2 = PL(o, s + 1);
      dst2.reverse = dstp.reverse;
      const int pe = vksift_hip_blur_pair(srcp, dstp, dst2, &inst->taps[s * VKSIFT_MAX_TAPS], inst->ntaps[s], &inst->taps[(s + 1) * VKSIFT_MAX_TAPS],
                                          inst->ntaps[s + 1], count, sp);
      if (pe > 0)
        TRY(pe, "two-scale blur");
      if (pe == 0)
      {
        s++;
        nb_o++;
        continue;
      }
    }
    int fused_ds = -1;
    if (s == inst->S && o + 1 < L->n_oct)
    {
      /* scale S also seeds the next octave (sift_detector.c:1003-1034): stored by the same pass when the sizes halve exactly */
      fused_ds = vksift_hip_blur_downsample(srcp, dstp, PL(o + 1, 0), &inst->taps[s * VKSIFT_MAX_TAPS], inst->ntaps[s], count, sp);
      if (fused_ds > 0)
        TRY(fused_ds, "blur + down-sampling");
      if (fused_ds == 0)
        *g0_done = true;
    }
    if (fused_ds < 0)
      TRY(vksift_hip_blur(srcp, dstp, &inst->taps[s * VKSIFT_MAX_TAPS], inst->ntaps[s], count, sp), "blur");
    nb_o++;
  }
  c->jobs[o].scan_reverse = inst->alt_order ? ((li & 1u) ^ 1u) : 0u;
#undef PL
  vksift_hip_range_pop();
  c->nblur_all += nb_o;
  if (o == 0)
  {
    c->nblur += nb_o;
    if (c->prof && (group_flags & PYR_LAST_GROUP))
      vksift_hip_event_record(c->PS->ev_pt[1], sp);
  }
  return 0;
}

/* Scales S+1 and S+2 of the octaves marked in c->tail[] among [o0, o1): they feed nothing but the extrema scan (scale S seeds the next octave,
 * they do not), so they wait until the chain seed -> ... -> scale S has been queued for every octave and go as ONE launch per SCALE over all
 * those octaves (vksift_hip_blur_multi: a flat multi-octave grid, csrc/hip/multi.h) — 2 launches instead of 2 per octave; shapes that kernel
 * does not serve take one launch per octave and scale as before. Same kernel body either way: bit-identical planes. */
static int enqueue_tail(DetectCtx *c, uint32_t o0, uint32_t o1, vksift_hip_stream sp)
{
  vksift_Instance inst = c->inst;
  const PyrLayout *L = c->L;
  for (uint32_t s = inst->S + 1u; s < inst->S + 3u; s++)
  {
    vksift_hip_Plane src[VKSIFT_MAX_OCTAVES], dst[VKSIFT_MAX_OCTAVES];
    uint32_t n = 0;
    for (uint32_t o = o0; o < o1 && o < L->n_oct; o++)
      if (c->tail[o])
      {
        src[n] = plane_at(inst, o, L->gauss_off[o], s - 1u);
        dst[n] = plane_at(inst, o, L->gauss_off[o], s);
        dst[n].reverse = inst->alt_order ? (s & 1u) : 0u;
        n++;
      }
    for (uint32_t i0 = 0; i0 < n; i0 += 8u)
    {
      const uint32_t k = n - i0 < 8u ? n - i0 : 8u;
      const int me = vksift_hip_tune_get(VKSIFT_TUNE_TAIL_MULTI) == 1
                         ? -1
                         : vksift_hip_blur_multi(src + i0, dst + i0, k, &inst->taps[s * VKSIFT_MAX_TAPS], inst->ntaps[s], c->count, sp);
      if (me > 0)
        TRY(me, "multi-octave blur");
      if (me == 0)
      {
        c->nblur_all++;
        continue;
      }
      for (uint32_t i = i0; i < i0 + k; i++)
      {
        TRY(vksift_hip_blur(src[i], dst[i], &inst->taps[s * VKSIFT_MAX_TAPS], inst->ntaps[s], c->count, sp), "blur");
        c->nblur_all++;
      }
    }
  }
  for (uint32_t o = o0; o < o1 && o < L->n_oct; o++)
    if (c->tail[o])
      c->jobs[o].scan_reverse = inst->alt_order ? (((inst->S + 2u) & 1u) ^ 1u) : 0u; /* opposite to the last launch */
  return 0;
}

/* Everything a detection puts on the GPU, from the image upload to the count read-back: the part a hipGraph captures.
 *
 * Two streams at most for a batch. The scale-space of all octaves is built on one (octave o+1 is seeded by scale S of octave o anyway;
 * small detections fork the scales behind S onto a side stream, see DetectCtx::fork: there the launch-to-launch latency is the cost,
 * not the bandwidth);
 * then every keypoint stage — ExtractKeypoints, ComputeOrientation, ComputeDescriptors — is ONE chain of launches for all
 * octaves on the instance stream (vksift_hip_*_multi), like the reference records the dispatches of all octaves of a stage into
 * one command buffer (sift_detector.c:1106-1259). With two pyramid buffers the scale-space has a stream of its own, ordered
 * behind the last reader of the buffer it recycles (detect_impl), so that the construction for detection N+1 runs beside the
 * matching of detection N.
 * Measured on MI355X (128 x 640x480 per call): anything more concurrent is not faster — per-octave chains on per-octave
 * streams (rounds 1-2) let the coarse octaves trickle through ~50 launches too small to fill the chip (2.9 ms of a 6.5 ms step
 * for a third of the pixels), eight hardware queues instead of four cost 10 %: memory-bound and VALU-bound kernels beside each
 * other take the sum of their times (the blur launches keep the VALUs half busy themselves). */
static int enqueue_detection(DetectCtx *c)
{
  vksift_Instance inst = c->inst;
  const PyrLayout *L = c->L;
  vksift_hip_stream st = inst->stream;
  vksift_hip_stream sp = c->overlap ? inst->pyr_stream : st;

  bool g0_done = false, oct0_done = false;
  if (c->upload)
  {
    /* The copies run on a stream of their own, behind the previous reader of d_input only (the seed pass of the previous
     * detection, whichever stream it ran on) — not behind the gates of the scale-space stream: the staging buffer is then free
     * again (ev_staging) as soon as the bus has taken the images, and a caller that queues the next batch early is not held up
     * until this detection's turn on the GPU has come. Captured sequences keep everything on the capturing stream.
     * The batch goes in groups: stage a group, queue its copy, queue octave 0's scale-space for THAT group behind the copy, stage
     * the next group meanwhile. 128 VGA frames are 39 MB = 1.6 ms on the bus: as one copy in front of whole-batch launches that
     * is 1.6 ms of idle GPU; in 4 groups of 32 the bus and the blur chain work side by side and octave 0 is complete 0.4 ms
     * after the last byte has arrived. (Groups below 32 frames lose more in launch efficiency than they hide.) */
    vksift_hip_stream su = c->capturing ? sp : inst->up_stream;
    if (inst->input_free_valid && !c->capturing)
      TRY(vksift_hip_stream_wait_event(su, inst->ev_input_free), "input buffer recycle");
    uint32_t per = (uint32_t)((((size_t)4 << 20) + c->img_bytes - 1) / c->img_bytes); /* >= 4 MB per copy */
    if (per < (c->count + VKSIFT_UP_GROUPS - 1u) / VKSIFT_UP_GROUPS)
      per = (c->count + VKSIFT_UP_GROUPS - 1u) / VKSIFT_UP_GROUPS;
    if (per < 32u)
      per = 32u;
    /* ... when the GPU would otherwise wait for the bus. With the previous detection still running (a caller that queues the next
     * batch before fetching the current one) the copies are hidden anyway, and whole-batch launches are the better launches:
     * 512 frames, pipelined protocol, 21.45 -> 21.85 k frames/s */
    const bool grouped = !c->capturing && L->n_oct > 0 && c->count >= 2u * per && !c->gpu_busy;
    if (!grouped)
      per = c->count;
    for (uint32_t i0 = 0, g = 0; i0 < c->count; g++)
    {
      uint32_t i1 = i0 + per;
      if (i1 >= c->count || c->count - i1 < per / 2u)
        i1 = c->count; /* a tail shorter than half a group joins the last one */
      if (!c->prestaged)
        stage_images(inst->h_input, c->images, i0, i1, c->img_bytes);
      if (!c->zero_copy)
        TRY(vksift_hip_memcpy_h2d(inst->d_input + (size_t)i0 * c->img_bytes, inst->h_input + (size_t)i0 * c->img_bytes, c->img_bytes * (i1 - i0), su), "image upload");
      if (grouped)
      {
        TRY(vksift_hip_event_record(inst->ev_up[g], su), "event record");
        TRY(vksift_hip_stream_wait_event(sp, inst->ev_up[g]), "image upload");
        TRY(enqueue_pyramid(c, 0, sp, i0, i1 - i0, (i0 == 0 ? PYR_FIRST_GROUP : 0) | (i1 == c->count ? PYR_LAST_GROUP : 0), &g0_done), "scale space construction");
        oct0_done = true;
      }
      i0 = i1;
    }
    if (!c->capturing && !c->zero_copy)
    {
      /* the pinned staging buffer is free again as soon as these copies have run; the seed pass waits for them */
      TRY(vksift_hip_event_record(inst->ev_staging, su), "event record");
      inst->staging_pending = true;
      if (!grouped)
        TRY(vksift_hip_stream_wait_event(sp, inst->ev_staging), "image upload");
    }
  }
  if (c->prof)
    vksift_hip_event_record(c->PS->ev_t[1], st);

  /* recClearBufferDataCmds (sift_detector.c:1081-1104); a forked detection clears on the side stream, beside the seed launch: the side
   * stream is ordered behind everything queued on the trunk stream so far — the previous detection's readers of the counters and masks —
   * and the trunk never waits for it (a fork in the middle of the trunk costs its next launch ~5 us) */
  if (!c->fork)
    TRY(vksift_hip_memset(inst->d_found + (size_t)c->first_buf * VKSIFT_MAX_OCTAVES, 0, sizeof(uint32_t) * VKSIFT_MAX_OCTAVES * c->count, st), "counter reset");
  else
  {
    TRY(vksift_hip_event_record(inst->ev_join[1], sp), "event record");
    TRY(vksift_hip_stream_wait_event(inst->pyr_stream, inst->ev_join[1]), "scale fork");
    TRY(vksift_hip_memset(inst->d_found + (size_t)c->first_buf * VKSIFT_MAX_OCTAVES, 0, sizeof(uint32_t) * VKSIFT_MAX_OCTAVES * c->count, inst->pyr_stream), "counter reset");
    if (L->n_oct > 0 && vksift_hip_clear_segment_masks(c->jobs, L->n_oct, c->count, inst->pyr_stream) == 0)
      for (uint32_t o = 0; o < L->n_oct; o++)
        c->jobs[o].masks_cleared = 1u;
  }

  /* The trailing octaves whose planes fit the LDS are built by ONE launch (vksift_hip_octave_chain: a workgroup per image walks all
   * their scales): from the first octave >= 1 behind which every octave qualifies. */
  uint32_t chain_from = L->n_oct;
  /* (forked = small detections only: in a batch the per-scale launches are faster — one workgroup per image keeps 16 waves on a CU for
   * 80 us where the launches spread an octave over the chip: 512 x 640x480 23.3 k frames/s with the chain, 23.6 k without) */
  if (inst->lds_chain && c->fork && !inst->fp16 && inst->S + 3u <= 8u)
  {
    while (chain_from > 1u && (L->w[chain_from - 1u] & 3u) == 0u && (uint64_t)L->w[chain_from - 1u] * L->h[chain_from - 1u] <= inst->lds_chain_max &&
           L->w[chain_from - 1u] >= 8u && L->h[chain_from - 1u] >= 8u && L->n_oct - (chain_from - 1u) <= 4u)
      chain_from--;
  }
  for (uint32_t o = oct0_done ? 1u : 0u; o < L->n_oct; o++)
  {
    if (o == chain_from)
    {
      vksift_hip_Plane layers[4 * 8];
      const uint32_t nl = inst->S + 3u, no = L->n_oct - o;
      for (uint32_t q = 0; q < no; q++)
        for (uint32_t l = 0; l < nl; l++)
          layers[q * nl + l] = plane_at(inst, o + q, L->gauss_off[o + q], l);
      if (!g0_done)
        TRY(vksift_hip_downsample(plane_at(inst, o - 1u, L->gauss_off[o - 1u], inst->S), layers[0], c->count, sp), "downsample");
      /* (lds_chain_refuse: test hook — S = nl is outside the shim's domain, so it declines exactly like a shape it does not cover) */
      const int ce = vksift_hip_octave_chain(layers, no, nl, inst->lds_chain_refuse ? nl : inst->S, inst->taps, inst->ntaps, c->count, sp);
      if (ce > 0)
        TRY(ce, "coarse-octave chain");
      if (ce == 0)
      {
        for (uint32_t q = o; q < L->n_oct; q++)
          c->jobs[q].scan_reverse = 0u;
        break;
      }
      /* not covered after all: the octave's seed is in place, the per-scale launches follow — for a forked detection trunk AND
       * branch, so the branch loop below has to reach these octaves too (it stops at chain_from) */
      g0_done = true;
      chain_from = L->n_oct;
    }
    /* a batch: octave 0 (and every octave whose last two scales take the four-texel kernel) in full, the coarser ones up to scale S */
    bool tail_o = c->fork;
    if (c->tail_batch && o >= 1u)
    {
      const vksift_hip_Plane p = plane_at(inst, o, L->gauss_off[o], 0);
      tail_o = vksift_hip_blur_form(p, p, inst->ntaps[inst->S + 1u], c->count) == 1 && vksift_hip_blur_form(p, p, inst->ntaps[inst->S + 2u], c->count) == 1;
    }
    c->tail[o] = tail_o;
    TRY(enqueue_pyramid(c, o, sp, 0, c->count, PYR_FIRST_GROUP | PYR_LAST_GROUP | (tail_o ? PYR_TRUNK : 0), &g0_done), "scale space construction");
    if (c->fork && (o + 1u == chain_from || o + 1u == L->n_oct))
      TRY(vksift_hip_event_record(inst->ev_fork[0], sp), "event record"); /* scale S of the last per-scale octave is queued: the tail may follow */
  }
  if (c->tail_batch)
    TRY(enqueue_tail(c, 1u, L->n_oct, sp), "scale space construction");
  if (c->zero_copy && !c->capturing)
  {
    /* the staging buffer was the seed launch's source: free again once that has run (octave 0's launches are on sp) */
    TRY(vksift_hip_event_record(inst->ev_staging, sp), "event record");
    inst->staging_pending = true;
  }

  if (c->fork)
  {
    /* Branch: the scales behind S of every forked octave on the side stream (which already holds the two clears) */
    vksift_hip_stream side = inst->pyr_stream;
    const uint32_t o0 = oct0_done ? 1u : 0u, o_end = chain_from < L->n_oct ? chain_from : L->n_oct;
    if (o0 < o_end)
    {
      /* ONE launch per scale over all forked octaves (round 6; one per octave and scale before: eight dependent launches on this stream
       * were the end of a 640x480 detection's scale-space, 173 us after its start — with two the LDS chain of the coarsest octave is) */
      TRY(vksift_hip_stream_wait_event(side, inst->ev_fork[0]), "scale fork");
      TRY(enqueue_tail(c, o0, o_end, side), "scale space construction");
    }
    /* the side stream rejoins (the clears at least are on it) */
    TRY(vksift_hip_event_record(inst->ev_join[0], side), "event record");
    TRY(vksift_hip_stream_wait_event(sp, inst->ev_join[0]), "scale join");
  }
  if (c->prof)
    vksift_hip_event_record(c->PS->ev_pt[2], sp); /* every octave's scale-space is queued (forked branches have joined) */
  if (c->overlap)
  {
    TRY(vksift_hip_event_record(inst->ev_pyr_done, sp), "event record");
    TRY(vksift_hip_stream_wait_event(st, inst->ev_pyr_done), "scale space ready");
  }
  if (L->n_oct > 0)
  {
    if (c->prof)
      vksift_hip_event_record(c->PS->ev_t[2], st);
    vksift_hip_range_push("ExtractKeypoints");
    TRY(vksift_hip_extract_keypoints_multi(c->jobs, L->n_oct, c->count, st, c->prof ? c->PS->ev_scan : NULL), "keypoint extraction");
    vksift_hip_range_pop();
    if (c->prof)
      vksift_hip_event_record(c->PS->ev_t[3], st);
    vksift_hip_range_push("ComputeOrientation");
    TRY(vksift_hip_orientations_multi(c->jobs, L->n_oct, c->count, st), "orientation");
    vksift_hip_range_pop();
    if (c->prof)
      vksift_hip_event_record(c->PS->ev_t[4], st);
    vksift_hip_range_push("ComputeDescriptors");
    if (c->dense || c->post)
    {
      /* the matcher's view of the buffers comes out of the same launch (pack_BufferMemory, sift_memory.c:957-1047): no gather pass;
       * so do the posted records of a single-image detection (vksift_internal.h: h_post): no pack launch behind the descriptors */
      const BufferInfo *b0 = &inst->bufs[c->first_buf];
      vksift_hip_DenseRows dr;
      memset(&dr, 0, sizeof(dr));
      if (c->dense)
      {
        dr.desc = inst->d_cache_desc + (uint64_t)c->first_buf * inst->desc_slot_stride, dr.desc_img_stride = inst->desc_slot_stride;
        dr.norm = inst->d_cache_norm + (uint64_t)c->first_buf * inst->cache_norm_stride, dr.norm_img_stride = inst->cache_norm_stride;
        dr.n = inst->d_cache_n + c->first_buf, dr.n_img_stride = 1;
      }
      if (c->post)
      {
        dr.post = inst->h_post[c->first_buf & 1u], dr.post_img_stride = 0;
        dr.found_post = inst->h_found + (size_t)c->first_buf * VKSIFT_MAX_OCTAVES, dr.found_post_n = VKSIFT_MAX_OCTAVES;
      }
      dr.nsec = b0->nb_sections;
      for (uint32_t o = 0; o < b0->nb_sections; o++)
        dr.sec_cap[o] = b0->sec_cap[o];
      TRY(vksift_hip_descriptors_multi_dense(c->jobs, L->n_oct, c->count, &dr, st), "descriptor");
    }
    else
      TRY(vksift_hip_descriptors_multi(c->jobs, L->n_oct, c->count, st), "descriptor");
    vksift_hip_range_pop();
    if (c->overlap)
    {
      /* the next detection's scale-space may start here: beside the matching that usually follows, not beside the descriptors.
       * Gates at the start of the orientation / descriptor stage, or none at all, give the same frames/s within 1 % and turn
       * every stage interval into a measurement of the contention instead of the kernel. */
      TRY(vksift_hip_event_record(inst->ev_desc_start, st), "event record");
      inst->desc_start_valid = true;
    }
    if (c->prof)
      vksift_hip_event_record(c->PS->ev_t[5], st);
  }
  else if (c->prof)
  {
    /* an image too small for a single octave launches nothing: the stage events of this call are recorded here, so that its
     * (zero) intervals are not measured against the events of an earlier detection */
    for (int i = 2; i <= 5; i++)
      vksift_hip_event_record(c->PS->ev_t[i], st);
    vksift_hip_event_record(c->PS->ev_scan, st);
    vksift_hip_event_record(c->PS->ev_pt[0], st);
    vksift_hip_event_record(c->PS->ev_pt[1], st);
    vksift_hip_event_record(c->PS->ev_pt[2], st);
  }
  if (inst->pyr_pingpong && !c->capturing)
  {
    /* everything that reads this call's pyramid runs on the instance stream (also recorded by the calls that do not overlap —
     * tiny images — so that a later overlapped call never recycles the buffer under them) */
    TRY(vksift_hip_event_record(inst->ev_pyr_free[inst->pyr_cur], st), "event record");
    inst->pyr_free_valid[inst->pyr_cur] = true;
  }
  inst->last_blur_launches = c->nblur, inst->last_blur_launches_all = c->nblur_all;
  /* profiling: the scale-space interval is octave 0's (77 % of the bytes), the scan interval covers the scan launch of all octaves */
  inst->last_alg_bytes = algorithmic_pyramid_bytes(inst, c->w, c->h, 1u) * c->count;
  /* "the extrema scan adds 20 B/px.octave" = one read of the S+2 DoG layers */
  inst->last_scan_bytes = 0;
  for (uint32_t o = 0; o < L->n_oct; o++)
    inst->last_scan_bytes += (uint64_t)L->w[o] * L->h[o] * pyr_texel_bytes(inst) * (inst->S + 2) * c->count;

  if (c->post)
    return 0; /* records and counters were posted by the descriptor launch */
  /* recCopySIFTCountCmds (sift_detector.c:1261-1291) */
  TRY(vksift_hip_post_words(inst->h_found + (size_t)c->first_buf * VKSIFT_MAX_OCTAVES, inst->d_found + (size_t)c->first_buf * VKSIFT_MAX_OCTAVES,
                            (size_t)VKSIFT_MAX_OCTAVES * c->count, st),
      "count read-back");
  return 0;
}

/* hipGraph replay (VKSIFT_GRAPH=1): the launch sequence depends only on (resolution, batch, first buffer, input pointer) —
 * counts and candidate lists live on the device — so it is captured once per such key and replayed with a single launch.
 * Returns the cache entry for the key (hit: ->exec != NULL) or the least recently used entry, emptied (miss). */
static DetectGraph *graph_lookup(vksift_Instance inst, const DetectCtx *c)
{
  DetectGraph *victim = &inst->graphs[0];
  for (int i = 0; i < VKSIFT_GRAPH_CACHE; i++)
  {
    DetectGraph *g = &inst->graphs[i];
    if (g->exec && g->w == c->w && g->h == c->h && g->count == c->count && g->first_buf == c->first_buf && g->d_src == c->d_src && g->post == c->post &&
        g->dense == c->dense)
      return g;
    if (g->stamp < victim->stamp)
      victim = g;
  }
  vksift_hip_graph_destroy(victim->exec);
  memset(victim, 0, sizeof(*victim));
  return victim;
}

static void detect_impl(vksift_Instance inst, const uint8_t *const *images, const uint8_t *d_images, bool prestaged, uint32_t count, uint32_t w, uint32_t h,
                        uint32_t first_buf, const char *fn)
{
  vksift_hip_stream st = inst->stream;
  bool capturing = false;
  bool seq_assigned = false; /* the target buffers already carry the sequence number of this (not yet queued) detection */

  bool valid = count >= 1 && count <= inst->det_cap && buffer_idx_valid(inst, first_buf) && buffer_idx_valid(inst, first_buf + count - 1) &&
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
   * image staging buffer, and (when profiling) the event set of the detection before the previous one. */
  if (inst->staging_pending && images) /* (prestaged images: the deferring call waited before it wrote the first one) */
  {
    HIP_CHECK(vksift_hip_event_sync(inst->ev_staging), "staging synchronisation");
    inst->staging_pending = false;
  }
  ProfSet *PS = &inst->prof[inst->prof_cur];
  if (inst->profiling)
  {
    /* recycle the older event set: the host never waits for the call it just queued */
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
    /* n_oct == 0 (shortest side below 32 pixels, 16 with up-sampling): no scale-space, the detection finds nothing */
    if ((L.img_floats > inst->pyr_img_stride || L.seg_total > inst->seg_cap || L.cand_total > inst->cand_cap) && grow_image_scratch(inst, &L) != 0)
    {
      logError(LOG_TAG, "Failed to fit the scale-space of a %ux%u image in device memory", w, h);
      goto gpu_error;
    }
    inst->lay = L;
    inst->cur_w = w;
    inst->cur_h = h;
  }
  inst->cur_batch = count;
  const uint64_t seq = inst->det_seq + 1u;
  for (uint32_t i = 0; i < count; i++)
  {
    set_buffer_sections(inst, first_buf + i, inst->lay.n_oct, w, h);
    inst->bufs[first_buf + i].seq = seq;       /* its counters are valid once detection `seq` has completed */
    inst->cache_valid[first_buf + i] = false; /* the matcher's view of the buffer is rebuilt on its next matching */
  }
  seq_assigned = true;

  DetectCtx c;
  c.inst = inst, c.L = &inst->lay, c.PS = PS;
  c.prof = inst->profiling;
  /* Overlapping detections (VKSIFT_PYR_PINGPONG=1): with two pyramid buffers the scale-space construction of this call
   * does not depend on anything the previous call (or a matching still in flight) reads or writes, so it runs on its own
   * stream, ordered only behind the last reader of the pyramid buffer it recycles; everything that touches the SIFT
   * buffers and the extraction scratch stays in instance-stream order. */
  c.overlap = inst->pyr_pingpong && c.L->n_oct > 0 && count >= inst->overlap_min_count;
  c.upload = images != NULL || prestaged;
  c.prestaged = prestaged;
  c.w = w, c.h = h, c.count = count, c.first_buf = first_buf;
  c.img_bytes = (size_t)w * h;
  c.nblur = 0, c.nblur_all = 0;
  memset(c.tail, 0, sizeof(c.tail));
  /* (VKSIFT_TUNE_TAIL_MULTI = 1: every octave in full, one launch per octave and scale — A/B and the bit-identity matrix) */
  c.tail_batch = !c.fork && count >= 8u && c.L->n_oct > 1u && vksift_hip_tune_get(VKSIFT_TUNE_TAIL_MULTI) == 0;
  c.capturing = false;
  c.gpu_busy = detect_running(inst);
  /* forked scale-space + LDS chain are latency measures for ONE image (or a handful): a batch on a single-buffer instance
   * (batch_cap < 8 or VKSIFT_PYR_PINGPONG=0) fills the chip with its per-scale launches and takes those */
  c.fork = inst->fork_scales && !c.overlap && !c.prof && count <= VKSIFT_FORK_MAX_COUNT && (uint64_t)count * w * h <= inst->fork_max_pixels;
  /* feature posting for single-image detections whose records fit the slot (every section is capacity-bounded) */
  /* once the instance has matched (its cache blocks exist) a detection leaves the matcher's rows of its buffers behind itself */
  c.dense = inst->d_cache_desc != NULL && inst->d_cache_norm != NULL && c.L->n_oct > 0 && c.L->n_oct == inst->bufs[first_buf].nb_sections &&
            c.L->n_oct <= 16u && vksift_hip_tune_get(VKSIFT_TUNE_DENSE_ROWS) == 0;
  c.post = false;
  if (count == 1 && inst->post_enabled && inst->post_on && c.L->n_oct > 0 && inst->bufs[first_buf].nb_sections > 0 && inst->bufs[first_buf].nb_sections <= 16)
  {
    if (!inst->h_post[0])
    {
      const size_t cap = (size_t)inst->cfg.max_nb_sift_per_buffer * FEAT_BYTES + 4096u;
      inst->h_post[0] = (uint8_t *)vksift_hip_host_malloc(cap);
      inst->h_post[1] = (uint8_t *)vksift_hip_host_malloc(cap);
      inst->post_cap = (inst->h_post[0] && inst->h_post[1]) ? cap : 0;
      if (!inst->post_cap)
      {
        vksift_hip_host_free(inst->h_post[0]);
        vksift_hip_host_free(inst->h_post[1]);
        inst->h_post[0] = inst->h_post[1] = NULL;
        inst->post_enabled = false;
      }
    }
    uint64_t rows = 0;
    for (uint32_t o = 0; o < inst->bufs[first_buf].nb_sections; o++)
      rows += inst->bufs[first_buf].sec_cap[o];
    c.post = inst->post_cap != 0 && rows * FEAT_BYTES <= inst->post_cap;
    if (c.post)
    {
      const uint32_t slot = first_buf & 1u;
      if (inst->post_seq[slot] != 0 && !inst->post_fetched[slot] && ++inst->post_idle >= VKSIFT_POST_IDLE)
        inst->post_on = false, c.post = false; /* posted and overwritten without ever being fetched, too many times in a row */
      inst->post_seq[slot] = 0; /* the slot is about to be rewritten: valid again once this detection is queued */
    }
  }
  PS->overlap = true; /* the scale-space interval is the one between ev_pt[0] and ev_pt[1] (octave 0) */
  if (c.prof)
    vksift_hip_event_record(PS->ev_t[0], st);
  if (c.overlap)
  {
    if (inst->pyr_nbuf == 2u)
      inst->pyr_cur ^= 1;
    inst->d_pyr = inst->d_pyr_buf[inst->pyr_cur];
    if (inst->pyr_free_valid[inst->pyr_cur])
      HIP_CHECK(vksift_hip_stream_wait_event(inst->pyr_stream, inst->ev_pyr_free[inst->pyr_cur]), "pyramid buffer recycle");
    /* not before the previous detection's descriptors are done (see enqueue_detection) */
    if (inst->desc_start_valid)
      HIP_CHECK(vksift_hip_stream_wait_event(inst->pyr_stream, inst->ev_desc_start), "overlap gate");
  }

  /* host images are staged into pinned memory while the sequence is enqueued (enqueue_detection): the caller may reuse its
   * memory as soon as we return (sift_memory.c:943) */
  c.images = images;
  /* One host image (at most 1 MB): no copy into device memory first — the fused up-sampling + seed launch reads every source byte once, and
   * reads them out of the pinned staging buffer over the bus (300 KB: ~6 us of bus time inside a 9 us launch) instead of behind a 9 us copy */
  c.zero_copy = c.upload && count == 1u && c.img_bytes <= ((size_t)1 << 20) && vksift_hip_tune_get(VKSIFT_TUNE_ZERO_COPY) == 0;
  c.d_src = c.upload ? (c.zero_copy ? inst->h_input : inst->d_input) : d_images;
  build_jobs(&c);

  /* host-visible events (staging, completion, profiling) stay outside a captured region; a captured graph holds the address
   * of ONE pyramid buffer, so instances with two (ping-pong) never replay, nor does a detection whose scale-space overlaps */
  const bool replay = inst->use_graphs && !c.prof && !c.overlap && !(inst->pyr_pingpong && inst->pyr_nbuf == 2u) &&
                      (uint64_t)count * w * h <= inst->graph_max_pixels;
  DetectGraph *dg = replay ? graph_lookup(inst, &c) : NULL;
  if (dg && dg->exec)
  {
    if (images) /* the upload is a node of the graph: the staging buffer has to be filled before the replay */
      stage_images(inst->h_input, images, 0, count, c.img_bytes);
    HIP_CHECK(vksift_hip_graph_launch(dg->exec, st), "detection graph launch");
    inst->graph_miss_run = 0;
  }
  else
  {
    if (dg && ++inst->graph_miss_run > 4u * VKSIFT_GRAPH_CACHE)
    {
      /* the caller keeps changing shape / buffer / input pointer: captures would only add cost */
      inst->use_graphs = false;
      dg = NULL;
    }
    if (dg)
    {
      if (vksift_hip_capture_begin(st) == 0)
        capturing = c.capturing = true;
      else
        dg = NULL;
    }
    if (enqueue_detection(&c) != 0)
      goto gpu_error;
    if (capturing)
    {
      capturing = false;
      vksift_hip_graph exec = NULL;
      HIP_CHECK(vksift_hip_capture_end(st, &exec), "detection graph capture");
      dg->exec = exec;
      dg->w = w, dg->h = h, dg->count = count, dg->first_buf = first_buf, dg->d_src = c.d_src, dg->post = c.post, dg->dense = c.dense;
      HIP_CHECK(vksift_hip_graph_launch(dg->exec, st), "detection graph launch");
    }
  }
  if (dg)
    dg->stamp = ++inst->graph_stamp;
  if (dg && inst->pyr_pingpong)
  {
    /* (recorded by enqueue_detection for every sequence that is not captured) a later overlapped detection recycles the buffer
     * behind this one's readers */
    HIP_CHECK(vksift_hip_event_record(inst->ev_pyr_free[inst->pyr_cur], st), "event record");
    inst->pyr_free_valid[inst->pyr_cur] = true;
  }
  if (c.upload && dg)
  {
    /* graph replay: the upload is a node of the graph, the staging buffer is busy until the graph has run */
    HIP_CHECK(vksift_hip_event_record(inst->ev_staging, st), "event record");
    inst->staging_pending = true;
  }
  inst->device_input_last = !c.upload;
  if (c.dense)
    for (uint32_t i = 0; i < count; i++)
      inst->cache_valid[first_buf + i] = true; /* in stream order in front of every matching queued from here on */
  if (c.prof)
  {
    vksift_hip_event_record(PS->ev_t[6], st);
    PS->valid = true;
    PS->accounted = false;
    PS->blur_launches = inst->last_blur_launches, PS->blur_launches_all = inst->last_blur_launches_all;
    PS->alg_bytes = inst->last_alg_bytes;
    PS->scan_bytes = inst->last_scan_bytes;
  }
  {
    DetectSlot *d = &inst->det_ring[seq % VKSIFT_DETECT_RING];
    d->seq = seq, d->first = first_buf, d->count = count;
    if (c.post)
      inst->post_seq[first_buf & 1u] = seq, inst->post_buf[first_buf & 1u] = first_buf, inst->post_fetched[first_buf & 1u] = false;
    inst->det_seq = seq; /* from here on the buffers are "pending" even if the record below fails (wait_detect_seq then syncs a stale event: harmless) */
    HIP_CHECK(vksift_hip_event_record(d->ev, st), "event record");
  }
  return;

gpu_error:
  if (capturing)
  {
    vksift_hip_graph dead = NULL;
    (void)vksift_hip_capture_end(st, &dead);
    vksift_hip_graph_destroy(dead);
  }
  if (seq_assigned)
  {
    /* The detection never got its sequence number (det_seq and the ring slot advance on success only): left as they are the
     * buffers would wait for a detection that does not exist — vksift_isBufferAvailable() false for ever, the accessors syncing
     * a foreign ring event — and the next successful detection would alias the number. They become completed, EMPTY buffers:
     * whatever part of the chain was queued may still write counters, so the stream is drained before the host mirror is zeroed. */
    (void)vksift_hip_stream_sync(st);
    if (inst->pyr_stream)
      (void)vksift_hip_stream_sync(inst->pyr_stream);
    if (inst->side_stream)
      (void)vksift_hip_stream_sync(inst->side_stream);
    for (uint32_t i = 0; i < count; i++)
    {
      inst->bufs[first_buf + i].seq = 0;
      memset(inst->h_found + (size_t)(first_buf + i) * VKSIFT_MAX_OCTAVES, 0, sizeof(uint32_t) * VKSIFT_MAX_OCTAVES);
    }
    (void)vksift_hip_memset(inst->d_found + (size_t)first_buf * VKSIFT_MAX_OCTAVES, 0, sizeof(uint32_t) * VKSIFT_MAX_OCTAVES * count, st);
  }
  logError(LOG_TAG, "%s error: Failed to start the detection pipeline.", fn);
  inst->error_cb(VKSIFT_VULKAN_ERROR);
}

/* ------------------------------------------------------------------------------------------------ */
/* deferred submission (vksift_internal.h: defer_enabled)                                           */
/* ------------------------------------------------------------------------------------------------ */
/* the staged images as ONE batched detection. Failures are reported through the error callback of whichever call got here. */
void flush_deferred(vksift_Instance inst)
{
  const uint32_t n = inst->pend_n;
  if (n == 0)
    return;
  inst->pend_n = 0;
  inst->defer_batches++;
  inst->defer_images += n;
  vksift_hip_set_device(inst->device);
  detect_impl(inst, NULL, NULL, true, n, inst->pend_w, inst->pend_h, inst->pend_first, "vksift_detectFeatures()");
}

/* true: the image was staged (and the batch launched if that filled it); false: the caller launches it the direct way */
static bool defer_detect(vksift_Instance inst, const uint8_t *image, uint32_t w, uint32_t h, uint32_t buf)
{
  if (inst->pend_n)
  {
    /* a batch is one resolution into consecutive buffers; a buffer named twice keeps the order of its two detections */
    if (buf != inst->pend_first + inst->pend_n || w != inst->pend_w || h != inst->pend_h)
      flush_deferred(inst);
  }
  else if (inst->defer_grow && inst->det_cap < inst->defer_max)
  {
    /* the previous batch filled the capacity: twice as much for this one (the blocks follow what the caller does: an instance
     * with 1000 SIFT buffers whose caller detects two images at a time holds the scratch of two) */
    uint32_t cap = inst->det_cap * 2u;
    cap = cap > inst->defer_max ? inst->defer_max : cap;
    /* ... within a third of what the device has left */
    const uint64_t per_image = pyr_texel_bytes(inst) * inst->pyr_img_stride * inst->pyr_nbuf + 12u * inst->seg_cap + 8u * inst->cand_cap +
                               2u * (uint64_t)inst->max_image_size + (4u * VKSIFT_HIP_MAX_ORI + 4u) * inst->ori_cap;
    const uint64_t room = vksift_hip_device_free_mem() / 3u;
    inst->defer_grow = false;
    if ((uint64_t)(cap - inst->det_cap) * per_image > room || resize_detect_scratch(inst, NULL, cap) != 0)
      inst->defer_max = inst->det_cap; /* this is as far as it goes */
  }
  if (inst->det_cap < 2u || inst->h_input == NULL)
  {
    /* a single-image instance: room for two first (then doubling, see above). The second call of a run pays for it, once. */
    if (inst->h_input == NULL || resize_detect_scratch(inst, NULL, 2u) != 0)
    {
      inst->defer_enabled = false;
      return false;
    }
  }
  if (inst->pend_n == 0)
  {
    if (inst->staging_pending)
    {
      if (vksift_hip_event_sync(inst->ev_staging) != 0)
        return false;
      inst->staging_pending = false;
    }
    inst->pend_first = buf, inst->pend_w = w, inst->pend_h = h;
  }
  memcpy(inst->h_input + (size_t)inst->pend_n * w * h, image, (size_t)w * h);
  inst->pend_n++;
  const uint32_t full = inst->det_cap < inst->defer_max ? inst->det_cap : inst->defer_max;
  if (inst->pend_n >= full)
  {
    inst->defer_grow = inst->det_cap < inst->defer_max;
    flush_deferred(inst);
  }
  else if (inst->defer_chunk && inst->pend_n >= inst->defer_chunk && !detect_running(inst))
  {
    /* an idle GPU and a worthwhile number of staged images: launch them now, beside the staging of the rest of the caller's run (the
     * strictly serial pattern "detect into N buffers, then read them" otherwise leaves the GPU idle for the whole staging phase and the host
     * idle for the whole detection). With a detection in flight — the pattern with two buffer sets — nothing is launched early: whole runs
     * make the better batches. */
    inst->defer_grow = inst->det_cap < inst->defer_max; /* the caller's runs are longer than this chunk */
    flush_deferred(inst);
  }
  return true;
}

void vksift_detectFeatures(vksift_Instance instance, const uint8_t *image_data, const uint32_t image_width, const uint32_t image_height,
                           const uint32_t gpu_buffer_id)
{
  vksift_Instance inst = instance;
  const uint8_t *imgs[1] = {image_data};
  vksift_hip_set_device(inst->device);
  /* The first detection after any other call is launched at once — detect + read, the reference's own loop
   * (src/perf/wrappers/vulkansift_wrapper.cpp:30-33), and the two-buffer ping-pong keep their path and their latency — unless the
   * caller's last run of detect calls held several. From the second call of a run on the images are staged and go as one batch. */
  const bool run = inst->epoch_detects > 0 || inst->batch_mode;
  inst->epoch_detects++;
  if (inst->defer_enabled && (run || inst->pend_n) && !inst->profiling && image_data != NULL && buffer_idx_valid(inst, gpu_buffer_id) &&
      resolution_valid(inst, image_width, image_height) && (image_width < image_height ? image_width : image_height) >= 16u)
  {
    if (defer_detect(inst, image_data, image_width, image_height, gpu_buffer_id))
      return;
  }
  /* invalid arguments take the direct path too: it reports them */
  if (inst->pend_n)
    flush_deferred(inst);
  detect_impl(inst, imgs, NULL, false, 1, image_width, image_height, gpu_buffer_id, "vksift_detectFeatures()");
}

static bool ext_batch_count_valid(vksift_Instance inst, uint32_t count, const char *fn)
{
  if (count <= inst->batch_cap)
    return true;
  logError(LOG_TAG, "%s error: invalid input.", fn);
  inst->error_cb(VKSIFT_INVALID_INPUT_ERROR);
  return false;
}

void vksift_ext_detectFeaturesBatch(vksift_Instance instance, const uint8_t *const *images, uint32_t count, uint32_t image_width, uint32_t image_height,
                                    uint32_t first_gpu_buffer_id)
{
  vksift_hip_set_device(instance->device);
  defer_sync(instance);
  if (ext_batch_count_valid(instance, count, "vksift_ext_detectFeaturesBatch()"))
    detect_impl(instance, images, NULL, false, count, image_width, image_height, first_gpu_buffer_id, "vksift_ext_detectFeaturesBatch()");
}

void vksift_ext_detectFeaturesBatchDevice(vksift_Instance instance, const uint8_t *d_images, uint32_t count, uint32_t image_width, uint32_t image_height,
                                          uint32_t first_gpu_buffer_id)
{
  vksift_hip_set_device(instance->device);
  defer_sync(instance);
  if (d_images == NULL)
  {
    logError(LOG_TAG, "vksift_ext_detectFeaturesBatchDevice() error: invalid input.");
    instance->error_cb(VKSIFT_INVALID_INPUT_ERROR);
    return;
  }
  if (ext_batch_count_valid(instance, count, "vksift_ext_detectFeaturesBatchDevice()"))
    detect_impl(instance, NULL, d_images, false, count, image_width, image_height, first_gpu_buffer_id, "vksift_ext_detectFeaturesBatchDevice()");
}
