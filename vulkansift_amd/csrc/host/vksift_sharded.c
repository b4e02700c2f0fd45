/*
 * vksift_sharded.c — the multi-GPU matcher behind a C entry point (BASELINE.json north_star: "the matcher shards
 * the query set across the GPUs of one node with an RCCL all-gather of the reference descriptors over xGMI").
 *
 * The reference is single-GPU (vulkansift.h:32-34): Get2NearestNeighbors.comp (sift_matcher.c:246-279) scans ALL of B in index
 * order for every row of A. Sharding the ROWS OF A keeps that scan — and with it the tie rules (quirk Q7, strict '<') — intact,
 * so the records are bit-identical for every world size. The only exchange is one all-gather of B's descriptor rows (uint8,
 * 128 B each; 6.4 MB for 50 k rows): issued first, on its own stream, while the instance stream runs the norm pre-pass of the
 * local A rows; the B norms and the MFMA matcher follow once the gather has landed.
 *
 * RCCL is loaded lazily (dlopen of librccl.so) so that single-GPU users of libvulkansift.so do not depend on it. One process per
 * GPU; the 128-byte unique id travels from rank 0 to the other ranks by whatever host channel the application has
 * (torch.distributed / MPI / a socket).
 */
#define _GNU_SOURCE /* RTLD_DEFAULT */
#include <dlfcn.h>

#include "vksift_internal.h"

typedef struct
{
  char internal[128];
} rccl_UniqueId; /* == ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128) */
typedef void *rccl_Comm;
enum
{
  RCCL_UINT8 = 1 /* ncclUint8 */
};

static struct
{
  void *so;
  int (*GetUniqueId)(rccl_UniqueId *);
  int (*CommInitRank)(rccl_Comm *, int, rccl_UniqueId, int);
  int (*CommDestroy)(rccl_Comm);
  int (*CommAbort)(rccl_Comm);
  int (*AllGather)(const void *, void *, size_t, int, rccl_Comm, void *);
  int (*CommCount)(const rccl_Comm, int *);    /* optional: what the communicator itself says about its size ... */
  int (*CommUserRank)(const rccl_Comm, int *); /* ... and this rank (vksift_ext_shardGroupInfo) */
  const char *(*GetErrorString)(int);
} g_rccl;
#define RTLD_DEFAULT_SENTINEL ((void *)&g_rccl) /* "bound through the global scope": nothing to dlclose */

static bool rccl_bind(void *so)
{
  *(void **)&g_rccl.GetUniqueId = dlsym(so, "ncclGetUniqueId");
  *(void **)&g_rccl.CommInitRank = dlsym(so, "ncclCommInitRank");
  *(void **)&g_rccl.CommDestroy = dlsym(so, "ncclCommDestroy");
  *(void **)&g_rccl.CommAbort = dlsym(so, "ncclCommAbort");
  *(void **)&g_rccl.AllGather = dlsym(so, "ncclAllGather");
  *(void **)&g_rccl.GetErrorString = dlsym(so, "ncclGetErrorString");
  *(void **)&g_rccl.CommCount = dlsym(so, "ncclCommCount");
  *(void **)&g_rccl.CommUserRank = dlsym(so, "ncclCommUserRank");
  return g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllGather;
}

/* Order of preference: (1) an RCCL the process has ALREADY loaded — in a torch.distributed process that is the copy torch bundles;
 * a second copy would mean two communicator runtimes in one process — found through the global symbol scope or by asking the
 * loader for the resident library without loading anything (RTLD_NOLOAD); (2) a private load (RTLD_LOCAL: our copy must not
 * interpose nccl* symbols other libraries resolve later), versioned name first (runtime-only ROCm installs ship no librccl.so). */
static bool rccl_load(void)
{
  if (g_rccl.so)
    return true;
  void *so = NULL;
  if (dlsym(RTLD_DEFAULT, "ncclAllGather") != NULL && rccl_bind(RTLD_DEFAULT))
  {
    g_rccl.so = RTLD_DEFAULT_SENTINEL;
    return true;
  }
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (unsigned i = 0; i < sizeof(names) / sizeof(names[0]) && !so; i++)
    so = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);
  for (unsigned i = 0; i < sizeof(names) / sizeof(names[0]) && !so; i++)
    so = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!so)
  {
    logError(LOG_TAG, "RCCL not found (librccl.so.1): %s", dlerror());
    return false;
  }
  if (!rccl_bind(so))
  {
    logError(LOG_TAG, "librccl.so lacks a required entry point");
    dlclose(so);
    memset(&g_rccl, 0, sizeof(g_rccl));
    return false;
  }
  g_rccl.so = so;
  return true;
}

struct vksift_ext_ShardGroup_T
{
  int device;
  uint32_t world, rank;
  rccl_Comm comm;
  vksift_ext_AllGatherFn transport; /* non-NULL: the application's own all-gather stands in for ncclAllGather (comm == NULL) */
  void *transport_user;
  vksift_hip_stream stream, comm_stream;
  vksift_hip_event ev_fork, ev_gathered, ev_t0, ev_t1;
  uint8_t *d_b_full;
  uint32_t *d_scratch;
  size_t b_cap, scratch_cap; /* bytes / u32 elements */
  uint32_t comm_ranks, comm_rank; /* ncclCommCount / ncclCommUserRank of the communicator (0 / 0: a transport group, or an RCCL without them) */
  bool timed;
  bool broken; /* the communicator was aborted after a local failure: every later call fails */
};

vksift_Result vksift_ext_shardGetUniqueId(uint8_t id[VKSIFT_EXT_SHARD_ID_BYTES])
{
  if (!id || !rccl_load())
    return VKSIFT_VULKAN_ERROR;
  rccl_UniqueId u;
  const int e = g_rccl.GetUniqueId(&u);
  if (e != 0)
  {
    logError(LOG_TAG, "ncclGetUniqueId failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    return VKSIFT_VULKAN_ERROR;
  }
  memcpy(id, u.internal, VKSIFT_EXT_SHARD_ID_BYTES);
  return VKSIFT_SUCCESS;
}

/* id != NULL: RCCL communicator (collective: ncclCommInitRank); id == NULL: the caller's transport */
static vksift_Result shard_group_create(vksift_ext_ShardGroup *out, int gpu_device_index, uint32_t world, uint32_t rank, const uint8_t *id,
                                        vksift_ext_AllGatherFn transport, void *transport_user)
{
  if (!out || *out != NULL || world == 0 || rank >= world || (!id && !transport))
    return VKSIFT_INVALID_INPUT_ERROR;
  if (!vksift_g_loaded || (id && !rccl_load()))
    return VKSIFT_VULKAN_ERROR;
  if (gpu_device_index < 0 || gpu_device_index >= vksift_hip_device_count() || vksift_hip_set_device(gpu_device_index) != 0)
    return VKSIFT_VULKAN_ERROR;
  vksift_ext_ShardGroup g = (vksift_ext_ShardGroup)calloc(1, sizeof(*g));
  if (!g)
    return VKSIFT_VULKAN_ERROR;
  g->device = gpu_device_index, g->world = world, g->rank = rank;
  g->transport = id ? NULL : transport, g->transport_user = transport_user;
  if (id)
  {
    rccl_UniqueId u;
    memcpy(u.internal, id, VKSIFT_EXT_SHARD_ID_BYTES);
    const int e = g_rccl.CommInitRank(&g->comm, (int)world, u, (int)rank);
    if (e != 0)
    {
      logError(LOG_TAG, "ncclCommInitRank failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
      free(g);
      return VKSIFT_VULKAN_ERROR;
    }
    /* the communicator's own view of the job: a group whose RCCL communicator disagrees with the (world, rank) it was created
     * for would all-gather into the wrong slots — refused here, and reported to the caller through vksift_ext_shardGroupInfo */
    int cn = 0, cr = 0;
    if (g_rccl.CommCount && g_rccl.CommUserRank && g_rccl.CommCount(g->comm, &cn) == 0 && g_rccl.CommUserRank(g->comm, &cr) == 0)
    {
      g->comm_ranks = (uint32_t)cn, g->comm_rank = (uint32_t)cr;
      if (g->comm_ranks != world || g->comm_rank != rank)
      {
        logError(LOG_TAG, "vksift_ext_shardGroupCreate() failure: the RCCL communicator reports rank %d of %d, the group was created as rank %u of %u", cr, cn,
                 rank, world);
        if (g_rccl.CommAbort)
          g_rccl.CommAbort(g->comm);
        free(g);
        return VKSIFT_VULKAN_ERROR;
      }
    }
  }
  g->stream = vksift_hip_stream_create();
  g->comm_stream = vksift_hip_stream_create();
  g->ev_fork = vksift_hip_event_create();
  g->ev_gathered = vksift_hip_event_create();
  g->ev_t0 = vksift_hip_event_create();
  g->ev_t1 = vksift_hip_event_create();
  if (!g->stream || !g->comm_stream || !g->ev_fork || !g->ev_gathered || !g->ev_t0 || !g->ev_t1)
  {
    /* a NULL stream would silently mean the default stream: fail the creation instead (the communicator exists on the peers too:
     * abort ours, they notice at their first collective like for any rank that disappears) */
    logError(LOG_TAG, "vksift_ext_shardGroupCreate() failure: stream / event creation");
    vksift_ext_shardGroupDestroy(&g);
    return VKSIFT_VULKAN_ERROR;
  }
  *out = g;
  return VKSIFT_SUCCESS;
}

vksift_Result vksift_ext_shardGroupCreate(vksift_ext_ShardGroup *out, int gpu_device_index, uint32_t world, uint32_t rank,
                                          const uint8_t id[VKSIFT_EXT_SHARD_ID_BYTES])
{
  if (!id)
    return VKSIFT_INVALID_INPUT_ERROR;
  return shard_group_create(out, gpu_device_index, world, rank, id, NULL, NULL);
}

vksift_Result vksift_ext_shardGroupCreateWithTransport(vksift_ext_ShardGroup *out, int gpu_device_index, uint32_t world, uint32_t rank,
                                                       vksift_ext_AllGatherFn all_gather, void *user)
{
  if (!all_gather)
    return VKSIFT_INVALID_INPUT_ERROR;
  return shard_group_create(out, gpu_device_index, world, rank, NULL, all_gather, user);
}

void vksift_ext_shardGroupInfo(vksift_ext_ShardGroup g, uint32_t *world, uint32_t *rank, uint32_t *rccl_ranks, uint32_t *rccl_rank)
{
  if (world)
    *world = g ? g->world : 0;
  if (rank)
    *rank = g ? g->rank : 0;
  if (rccl_ranks)
    *rccl_ranks = g ? g->comm_ranks : 0;
  if (rccl_rank)
    *rccl_rank = g ? g->comm_rank : 0;
}

void vksift_ext_shardGroupLayout(uint32_t n_total, uint32_t world, uint32_t rank, uint32_t *block_rows, uint32_t *first_row, uint32_t *nb_rows)
{
  /* equal blocks of ceil(n_total / world) rows: what vksift_ext_matchSharded all-gathers (the last blocks may be short or empty) */
  const uint32_t blk = world ? (uint32_t)(((uint64_t)n_total + world - 1) / world) : 0;
  const uint64_t lo64 = (uint64_t)rank * blk;
  const uint32_t lo = lo64 < n_total ? (uint32_t)lo64 : n_total;
  const uint32_t hi = (uint64_t)lo + blk < n_total ? lo + blk : n_total;
  if (block_rows)
    *block_rows = blk;
  if (first_row)
    *first_row = lo;
  if (nb_rows)
    *nb_rows = hi - lo;
}

/* the exchange: ncclAllGather on the communicator, or the application's transport */
static int shard_all_gather(vksift_ext_ShardGroup g, const void *send, void *recv, size_t bytes_per_rank, vksift_hip_stream s)
{
  if (g->transport)
  {
    const int te = g->transport(g->transport_user, send, recv, bytes_per_rank, g->rank, g->world, s);
    if (te != 0)
      logError(LOG_TAG, "the all-gather transport failed (%d)", te);
    return te;
  }
  const int ne = g_rccl.AllGather(send, recv, bytes_per_rank, RCCL_UINT8, g->comm, s);
  if (ne != 0)
    logError(LOG_TAG, "ncclAllGather failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(ne) : "?");
  return ne;
}

void vksift_ext_shardGroupDestroy(vksift_ext_ShardGroup *gp)
{
  if (!gp || !*gp)
    return;
  vksift_ext_ShardGroup g = *gp;
  vksift_hip_set_device(g->device);
  if (g->comm_stream)
    vksift_hip_stream_sync(g->comm_stream);
  if (g->stream)
    vksift_hip_stream_sync(g->stream);
  if (g->comm && g_rccl.CommDestroy)
    g_rccl.CommDestroy(g->comm);
  vksift_hip_free(g->d_b_full);
  vksift_hip_free(g->d_scratch);
  vksift_hip_event_destroy(g->ev_fork);
  vksift_hip_event_destroy(g->ev_gathered);
  vksift_hip_event_destroy(g->ev_t0);
  vksift_hip_event_destroy(g->ev_t1);
  vksift_hip_stream_destroy(g->comm_stream);
  vksift_hip_stream_destroy(g->stream);
  free(g);
  *gp = NULL;
}

/* Device scratch for na query rows against nb_rows (padded) reference rows. Returns 0, or a bit mask: 1 = the all-gather's
 * receive buffer is missing, 2 = the matcher's scratch is missing. */
static int shard_reserve(vksift_ext_ShardGroup g, uint32_t na, size_t nb_rows)
{
  const size_t b_bytes = nb_rows * 128u;
  /* norms of A, norms of B (padded rows included), redo flags, partial lists of the stream-decomposed kernel */
  const size_t scratch = vksift_hip_match_scratch_u32(na, (uint32_t)nb_rows);
  if (b_bytes <= g->b_cap && scratch <= g->scratch_cap)
    return 0;
  vksift_hip_stream_sync(g->comm_stream);
  vksift_hip_stream_sync(g->stream);
  if (b_bytes > g->b_cap)
  {
    vksift_hip_free(g->d_b_full);
    g->d_b_full = (uint8_t *)vksift_hip_malloc(b_bytes);
    g->b_cap = g->d_b_full ? b_bytes : 0;
  }
  if (scratch > g->scratch_cap)
  {
    vksift_hip_free(g->d_scratch);
    g->d_scratch = (uint32_t *)vksift_hip_malloc(scratch * sizeof(uint32_t));
    g->scratch_cap = g->d_scratch ? scratch : 0;
  }
  return (g->d_b_full ? 0 : 1) | (g->d_scratch ? 0 : 2);
}

vksift_Result vksift_ext_shardGroupReserve(vksift_ext_ShardGroup g, uint32_t max_na, uint32_t max_nb_total)
{
  if (!g || g->broken)
    return VKSIFT_INVALID_INPUT_ERROR;
  vksift_hip_set_device(g->device);
  const size_t shard = ((size_t)max_nb_total + g->world - 1) / g->world;
  if (shard_reserve(g, max_na, shard * g->world) != 0)
  {
    logError(LOG_TAG, "vksift_ext_shardGroupReserve() error: out of device memory");
    return VKSIFT_VULKAN_ERROR;
  }
  return VKSIFT_SUCCESS;
}

/* Failure discipline of a collective call: a rank that returns before ncclAllGather leaves every peer blocked inside it.
 *   - arguments that every rank passes identically (nb_shard, nb_total) are checked first: a violation fails on EVERY rank, nobody
 *     enters the collective;
 *   - rank-local failures (a NULL local pointer, no memory for the matcher's scratch) are remembered, the rank STILL takes part in
 *     the all-gather (sending its slot of the receive buffer in place of a missing shard), skips its own matching and returns the error;
 *   - the one failure that makes taking part impossible — no memory for the receive buffer — aborts the communicator
 *     (ncclCommAbort) and marks the group broken. vksift_ext_shardGroupReserve() on every rank, agreed over the host channel
 *     before the first matching, removes that case: a call within the reservation allocates nothing. */
vksift_Result vksift_ext_matchSharded(vksift_ext_ShardGroup g, const uint8_t *d_a_rows, uint32_t na, uint32_t a_index_base, const uint8_t *d_b_shard,
                                      uint32_t nb_shard, uint32_t nb_total, uint8_t *d_matches)
{
  if (!g || g->broken)
    return VKSIFT_INVALID_INPUT_ERROR;
  if (nb_shard == 0 || nb_total < 2 || (uint64_t)nb_shard * g->world < nb_total)
    return VKSIFT_INVALID_INPUT_ERROR; /* the same on every rank */
  vksift_hip_set_device(g->device);
  const bool local_args_ok = (na == 0 || (d_a_rows && d_matches)) && d_b_shard;
  const int missing = shard_reserve(g, na, (size_t)nb_shard * g->world);
  if (missing & 1)
  {
    logError(LOG_TAG, "vksift_ext_matchSharded() error: no device memory for the gathered reference set; aborting the communicator");
    if (g->comm && g_rccl.CommAbort)
      g_rccl.CommAbort(g->comm);
    g->comm = NULL;
    g->broken = true;
    return VKSIFT_VULKAN_ERROR;
  }
  const bool compute = local_args_ok && !(missing & 2);
  const uint8_t *send = d_b_shard ? d_b_shard : g->d_b_full + (size_t)g->rank * nb_shard * 128u;
  uint32_t *norm_a = g->d_scratch, *norm_b = norm_a + na, *rest = norm_b + (size_t)nb_shard * g->world;
  vksift_hip_range_push("Sharded matching");
  int e = vksift_hip_event_record(g->ev_t0, g->stream);
  /* 1. the exchange first, on its own stream (behind everything already queued on the group's stream) */
  if (e == 0)
    e = vksift_hip_event_record(g->ev_fork, g->stream);
  if (e == 0)
    e = vksift_hip_stream_wait_event(g->comm_stream, g->ev_fork);
  {
    /* entered even when the event calls above failed: the peers are (or will be) inside it */
    if (shard_all_gather(g, send, g->d_b_full, (size_t)nb_shard * 128u, g->comm_stream) != 0)
      e = -1;
  }
  if (e == 0)
    e = vksift_hip_event_record(g->ev_gathered, g->comm_stream);
  /* 2. meanwhile: the pre-pass of the local query rows */
  if (e == 0 && compute)
    e = vksift_hip_shifted_norms(d_a_rows, na, norm_a, g->stream);
  /* 3. B has landed: its norms, then every local row of A against ALL of B in index order */
  if (e == 0)
    e = vksift_hip_stream_wait_event(g->stream, g->ev_gathered);
  if (e == 0 && compute)
    e = vksift_hip_shifted_norms(g->d_b_full, nb_total, norm_b, g->stream);
  if (e == 0 && compute)
    e = vksift_hip_match_2nn_prenormed(d_a_rows, norm_a, na, a_index_base, g->d_b_full, norm_b, nb_total, rest,
                                       g->scratch_cap - na - (size_t)nb_shard * g->world, d_matches, g->stream);
  if (e == 0)
    e = vksift_hip_event_record(g->ev_t1, g->stream);
  vksift_hip_range_pop();
  if (e != 0)
  {
    logError(LOG_TAG, "vksift_ext_matchSharded() error: %s", e > 0 ? vksift_hip_error_string(e) : "collective failed");
    return VKSIFT_VULKAN_ERROR;
  }
  if (!compute)
  {
    logError(LOG_TAG, "vksift_ext_matchSharded() error: %s (the all-gather was entered, no records were produced on this rank)",
             local_args_ok ? "out of device memory" : "invalid input");
    return local_args_ok ? VKSIFT_VULKAN_ERROR : VKSIFT_INVALID_INPUT_ERROR;
  }
  g->timed = true;
  return VKSIFT_SUCCESS;
}

vksift_Result vksift_ext_shardGroupSynchronize(vksift_ext_ShardGroup g, float *last_match_ms)
{
  if (!g || g->broken)
    return VKSIFT_INVALID_INPUT_ERROR;
  vksift_hip_set_device(g->device);
  if (vksift_hip_stream_sync(g->stream) != 0)
    return VKSIFT_VULKAN_ERROR;
  if (last_match_ms)
    *last_match_ms = g->timed ? vksift_hip_event_elapsed_ms(g->ev_t0, g->ev_t1) : -1.f;
  return VKSIFT_SUCCESS;
}
