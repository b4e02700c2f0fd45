/*
 * vksift_ext.h — additive extensions to the vksift_* API for the MI355X build.
 *
 * Nothing here exists in the reference; none of it changes the layout or behaviour of the reference
 * API (include/vulkansift/vulkansift.h). The extensions expose what a 288 GB / 8 TB/s device makes
 * worthwhile: batched detection (the reference handles one image at a time per instance,
 * vulkansift.c:326-327), device-resident inputs/outputs for multi-GPU pipelines, and stage timings
 * taken with HIP events on the instance's own stream.
 */
#ifndef VKSIFT_EXT_H
#define VKSIFT_EXT_H

#include "vulkansift/vulkansift.h"

#ifdef __cplusplus
extern "C"
{
#endif

  /* Like vksift_createInstance, but reserves `batch_capacity` pyramids so that up to that many
   * same-sized images can be processed by one vksift_ext_detectFeaturesBatch call.
   * config->sift_buffer_count must be >= batch_capacity. */
  VKSIFT_EXPORT vksift_Result vksift_ext_createInstanceBatched(vksift_Instance *instance_ptr, const vksift_Config *config, uint32_t batch_capacity);

  /* Detect on `count` images of identical resolution; image i fills SIFT buffer first_gpu_buffer_id+i.
   * Same asynchronous contract and error behaviour as vksift_detectFeatures. */
  VKSIFT_EXPORT void vksift_ext_detectFeaturesBatch(vksift_Instance instance, const uint8_t *const *images, uint32_t count, uint32_t image_width,
                                                    uint32_t image_height, uint32_t first_gpu_buffer_id);
  /* Same, images already in device memory (contiguous, image i at d_images + i*width*height). */
  VKSIFT_EXPORT void vksift_ext_detectFeaturesBatchDevice(vksift_Instance instance, const uint8_t *d_images, uint32_t count, uint32_t image_width,
                                                          uint32_t image_height, uint32_t first_gpu_buffer_id);

  /* 2-NN matching of `count` buffer pairs (A[i], B[i]) in one launch sequence; count <= batch capacity.
   * Same asynchronous contract and error behaviour as vksift_matchFeatures; pair 0 is also what
   * vksift_getMatchesNumber / vksift_downloadMatches return. */
  VKSIFT_EXPORT void vksift_ext_matchFeaturesBatch(vksift_Instance instance, uint32_t count, const uint32_t *gpu_buffer_ids_A,
                                                   const uint32_t *gpu_buffer_ids_B);
  VKSIFT_EXPORT uint32_t vksift_ext_getMatchesNumberBatch(vksift_Instance instance, uint32_t pair);
  VKSIFT_EXPORT void vksift_ext_downloadMatchesBatch(vksift_Instance instance, uint32_t pair, vksift_Match_2NN *matches);

  /* ---- GPU-side match filtering ------------------------------------------------------------------
   * What callers of the reference do on the CPU after vksift_downloadMatches (src/examples/test_sift_match.cpp:90-107,
   * src/perf/perf_common.cpp:123-169): match A->B and, with cross_check, B->A; keep a match iff it is mutual and passes
   * Lowe's ratio test d1/d2 < ratio (in both directions when cross-checking). Only the survivors (16 B each, increasing
   * idx_a) leave the GPU. Asynchronous like vksift_matchFeatures; the forward 2-NN records stay available through
   * vksift_getMatchesNumber / vksift_downloadMatches (pair 0) and the ...Batch accessors. count <= batch capacity. */
  typedef struct
  {
    uint32_t idx_a, idx_b;
    float dist_a_b1, dist_a_b2;
  } vksift_ext_FilteredMatch;
  VKSIFT_EXPORT void vksift_ext_matchFeaturesFiltered(vksift_Instance instance, uint32_t count, const uint32_t *gpu_buffer_ids_A,
                                                      const uint32_t *gpu_buffer_ids_B, float ratio, bool cross_check);
  VKSIFT_EXPORT uint32_t vksift_ext_getFilteredMatchesNumber(vksift_Instance instance, uint32_t pair);
  VKSIFT_EXPORT void vksift_ext_downloadFilteredMatches(vksift_Instance instance, uint32_t pair, vksift_ext_FilteredMatch *matches);

  /* Deferred submission of vksift_detectFeatures (no counterpart in the reference, no change of its contract): consecutive plain
   * detect calls into consecutive SIFT buffers, with nothing asked in between, are staged and launched as ONE batched detection by
   * the first call that needs a result — any other entry point — or when 128 images (VKSIFT_DEFER_MAX) are staged, or 16 (VKSIFT_DEFER_CHUNK)
   * while the GPU has no detection to work on. The first detect
   * call after another entry point is launched at once unless the caller's previous run of detect calls held two or more, so
   * detect + read and the two-buffer ping-pong keep their latency. VKSIFT_DEFER=0 launches every call at once. Results are
   * identical either way. These counters say what the instance did: batches launched from staged images, and images in them. */
  VKSIFT_EXPORT void vksift_ext_getDeferredStats(vksift_Instance instance, uint64_t *nb_batches, uint64_t *nb_images);

  /* Stage timings (milliseconds, HIP events on the instance stream) of the last detect call.
   * Enabled with vksift_ext_setProfiling(instance, true); disabled by default. Blocking. */
  typedef struct
  {
    float upload_ms;      /* host->device image copy */
    float pyramid_ms;     /* input blit + all blur/DoG + down-sample launches */
    float extrema_ms;     /* detect + scan + emit */
    float orientation_ms;
    float descriptor_ms;
    float total_ms;
    uint32_t nb_blur_launches;
    uint64_t pyramid_algorithmic_bytes; /* SURVEY.md §8(d) definition, whole batch */
    float scan_ms;                      /* the streaming extrema scan of octave 0 alone (mask clear + the kernel that reads the S+3 planes) */
    uint64_t scan_algorithmic_bytes;    /* 4*(S+2) B per octave-0 pixel, whole batch */
    float pyramid_all_ms;               /* the scale-space construction of EVERY octave: first launch of octave 0 to the last blur launch of the
                                         * coarsest octave, on the stream they run on (pyramid_ms: octave 0 alone) */
    uint32_t nb_blur_launches_all;      /* launches inside that interval */
  } vksift_ext_DetectTimings;
  /* The struct grew twice (scan_ms, scan_algorithmic_bytes; pyramid_all_ms, nb_blur_launches_all) and may grow again at its end. The two getters without a size
   * argument therefore write only the first VKSIFT_EXT_DETECT_TIMINGS_V1_BYTES bytes — the struct of the first release, so a
   * client compiled against that header is never written past its storage; the ...Sized forms write min(out_bytes, sizeof)
   * bytes of the current struct (pass sizeof(vksift_ext_DetectTimings) of the header you compiled against). */
#define VKSIFT_EXT_DETECT_TIMINGS_V1_BYTES 40u
  VKSIFT_EXPORT void vksift_ext_setProfiling(vksift_Instance instance, bool enabled);
  VKSIFT_EXPORT void vksift_ext_getDetectTimings(vksift_Instance instance, vksift_ext_DetectTimings *out);
  VKSIFT_EXPORT void vksift_ext_getDetectTimingsSized(vksift_Instance instance, vksift_ext_DetectTimings *out, size_t out_bytes);
  /* Sums over all detect calls since profiling was enabled (or since the last reset); nb_blur_launches and
   * pyramid_algorithmic_bytes are summed too. Blocking. */
  VKSIFT_EXPORT void vksift_ext_getAccumulatedDetectTimings(vksift_Instance instance, vksift_ext_DetectTimings *sum, uint32_t *nb_calls, bool reset);
  VKSIFT_EXPORT void vksift_ext_getAccumulatedDetectTimingsSized(vksift_Instance instance, vksift_ext_DetectTimings *sum, size_t sum_bytes, uint32_t *nb_calls,
                                                                  bool reset);
  /* Clients compiled against THIS header get every field: the unsized names expand to the ...Sized forms with the size of the
   * struct they were compiled with. The exported symbols of the same names keep the 40-byte behaviour for binaries built against
   * the first release (which pass a 40-byte struct). */
#ifndef VKSIFT_BUILD
#define vksift_ext_getDetectTimings(instance, out) vksift_ext_getDetectTimingsSized((instance), (out), sizeof(vksift_ext_DetectTimings))
#define vksift_ext_getAccumulatedDetectTimings(instance, sum, nb_calls, reset) \
  vksift_ext_getAccumulatedDetectTimingsSized((instance), (sum), sizeof(vksift_ext_DetectTimings), (nb_calls), (reset))
#endif
  /* Where the scale-space was put: batch instances time a whole-batch blur launch on candidate memory ranges
   * when they allocate their scale-space buffers and keep the fastest. gbps[0 .. n): the rate of that launch (8 B per texel) on every
   * candidate in allocation order, chosen[0 .. 1]: the indices in use (chosen[1] = chosen[0] unless the instance holds two buffers: VKSIFT_PYR_PINGPONG=2). Returns n — 0 when the
   * buffers were allocated plainly (single-image instances, VKSIFT_PYR_PLACEMENT=0). */
  VKSIFT_EXPORT uint32_t vksift_ext_getScaleSpacePlacement(vksift_Instance instance, float gbps[8], uint32_t chosen[2]);

  /* Page-locked result buffers. vksift_downloadFeatures / vksift_ext_downloadMatchesBatch copy device -> pinned staging -> the caller's
   * (pageable) array: the second hop is a host memcpy at ~10 GB/s, 15 ms per 512 VGA frames' features — more than the bus takes. A
   * destination that is page-locked — registered here (hipHostRegister underneath), or any memory hipHostMalloc returned — receives the
   * records of a batched detection / matching by DMA straight from device memory: no staging, no host copy. Register once, reuse the
   * buffer. Both return VKSIFT_SUCCESS or VKSIFT_VULKAN_ERROR. For callers that fetch while the GPU is otherwise idle: with a detection
   * queued behind, every such transfer waits in the copy engine's ring (77 us each on MI355X against 7.5 us for the staged path, which
   * moves the whole detection in a few large pieces). */
  VKSIFT_EXPORT vksift_Result vksift_ext_pinHostMemory(void *ptr, size_t bytes);
  VKSIFT_EXPORT vksift_Result vksift_ext_unpinHostMemory(void *ptr);

  /* Time (ms) of the last matching pipeline (gather + 2-NN kernel), HIP events; needs profiling on. */
  VKSIFT_EXPORT float vksift_ext_getMatchTime(vksift_Instance instance);

  /* Copy the descriptors of a SIFT buffer, in download order, as dense 128-byte rows into caller
   * provided DEVICE memory (>= vksift_getFeaturesNumber()*128 bytes, 16-byte aligned). Blocking.
   * Returns the number of rows written. Used to feed the RCCL all-gather of the sharded matcher. */
  VKSIFT_EXPORT uint32_t vksift_ext_exportDescriptorsDevice(vksift_Instance instance, uint32_t gpu_buffer_id, uint8_t *d_descriptors);

  /* ---------------------------------------------------------------------------------------------------------------
   * Sharded 2-NN matching over the GPUs of one node. One process per GPU. The query rows of A are sharded
   * by the caller (any split); the reference set B is sharded in equal blocks of nb_shard = ceil(nb_total / world) rows
   * (rank r holds rows [r*nb_shard, ...); rows past nb_total are padding) and all-gathered ONCE inside the call: an RCCL
   * all-gather of uint8 rows over xGMI, issued first and overlapped with the norm pre-pass of the local A rows. Every
   * rank then scans all of B in index order, so the records are bit-identical to a single-GPU vksift_matchFeatures for
   * every world size. RCCL is loaded on first use (dlopen).
   *   rank 0: vksift_ext_shardGetUniqueId(id), then send the 128 bytes to the other ranks by any host channel
   *   all   : vksift_ext_shardGroupCreate(&group, device, world, rank, id)      (collective: ncclCommInitRank)
   *   all   : vksift_ext_matchSharded(...)                                       (collective, asynchronous on the group's stream)
   *   all   : vksift_ext_shardGroupSynchronize(group, &ms)
   * d_a_rows / d_b_shard / d_matches are DEVICE pointers (dense 128-byte rows; na records of 20 bytes = vksift_Match_2NN with
   * idx_a = a_index_base + row). vksift_ext_exportDescriptorsDevice() produces such rows from a SIFT buffer. */
#define VKSIFT_EXT_SHARD_ID_BYTES 128
  typedef struct vksift_ext_ShardGroup_T *vksift_ext_ShardGroup;
  VKSIFT_EXPORT vksift_Result vksift_ext_shardGetUniqueId(uint8_t id[VKSIFT_EXT_SHARD_ID_BYTES]);
  VKSIFT_EXPORT vksift_Result vksift_ext_shardGroupCreate(vksift_ext_ShardGroup *group_ptr, int gpu_device_index, uint32_t world, uint32_t rank,
                                                          const uint8_t id[VKSIFT_EXT_SHARD_ID_BYTES]);
  /* The same group over the APPLICATION's transport instead of RCCL (an MPI job, a host-staged exchange, a test harness): the
   * callback stands in for ncclAllGather and has its contract — every rank contributes bytes_per_rank bytes at d_send, rank r's
   * block lands at d_recv + r * bytes_per_rank on every rank (d_send may already BE this rank's slot of d_recv), ordered on
   * hip_stream (a hipStream_t): it must see everything queued on hip_stream before the call, and work queued on hip_stream after
   * it returns must see the gathered data (a blocking implementation synchronises hip_stream first). Returns 0 on success. Not
   * collective itself; RCCL is neither loaded nor needed. */
  typedef int (*vksift_ext_AllGatherFn)(void *user, const void *d_send, void *d_recv, size_t bytes_per_rank, uint32_t rank, uint32_t world,
                                        void *hip_stream);
  VKSIFT_EXPORT vksift_Result vksift_ext_shardGroupCreateWithTransport(vksift_ext_ShardGroup *group_ptr, int gpu_device_index, uint32_t world, uint32_t rank,
                                                                       vksift_ext_AllGatherFn all_gather, void *user);
  /* The block layout of the reference set that vksift_ext_matchSharded all-gathers: block_rows = ceil(n_total / world) (= nb_shard),
   * rank `rank` holds rows [first_row, first_row + nb_rows) of B (nb_rows <= block_rows; the rest of its block is padding). Pure
   * arithmetic (no GPU needed); callers that shard the query rows the same way use first_row as a_index_base. */
  /* What the group runs on: (world, rank) as created, and ncclCommCount / ncclCommUserRank of its RCCL communicator — 0 / 0 for a group
   * over the application's transport. A creation whose communicator disagrees with (world, rank) fails; a caller that prints
   * rccl_ranks proves how many ranks RCCL itself saw (bench.py does). */
  VKSIFT_EXPORT void vksift_ext_shardGroupInfo(vksift_ext_ShardGroup group, uint32_t *world, uint32_t *rank, uint32_t *rccl_ranks, uint32_t *rccl_rank);
  VKSIFT_EXPORT void vksift_ext_shardGroupLayout(uint32_t n_total, uint32_t world, uint32_t rank, uint32_t *block_rows, uint32_t *first_row,
                                                 uint32_t *nb_rows);
  VKSIFT_EXPORT void vksift_ext_shardGroupDestroy(vksift_ext_ShardGroup *group_ptr);
  /* Local, not collective: reserves the device scratch for matchings of up to max_na local query rows against up to max_nb_total
   * reference rows. A vksift_ext_matchSharded within the reservation allocates nothing, so it cannot fail for resources before its
   * collective (a rank that cannot allocate the receive buffer inside matchSharded has to abort the communicator: see there). */
  VKSIFT_EXPORT vksift_Result vksift_ext_shardGroupReserve(vksift_ext_ShardGroup group, uint32_t max_na, uint32_t max_nb_total);
  /* Collective. Error discipline: nb_shard / nb_total (identical on every rank) are validated before anything is queued; a rank with
   * a local failure (NULL local pointer, no scratch memory) still enters the all-gather and then returns its error, so its peers are
   * not left blocked; only a missing receive buffer aborts the communicator (group unusable afterwards). */
  VKSIFT_EXPORT vksift_Result vksift_ext_matchSharded(vksift_ext_ShardGroup group, const uint8_t *d_a_rows, uint32_t na, uint32_t a_index_base,
                                                      const uint8_t *d_b_shard, uint32_t nb_shard, uint32_t nb_total, uint8_t *d_matches);
  VKSIFT_EXPORT vksift_Result vksift_ext_shardGroupSynchronize(vksift_ext_ShardGroup group, float *last_match_ms);

  /* Deterministic synthetic test image: 128 + sum of Gaussian blobs + uniform noise,
   * splitmix64-seeded, clamped to [0,255]. nb_blobs == 0 picks the density used by the benchmarks. */
  VKSIFT_EXPORT void vksift_ext_genSyntheticImage(uint64_t seed, uint32_t width, uint32_t height, uint32_t nb_blobs, uint8_t *out);
  /* Further deterministic image families (the parity tests' second and third opinion on what an image looks like):
   * BLOBS = vksift_ext_genSyntheticImage with the benchmark density; EDGES = flat rotated rectangles and checker patches over a
   * ramp (long step edges, corners, junctions, shapes cut by the border); FRACTAL = 1/f value noise (texture at every scale). */
#define VKSIFT_EXT_SYNTH_BLOBS 0u
#define VKSIFT_EXT_SYNTH_EDGES 1u
#define VKSIFT_EXT_SYNTH_FRACTAL 2u
  VKSIFT_EXPORT void vksift_ext_genSyntheticImageFamily(uint64_t seed, uint32_t width, uint32_t height, uint32_t family, uint8_t *out);
  /* Deterministic SIFT-like descriptor rows: min(255, trunc(512*|g|/||g||)), g ~ N(0,1)^128. */
  VKSIFT_EXPORT void vksift_ext_genSyntheticDescriptors(uint64_t seed, uint32_t rows, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif /* VKSIFT_EXT_H */
