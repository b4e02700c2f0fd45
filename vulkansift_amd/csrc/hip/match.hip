// match.hip — brute-force 2-nearest-neighbour descriptor matcher on the gfx950 matrix cores.
//
// Replaces Get2NearestNeighbors.comp (dispatch sift_matcher.c:246-279): for every row of A the two
// closest rows of B under the L2 distance of the 128 uint8 descriptor bytes, scanned in index order
// with strict '<' (ties keep the earlier index) after an unconditional initialisation from b[0], b[1].
//
// Formulation (exact integers): with a' = a - 128, b' = b - 128 as int8 (byte XOR 0x80),
//     d2(a,b) = sum (a-b)^2 = |a'|^2 + |b'|^2 - 2 a'.b'        (every term < 2^22, int32 exact)
// and a'.b' for a 16x16 block of (A rows x B rows) is two v_mfma_i32_16x16x64_i8 (K = 128).
// The N_A x N_B distance matrix never leaves registers: each lane folds its 4 outputs per MFMA into a
// running top-2 per A row (fused epilogue), lanes are merged once at the end.
//
// Bit-exactness with the reference's float comparison (quirk Q8): the shader compares
// sqrt(float(d2)) values, and two different d2 can round to the same float — but only for d2 >= 2^22
// (below, q -> sqrtf(q) is injective, and q < 2^22 <= h implies sqrtf(q) < 2048 <= sqrtf(h)). The MFMA
// kernels therefore work on integers only: a lane sees its B columns in increasing index order and
// keeps the two smallest (d2, index) pairs, lanes/chunks are merged lexicographically — identical to
// the reference's strict-'<' scan whenever every INSERTED candidate has d2 < 2^22 (always true for
// real SIFT descriptors: |a - b| <= 1024). A row that inserts a larger candidate is flagged and
// recomputed by k_match_redo, a scalar kernel that replays the reference's float loop verbatim.
// Quirk Q7 (d(b0) == d(b1) makes index 1 the best) is an index-priority swap of columns 0 and 1 for
// that A row. Quirk Q6 (b[0], b[1] read unconditionally): callers pad B to two rows.
//
// MFMA operand layout used (16x16x64 i8): lane l supplies 16 consecutive K bytes (l>>4)*16.. of A row
// (l&15) / B row (l&15); since A and B use the same K slicing any K permutation cancels in the dot
// product. Accumulator: lane l holds column (l&15), rows (l>>4)*4 + r, r = 0..3.
//
// WHICH KERNEL SERVES WHICH PROBLEM (the one table; every path gives the reference's records bit for bit, tests/test_gpu_match_*.py):
//
//   entry                                   problem                                   kernels
//   ---------------------------------------------------------------------------------------------------------------------------------
//   vksift_hip_match_2nn_async, n pairs     every slot with N_B <= 32 768             k_match_pk<8,128> (packed keys, branch free) + k_match_redo
//   (vksift_ext_matchFeaturesBatch,         a slot with N_B > 32 768 (the host's      + k_match_mfma<1,4,64> (N_A <= 32 768) / <2,4,64> (above): the pruning
//    the frames of a batched detection)     bound max_nb says whether one can exist)    kernel on grids that walk the slots; not queued when max_nb <= 32 768
//   vksift_hip_match_2nn_async, 1 pair      N_A <= 1 536 (count on the device)        k_match_mfma_split (one launch, B split over the waves)
//   (vksift_matchFeatures)                  N_A >  1 536                              k_match_mfma<2,8,128> stream decomposition + k_match_merge
//                                                                                     (both are queued: the count decides on the device) + k_match_redo
//   vksift_hip_match_2nn_desc / _prenormed  N_A x N_B >= 64 M and N_B <= 32 768        k_match_pk on row blocks x strided pieces of B (+ k_match_merge)
//   (device pointers: the sharded matcher,  N_A <= 1 536 and N_B <= 4 096             k_match_mfma_split
//    vksift_ext_matchSharded)               N_B > 32 768 (BASELINE config 4)          k_match_scan32<3,8,256> -> k_match_fix -> k_match_redo_rows
//                                           anything else                             k_match_mfma<2,8,128> stream decomposition + k_match_merge
//
//   k_match_redo / k_match_redo_rows replay the reference's float loop for the rows a kernel flagged (a distance >= 2^22, a packed key that
//   failed its verification, a tie the cell scan cannot order): zero rows for real descriptors.
//   Switches (A/B and the bit-identity matrix only): VKSIFT_MATCH_PK=0 (no packed-key kernel: the pruning kernels carry batches),
//   VKSIFT_MATCH_SCAN=0 (no cell scan: the stream decomposition carries large reference sets), vksift_hip_tune(VKSIFT_TUNE_SCAN_FORM).
//   The record-moving launches around the matcher (gather, pack, filter) live in records.hip.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <type_traits>

#include "vksift_hip.h"

namespace
{

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int BT = 64;          // B rows staged in LDS per iteration
constexpr int B_STRIDE = 144;   // bytes per staged row (128 + 16 pad: ds_read_b128 of 16 rows hits 16 distinct bank quads)
constexpr uint32_t QMAX = 0xFFFFFFFFu;

// |d - 128|^2 per row: sum d^2 - 256 sum d + 128*128^2
__global__ void __launch_bounds__(256) k_shifted_norms(const uint32_t *__restrict__ desc, uint32_t n, uint32_t *__restrict__ norms)
{
  uint32_t row = blockIdx.x * 256 + threadIdx.x;
  if (row >= n)
    return;
  const uint4 *p = (const uint4 *)(desc + (size_t)row * 32);
  uint32_t s2 = 0, s1 = 0;
#pragma unroll
  for (int j = 0; j < 8; j++)
  {
    uint4 v = p[j];
    s2 = __builtin_amdgcn_udot4(v.x, v.x, s2, false);
    s2 = __builtin_amdgcn_udot4(v.y, v.y, s2, false);
    s2 = __builtin_amdgcn_udot4(v.z, v.z, s2, false);
    s2 = __builtin_amdgcn_udot4(v.w, v.w, s2, false);
    s1 = __builtin_amdgcn_udot4(v.x, 0x01010101u, s1, false);
    s1 = __builtin_amdgcn_udot4(v.y, 0x01010101u, s1, false);
    s1 = __builtin_amdgcn_udot4(v.z, 0x01010101u, s1, false);
    s1 = __builtin_amdgcn_udot4(v.w, 0x01010101u, s1, false);
  }
  norms[row] = s2 - 256u * s1 + 128u * 128u * 128u;
}

// Per-slot strides of a batched matching launch (blockIdx.y = slot); all zero for a single pair.
struct SlotStrides
{
  uint64_t desc_a, desc_b; // dwords
  uint64_t norm_a, norm_b; // u32
  uint64_t matches;        // dwords
  uint32_t n;              // u32 between the {N_A, N_B} pairs
  uint64_t redo;           // u32
  uint32_t slot_fast;      // k_match_mfma: blockIdx.x is the slot, blockIdx.y the row block
  uint32_t use_ids;        // descriptor / norm strides address the per-buffer cache: entry = SlotIds::a/b[slot] instead of the slot
  uint32_t pk_nb_max;      // slots with N_B <= this are matched by k_match_pk: the pruning kernels of the same launch sequence skip them
  uint32_t nslots_loop;    // k_match_mfma: != 0: slots of the call; the grid's slot dimension is smaller and every workgroup walks it
};

// SIFT buffer (= cache entry) matched by each slot of a batched launch
struct SlotIds
{
  uint32_t a[VKSIFT_HIP_MATCH_SLOTS], b[VKSIFT_HIP_MATCH_SLOTS];
};

// {N_A, N_B} of every slot, from the per-buffer row counts of the cache (the matcher kernels and the host read them per slot)
__global__ void k_slot_counts(const uint32_t *__restrict__ cache_n, uint32_t cache_n_stride, SlotIds ids, uint32_t nslots, uint32_t *__restrict__ n_out,
                              uint32_t n_slot_stride)
{
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= nslots)
    return;
  n_out[(size_t)slot * n_slot_stride + 0] = cache_n[(size_t)ids.a[slot] * cache_n_stride];
  n_out[(size_t)slot * n_slot_stride + 1] = cache_n[(size_t)ids.b[slot] * cache_n_stride];
}

// Stream decomposition of the large single-pair matcher (k_match_mfma with `stream` set, k_match_merge): the list of
// (row block, B tile) pairs, row block major, is cut into runs of `span` tiles, one per workgroup of a grid of G. A row block
// is then covered by at most floor(tiles / span) + 2 runs, which must not exceed VKSIFT_HIP_MATCH_CHUNKS partial lists.
__device__ __forceinline__ uint32_t stream_span(uint32_t nblocks, uint32_t tiles, uint32_t G)
{
  const uint32_t even = (nblocks * tiles + G - 1u) / G, floor_ = (tiles + (uint32_t)VKSIFT_HIP_MATCH_CHUNKS - 3u) / ((uint32_t)VKSIFT_HIP_MATCH_CHUNKS - 2u);
  return max(max(even, floor_), 1u);
}

// compute units of the current device (workgroup slots of the stream decomposition)
uint32_t device_cus()
{
  // one slot per device, written once with the same value by whichever thread comes first: instances on different GPUs may be driven
  // from different host threads of one process (tests/test_gpu_two_instances.py)
  static uint32_t cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess)
    return 256u;
  if (dev < 0 || dev >= 64 || cached[dev] == 0)
  {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    if (dev < 0 || dev >= 64)
      return (uint32_t)n;
    cached[dev] = (uint32_t)n;
  }
  return cached[dev];
}

// VKSIFT_MATCH_SCAN=0: single pairs with large reference sets take the stream-decomposed pruning kernel instead of the cell scan
bool match_use_scan()
{
  static int cached = -1;
  if (cached < 0)
  {
    const char *e = getenv("VKSIFT_MATCH_SCAN");
    cached = (e && atoi(e) == 0) ? 0 : 1;
  }
  return cached == 1;
}

// VKSIFT_MATCH_PK=0: batches of pairs go through the pruning kernels only (A/B switch, same results)
bool match_use_pk()
{
  static int cached = -1;
  if (cached < 0)
  {
    const char *e = getenv("VKSIFT_MATCH_PK");
    cached = (e && atoi(e) == 0) ? 0 : 1;
  }
  return cached == 1;
}

struct Top2
{
  uint32_t q1, k1, q2, k2; // squared distances and index keys of best / second
};

constexpr uint32_t Q_EXACT = 1u << 22;
constexpr uint32_t SYNC_TILES = 8;  // tiles between two exchanges of the row-wide pruning bound in the steady state (power of two)

// Accumulator-space form of the pruning test (see k_match_mfma): D > acc_threshold(an, eff) is implied by an + par - 2 D < eff.
// Real accumulators stay above -2^22 (|a'.b'| <= 2^21, bn/2 <= 2^20): ACC_PASS lets all of them through, ACC_DEAD none.
constexpr int ACC_PASS = -(1 << 29), ACC_DEAD = -(1 << 30);
__device__ __forceinline__ int acc_threshold(uint32_t an, uint32_t eff)
{
  return eff >= (1u << 30) ? ACC_PASS : ((int)an - (int)eff) >> 1;
}

// value of lane ((lane & 15) + n) % 16 of the same 16-lane row: one DPP move
template <int N>
__device__ __forceinline__ uint32_t row_ror_n(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + N, 0xf, 0xf, false);
}
// (m1, m2) <- the two smallest of {m1, m2} of this lane and of the lane N positions further in the row
template <int N>
__device__ __forceinline__ void merge_ror(uint32_t &m1, uint32_t &m2)
{
  const uint32_t r1 = row_ror_n<N>(m1), r2 = row_ror_n<N>(m2);
  const uint32_t hi = max(m1, r1);
  m1 = min(m1, r1);
  m2 = min(hi, min(m2, r2));
} // below this, integer order of d2 == order of sqrtf(float(d2))

// median of three = the second largest of {m1, m2, x} when m1 >= m2 (v_med3_u32)
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) { return max(min(a, b), min(max(a, b), c)); }
__device__ __forceinline__ uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) { return max(max(a, b), c); }

__device__ __forceinline__ bool lex_less(uint32_t qa, uint32_t ka, uint32_t qb, uint32_t kb) { return qa < qb || (qa == qb && ka < kb); }

// in-lane insertion; keys arrive in increasing order so a tie never displaces a holder. Precondition: q < t.q2.
__device__ __forceinline__ void insert_seq(Top2 &t, uint32_t q, uint32_t key)
{
  if (q < t.q1)
  {
    t.q2 = t.q1, t.k2 = t.k1;
    t.q1 = q, t.k1 = key;
  }
  else
    t.q2 = q, t.k2 = key;
}

__device__ __forceinline__ Top2 merge2(const Top2 &a, const Top2 &b)
{
  Top2 r;
  const bool a_first = lex_less(a.q1, a.k1, b.q1, b.k1);
  const Top2 &w = a_first ? a : b; // winner of the best slot
  const Top2 &l = a_first ? b : a;
  r.q1 = w.q1, r.k1 = w.k1;
  const bool w2 = lex_less(w.q2, w.k2, l.q1, l.k1);
  r.q2 = w2 ? w.q2 : l.q1;
  r.k2 = w2 ? w.k2 : l.k1;
  return r;
}

// AT = 16-row A tiles per wave. Block = 4 waves = 64*AT A rows; B streams through LDS.
// gridDim.z > 1: B is split into gridDim.z chunks of whole 64-row tiles; every chunk writes its partial top-2 per A row to
// `partial` ([row][chunk][4], then one flag word per (row, chunk): bit0 = Q7 swap, bit1 = redo) and k_match_merge combines them.
template <int AT, int NW = 4, int BTT = BT>
__global__ void __launch_bounds__(64 * NW) k_match_mfma(const uint32_t *__restrict__ desc_a, const uint32_t *__restrict__ norm_a, uint32_t na,
                                                    uint32_t a_index_base, const uint32_t *__restrict__ desc_b, const uint32_t *__restrict__ norm_b,
                                                    uint32_t nb, uint32_t *__restrict__ matches, uint32_t *__restrict__ redo,
                                                    const uint32_t *__restrict__ n_dev, uint32_t na_lo, uint32_t na_hi, SlotStrides ss,
                                                    uint32_t *__restrict__ partial, SlotIds ids, uint32_t stream)
{
  // stream != 0 (single pair, grid = (G, 1, 1)): the (row block, B tile) pairs, row block major, are cut into G equal runs of
  // `span` tiles (stream_span); a workgroup works through its run, which may end one row block and start the next. Every
  // (row block, workgroup) piece writes a partial list (slot = workgroup - first workgroup of the block), k_match_merge
  // combines them. All resident workgroups get the same number of tiles whatever N_A is; a grid of row blocks x chunks
  // leaves up to a third of the CUs' workgroup slots idle (measured: 588 workgroups on 768 slots at 50k x 50k).
  const uint32_t nchunks = stream ? (uint32_t)VKSIFT_HIP_MATCH_CHUNKS : gridDim.z;
  uint32_t chunk = blockIdx.z;
  // ss.slot_fast: grid = (slots, row blocks) — see the batched launch
  const uint32_t rb0 = ss.slot_fast ? blockIdx.y : blockIdx.x, rb_step = ss.slot_fast ? gridDim.y : gridDim.x;
  // ss.nslots_loop != 0 (the size regimes of a batch that the packed-key kernel does not take): the grid covers only SOME slot
  // positions; a workgroup walks the slots slot0, slot0 + step, ... and works on those whose device-side counts fall in ITS regime —
  // for a batch of small sets every workgroup reads a few counts and ends (a grid sized for the capacity put thousands of
  // idle workgroups in front of the next detection's launches)
  const uint32_t slot0 = ss.slot_fast ? blockIdx.x : blockIdx.y, slot_step = ss.slot_fast ? gridDim.x : gridDim.y;
  const uint32_t *const desc_a_0 = desc_a, *const desc_b_0 = desc_b, *const norm_a_0 = norm_a, *const norm_b_0 = norm_b, *const n_dev_0 = n_dev;
  uint32_t *const matches_0 = matches, *const redo_0 = redo;
  for (uint32_t slot = slot0; slot == slot0 || slot < ss.nslots_loop; slot += slot_step)
  {
  const uint32_t ea = ss.use_ids ? ids.a[slot] : slot, eb = ss.use_ids ? ids.b[slot] : slot;
  desc_a = desc_a_0 + (size_t)ea * ss.desc_a, desc_b = desc_b_0 + (size_t)eb * ss.desc_b;
  norm_a = norm_a_0 + (size_t)ea * ss.norm_a, norm_b = norm_b_0 + (size_t)eb * ss.norm_b;
  matches = matches_0 + (size_t)slot * ss.matches;
  redo = redo_0 + (size_t)slot * ss.redo;
  if (n_dev_0)
  {
    n_dev = n_dev_0 + (size_t)slot * ss.n;
    // asynchronous path: the row counts were produced on the device by k_gather_sections; this instantiation only
    // serves na in (na_lo, na_hi] (the host launches one kernel per regime, the others skip the slot here)
    na = n_dev[0];
    nb = n_dev[1] < 2u ? 2u : n_dev[1];
    if (na <= na_lo || na > na_hi || nb <= ss.pk_nb_max)
      continue;
  }
  // Operand roles are swapped with respect to the textbook A x B^T: the B descriptors are the MFMA's A operand and the query
  // rows its B operand, so the 16x16 result block is indexed [B column][A row] and lane (col = lane & 15, grp = lane >> 4)
  // holds, for ITS A row `col` of the tile, the four B columns grp*4 + j of the block. Consequences:
  //   * one top-2 state per (lane, row tile) instead of four: the candidates of an A row live in 4 lanes (grp 0..3), each
  //     seeing a quarter of the columns in increasing order — its own second best is already a fair bound
  //   * the row-wide bound is a 2-step butterfly over grp (2 values, lanes ^16 and ^32): cheap enough for every tile
  //   * the pruning test needs no per-column work: the accumulator starts at C = -(bn >> 1), read as one 16-byte LDS vector,
  //     and "can any of these four candidates beat eff?" is max(acc) > thr with thr = floor((an - eff) / 2) per lane
  //     (an + (bn & 1) - 2 acc < eff  =>  acc > thr; conservative by the parity bit, the exact d2 is formed behind it)
  // two staging buffers: tile k+1 is written while tile k is consumed -> one barrier per tile
  __shared__ __attribute__((aligned(16))) uint8_t s_b2[2][BTT * B_STRIDE];
  __shared__ __attribute__((aligned(16))) int s_nbh2[2][BTT]; // -(bn >> 1), ACC_DEAD for rows beyond B: the MFMA's C operand
  __shared__ __attribute__((aligned(16))) uint32_t s_nb2[2][BTT];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, grp = lane >> 4;
  const uint32_t tiles = (nb + BTT - 1) / BTT, tiles_per_chunk = (tiles + gridDim.z - 1) / gridDim.z;
  uint32_t tb = chunk * tiles_per_chunk * BTT;
  uint32_t te = min(nb, tb + tiles_per_chunk * BTT);
  const uint32_t nblocks = (na + 16u * NW * AT - 1u) / (16u * NW * AT);
  const uint32_t span = stream ? stream_span(nblocks, tiles, gridDim.x) : 0u;
  uint32_t pos = blockIdx.x * span;
  const uint32_t pos_end = min(pos + span, nblocks * tiles);

  // the grid may be smaller than the number of 64*AT-row blocks (bounded launch): loop over row blocks
  for (uint32_t rb = rb0;; rb += rb_step)
  {
    if (span)
    {
      if (pos >= pos_end)
        break;
      rb = pos / tiles;
      const uint32_t t_first = pos - rb * tiles, t_cnt = min(tiles - t_first, pos_end - pos);
      tb = t_first * BTT, te = min(nb, (t_first + t_cnt) * BTT);
      chunk = blockIdx.x - (rb * tiles) / span;
      pos += t_cnt;
    }
    else if (rb * (16u * NW * AT) >= na)
      break;
    const uint32_t row_base = (rb * NW + wave) * (16 * AT);

    // query fragments (XOR 0x80 -> int8) and the norm of the row this lane accumulates
    v4i afrag[AT][2];
    uint32_t an[AT];
#pragma unroll
    for (int t = 0; t < AT; t++)
    {
      uint32_t r = row_base + t * 16 + col;
      if (r >= na)
        r = na - 1;
      const uint4 *p = (const uint4 *)(desc_a + (size_t)r * 32);
      uint4 v0 = p[grp], v1 = p[4 + grp];
      afrag[t][0] = v4i{(int)(v0.x ^ 0x80808080u), (int)(v0.y ^ 0x80808080u), (int)(v0.z ^ 0x80808080u), (int)(v0.w ^ 0x80808080u)};
      afrag[t][1] = v4i{(int)(v1.x ^ 0x80808080u), (int)(v1.y ^ 0x80808080u), (int)(v1.z ^ 0x80808080u), (int)(v1.w ^ 0x80808080u)};
      an[t] = norm_a[r];
    }

    Top2 st[AT];
    uint32_t eff[AT]; // min(own second best, row-wide bound): what a candidate has to beat
    int thr[AT];      // eff in accumulator space
#pragma unroll
    for (int t = 0; t < AT; t++)
    {
      st[t] = Top2{QMAX, QMAX, QMAX, QMAX};
      eff[t] = QMAX;
      thr[t] = ACC_PASS;
    }
    // Row-wide pruning bound: the second smallest d2 of the whole row so far. A later candidate that does not beat it cannot
    // be in the final top-2 (two candidates with smaller-or-equal d2 and smaller index already sit in the four lanes' lists,
    // which all take part in the final merge), so it is dropped before the insertion logic — exact.
    uint32_t swap_bits = 0;  // bit t: d2(b0) == d2(b1) for that A row (quirk Q7)
    uint32_t risky_bits = 0; // bit t: a candidate >= 2^22 was inserted -> the row goes to k_match_redo

    // B tiles are prefetched one tile ahead into registers (2 x 16 B per thread) so that the global-load latency of
    // tile t+1 hides behind the MFMA + epilogue work of tile t.
    // B tiles travel global -> registers -> LDS one tile ahead of the MFMAs (a second register set for two tiles ahead costs
    // the 64-row form an occupancy step and gains the others nothing: measured)
    constexpr int NTH = 64 * NW, NLD = (BTT * 8 + NTH - 1) / NTH;
    struct TileRegs
    {
      uint4 d[NLD];
      uint32_t n;
    };
    TileRegs pf0;
    auto fetch_tile = [&](uint32_t t0, TileRegs &pf) {
#pragma unroll
      for (int q = 0; q < NLD; q++)
      {
        int i = threadIdx.x + q * NTH;
        int r = i >> 3, c = i & 7;
        pf.d[q] = make_uint4(0, 0, 0, 0);
        if (i < BTT * 8 && t0 + r < nb)
          pf.d[q] = ((const uint4 *)(desc_b + (size_t)(t0 + r) * 32))[c];
      }
      pf.n = (threadIdx.x < BTT && t0 + threadIdx.x < nb) ? norm_b[t0 + threadIdx.x] : 0u;
    };
    // stage a fetched tile (zero beyond nb) into buffer `bufi`, converting to int8
    auto stage_tile = [&](uint32_t t0, int bufi, const TileRegs &pf) {
#pragma unroll
      for (int q = 0; q < NLD; q++)
      {
        int i = threadIdx.x + q * NTH;
        int r = i >> 3, c = i & 7;
        uint4 v = pf.d[q];
        v.x ^= 0x80808080u, v.y ^= 0x80808080u, v.z ^= 0x80808080u, v.w ^= 0x80808080u;
        if (i < BTT * 8)
          *(uint4 *)(s_b2[bufi] + r * B_STRIDE + c * 16) = v;
      }
      if (threadIdx.x < BTT)
      {
        s_nb2[bufi][threadIdx.x] = pf.n;
        s_nbh2[bufi][threadIdx.x] = t0 + threadIdx.x < nb ? -(int)(pf.n >> 1) : ACC_DEAD; // te is nb or a tile boundary: < te inside a tile == < nb
      }
    };
    __syncthreads(); // the previous row block has finished reading both buffers
    fetch_tile(tb, pf0);
    stage_tile(tb, 0, pf0);
    __syncthreads();

    int buf = 0;
    for (uint32_t t0 = tb; t0 < te; t0 += BTT, buf ^= 1)
    {
      const bool more = t0 + BTT < te;
      if (more)
        fetch_tile(t0 + BTT, pf0); // in flight during this tile's MFMAs
      const uint8_t *s_b = s_b2[buf];
      const int *s_nbh = s_nbh2[buf];
      const uint32_t *s_nb = s_nb2[buf];

      // The MFMAs of a sub-block depend on nothing the tests produce (their C operand is a property of the B columns), so the
      // MFMAs of sub-block s+1 are issued before the tests of sub-block s: the matrix pipe works while the VALU tests.
      auto issue = [&](int sub, v4i *acc) {
        // the 16 B rows of the sub-block as the MFMA's A operand (lane: row col, 16 bytes of K from grp), C = -(bn >> 1) of
        // the lane's four columns
        const uint8_t *pb = s_b + (sub * 16 + col) * B_STRIDE + grp * 16;
        const v4i b0 = *(const v4i *)pb;
        const v4i b1 = *(const v4i *)(pb + 64);
        const v4i cinit = *(const v4i *)(s_nbh + sub * 16 + grp * 4);
#pragma unroll
        for (int t = 0; t < AT; t++)
        {
          acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b0, afrag[t][0], cinit, 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b1, afrag[t][1], acc[t], 0, 0, 0);
        }
      };
      v4i accp[2][AT];
      issue(0, accp[0]);
#pragma unroll
      for (int sub = 0; sub < BTT / 16; sub++)
      {
        if (sub + 1 < BTT / 16)
          issue(sub + 1, accp[(sub + 1) & 1]);
        const v4i *acc = accp[sub & 1];
        const bool first = (sub == 0 && t0 == 0);
#pragma unroll
        for (int t = 0; t < AT; t++)
        {
          const int m = max(max(acc[t][0], acc[t][1]), max(acc[t][2], acc[t][3]));
          if (!first && !(m > thr[t]))
            continue;
          const uint32_t bc0 = t0 + sub * 16 + grp * 4; // first of this lane's four B columns
          const v4i bn4 = *(const v4i *)(s_nb + sub * 16 + grp * 4); // parity bits of the four column norms
          if (first)
          {
            // quirk Q7: d2(b0) == d2(b1) (both in the grp-0 lane of the row): index 1 becomes the best
            const uint32_t q0 = an[t] + ((uint32_t)bn4[0] & 1u) - 2u * (uint32_t)acc[t][0], q1 = an[t] + ((uint32_t)bn4[1] & 1u) - 2u * (uint32_t)acc[t][1];
            if (grp == 0 && q0 == q1)
              swap_bits |= 1u << t;
            if (grp == 0 && (q0 >= Q_EXACT || q1 >= Q_EXACT))
              risky_bits |= 1u << t; // the tie test itself needs the float comparison
          }
#pragma unroll
          for (int j = 0; j < 4; j++)
          {
            // columns beyond B carry ACC_DEAD accumulators: they only get here in the first block (B is padded to >= 2 rows,
            // so columns 0 and 1 are always real)
            if (!(acc[t][j] > thr[t]) || bc0 + j >= nb)
              continue;
            // (with a Q7 tie the two columns keep their arrival keys 0, 1 — equal d2, so "key 0 first" already is the order
            // index 1, index 0 once the final k ^ 1 of the swapped rows is applied)
            const uint32_t q = an[t] + ((uint32_t)bn4[j] & 1u) - 2u * (uint32_t)acc[t][j];
            if (q < eff[t])
            {
              if (q >= Q_EXACT)
                risky_bits |= 1u << t;
              insert_seq(st[t], q, bc0 + (uint32_t)j);
              eff[t] = min(eff[t], st[t].q2);
              thr[t] = acc_threshold(an[t], eff[t]);
            }
          }
        }
      }
      // ---- tighten the row-wide bound: the four lanes of a row exchange their two smallest d2 (after every tile at the
      // start, where the bound moves fast, then every SYNC_TILES tiles)
      const uint32_t tile_no = (t0 - tb) / BTT + 1u;
      if (tile_no <= SYNC_TILES || (tile_no & (SYNC_TILES - 1u)) == 0u)
      {
#pragma unroll
        for (int t = 0; t < AT; t++)
        {
          uint32_t m1 = st[t].q1, m2 = st[t].q2;
#pragma unroll
          for (int x = 16; x <= 32; x <<= 1)
          {
            const uint32_t r1 = __shfl_xor(m1, x, 64), r2 = __shfl_xor(m2, x, 64);
            const uint32_t hi = max(m1, r1);
            m1 = min(m1, r1);
            m2 = min(hi, min(m2, r2));
          }
          eff[t] = min(eff[t], m2);
          thr[t] = acc_threshold(an[t], eff[t]);
        }
      }
      if (more)
        stage_tile(t0 + BTT, buf ^ 1, pf0); // nobody reads that buffer any more (barrier at the end of the previous tile)
      __syncthreads();
    }

    // merge the 4 lanes that share an A row (butterfly over grp), then the grp-0 lane writes
    risky_bits |= __shfl_xor(risky_bits, 16, 64);
    risky_bits |= __shfl_xor(risky_bits, 32, 64);
    swap_bits |= __shfl_xor(swap_bits, 16, 64);
    swap_bits |= __shfl_xor(swap_bits, 32, 64);
#pragma unroll
    for (int t = 0; t < AT; t++)
    {
      Top2 s = st[t];
#pragma unroll
      for (int x = 16; x <= 32; x <<= 1)
      {
        Top2 o;
        o.q1 = __shfl_xor(s.q1, x, 64), o.k1 = __shfl_xor(s.k1, x, 64);
        o.q2 = __shfl_xor(s.q2, x, 64), o.k2 = __shfl_xor(s.k2, x, 64);
        s = merge2(s, o);
      }
      const uint32_t r = row_base + t * 16 + col;
      const uint32_t sw = (swap_bits >> t) & 1u, rk = (risky_bits >> t) & 1u;
      if (grp == 0 && r < na)
      {
        if (nchunks > 1)
        {
          uint32_t *pp = partial + ((size_t)r * nchunks + chunk) * 4;
          pp[0] = s.q1, pp[1] = s.k1, pp[2] = s.q2, pp[3] = s.k2;
          partial[(size_t)na * nchunks * 4 + (size_t)r * nchunks + chunk] = sw | (rk << 1);
        }
        else
        {
          uint32_t *m = matches + (size_t)r * 5;
          m[0] = a_index_base + r;
          m[1] = (sw && s.k1 < 2) ? (s.k1 ^ 1u) : s.k1;
          m[2] = (sw && s.k2 < 2) ? (s.k2 ^ 1u) : s.k2;
          m[3] = __float_as_uint(sqrtf((float)s.q1));
          m[4] = __float_as_uint(sqrtf((float)s.q2));
          redo[r] = rk;
        }
      }
    }
  } // row-block loop
  } // slot loop
}

// ---------------------------------------------------------------------------------------------------------------------------
// shared by the v_mfma_i32_32x32x32_i8 kernels below (k_match_scan32, k_match_pk)
typedef int v16i __attribute__((ext_vector_type(16)));

// position of 16-byte chunk c of staged row r inside the row (LDS swizzle): B tiles sit in LDS without padding, 16-byte chunks
// XOR-swizzled by ((row >> 1) & 7), so the four 16-lane groups of a ds_read_b128 hit 16 distinct bank quads (the 144-byte pitch
// of k_match_mfma: 33 % conflict cycles, SQ_LDS_BANK_CONFLICT)
__device__ __forceinline__ int swz(int r, int c) { return c ^ ((r >> 1) & 7); }

// ---------------------------------------------------------------------------------------------------------------------------
// k_match_scan32 + k_match_fix + k_match_redo_rows: the matcher for LARGE reference sets (beyond VKSIFT_HIP_MATCH_PK_NB rows).
//
// Every pruning kernel above pays for an insertion with a wave-wide detour, and a row sees ~2 ln N of them whatever the bound
// (k_match_mfma: 7 of its 10 issued instructions per MFMA are that detour; k_match32: 9 of 12). This scan has no detour and no
// index arithmetic at all. A lane (one query row per tile, 16 accumulators per 32-column sub-block = one CELL) only keeps, branch
// free, its two best cells: the largest accumulator of the sub-block (8 v_max3) folded into (M1, cell1, M2, cell2) by compares and
// selects — 20 VALU per 4 MFMAs (128 matrix-pipe cycles). Which columns of those cells are the two nearest neighbours, and their
// exact distances, is settled afterwards by k_match_fix from the descriptor bytes: 4 cells x 16 columns per row at most.
// Why that is exact. acc = a'.b' - (|b'|^2 >> 1), d2 = |a'|^2 + (|b'|^2 & 1) - 2 acc. Let T be the second largest cell maximum of a
// row (over both lanes, all pieces). Two different columns reach acc >= T, so the second smallest d2 is <= |a'|^2 + 1 - 2 T, and a
// column that beats or ties it has acc >= T: it lies in a cell whose maximum is >= T. A lane's cells arrive in column order and a
// later cell only replaces an earlier one when strictly larger, so the recorded cells are the earliest ones among equals. A lane
// keeps its THREE best cells and the VALUE of the fourth: a dropped cell is at most the lane's fourth maximum, which is at most T; it
// can only matter (on the parity bit) when it EQUALS T, i.e. when the lane's second, third and fourth all equal T. Those rows — and rows whose second
// distance reaches 2^22 (quirk Q8: float sqrt order) — go to the exact replay (k_match_redo_rows, one workgroup per row).
constexpr uint32_t CELL_NONE = 0x7FFFFFFFu;
constexpr int CELL_MIN = -2147483647 - 1;
struct Cell3
{
  int m1, m2, m3;
  uint32_t i1, i2, i3;
  int m4; // the fourth largest cell maximum, value only: "did the lane drop a cell that reaches the row's threshold?"
};
__device__ __forceinline__ int smed3(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }
// fold cell (m, idx) into the three best of a lane; cells arrive in column order, a later cell replaces an earlier one only when
// strictly larger (14 VALU: 3 compares, 5 selects, 2 min, 3 max, med3)
__device__ __forceinline__ void cell_update(Cell3 &s, int m, uint32_t idx)
{
  const bool c1 = m > s.m1, c2 = m > s.m2, c3 = m > s.m3;
  s.i3 = c2 ? s.i2 : (c3 ? idx : s.i3);
  s.i2 = c1 ? s.i1 : (c2 ? idx : s.i2);
  s.i1 = c1 ? idx : s.i1;
  s.m4 = max(s.m4, min(s.m3, m));
  s.m3 = max(s.m3, min(s.m2, m));
  s.m2 = smed3(s.m1, s.m2, m);
  s.m1 = max(s.m1, m);
}

// Partial lists: 16 words per (row, piece): per lane h two uint4 [m1 i1 m2 i2][m3 i3 m4 -]. Work decomposition, staging and LDS layout of k_match32.
template <int AT, int NW, int BTT, int PIPE = 1>
__global__ void __launch_bounds__(64 * NW) k_match_scan32(const uint32_t *__restrict__ desc_a, uint32_t na, const uint32_t *__restrict__ desc_b,
                                                      const uint32_t *__restrict__ norm_b, uint32_t nb, uint32_t *__restrict__ partial,
                                                      uint32_t *__restrict__ redo_list)
{
  constexpr uint32_t ROWS = 32u * AT * NW;
  if (blockIdx.x == 0 && threadIdx.x == 0)
    redo_list[0] = 0u; // the replay list k_match_fix appends to (a launch of its own until round 5: 5 us of a 370 us call)
  constexpr uint32_t nchunks = VKSIFT_HIP_MATCH_CHUNKS;
  __shared__ __attribute__((aligned(16))) uint8_t s_b2[2][BTT * 128];
  __shared__ __attribute__((aligned(16))) int s_nbh2[2][BTT]; // -(bn >> 1), ACC_DEAD for rows beyond B: the MFMA's C operand

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const uint32_t tiles = (nb + BTT - 1) / BTT;
  const uint32_t nblocks = (na + ROWS - 1u) / ROWS;
  const uint32_t span = stream_span(nblocks, tiles, gridDim.x);
  uint32_t pos = blockIdx.x * span;
  const uint32_t pos_end = min(pos + span, nblocks * tiles);

  constexpr int NTH = 64 * NW, NLD = (BTT * 8 + NTH - 1) / NTH;
  struct TileRegs
  {
    uint4 d[NLD];
    uint32_t n;
  };
  auto fetch_tile = [&](uint32_t t0, TileRegs &pf) {
#pragma unroll
    for (int q = 0; q < NLD; q++)
    {
      const int i = threadIdx.x + q * NTH;
      const int r = i >> 3, c = i & 7;
      pf.d[q] = make_uint4(0, 0, 0, 0);
      if (i < BTT * 8 && t0 + r < nb)
        pf.d[q] = ((const uint4 *)(desc_b + (size_t)(t0 + r) * 32))[c];
    }
    pf.n = (threadIdx.x < BTT && t0 + threadIdx.x < nb) ? norm_b[t0 + threadIdx.x] : 0u;
  };
  auto stage_tile = [&](uint32_t t0, int bufi, const TileRegs &pf) {
#pragma unroll
    for (int q = 0; q < NLD; q++)
    {
      const int i = threadIdx.x + q * NTH;
      const int r = i >> 3, c = i & 7;
      uint4 v = pf.d[q];
      v.x ^= 0x80808080u, v.y ^= 0x80808080u, v.z ^= 0x80808080u, v.w ^= 0x80808080u;
      if (i < BTT * 8)
        *(uint4 *)(s_b2[bufi] + r * 128 + swz(r, c) * 16) = v;
    }
    if ((int)threadIdx.x < BTT)
      s_nbh2[bufi][threadIdx.x] = t0 + threadIdx.x < nb ? -(int)(pf.n >> 1) : ACC_DEAD;
  };

  while (pos < pos_end)
  {
    const uint32_t rb = pos / tiles;
    const uint32_t t_first = pos - rb * tiles, t_cnt = min(tiles - t_first, pos_end - pos);
    const uint32_t tb = t_first * BTT, te = min(nb, (t_first + t_cnt) * BTT);
    const uint32_t chunk = blockIdx.x - (rb * tiles) / span;
    pos += t_cnt;
    const uint32_t row_base = (rb * NW + wave) * (32 * AT);

    v4i afrag[AT][4];
#pragma unroll
    for (int t = 0; t < AT; t++)
    {
      uint32_t r = row_base + t * 32 + j;
      if (r >= na)
        r = na - 1;
      const uint4 *p = (const uint4 *)(desc_a + (size_t)r * 32);
#pragma unroll
      for (int s = 0; s < 4; s++)
      {
        const uint4 v = p[2 * s + h];
        afrag[t][s] = v4i{(int)(v.x ^ 0x80808080u), (int)(v.y ^ 0x80808080u), (int)(v.z ^ 0x80808080u), (int)(v.w ^ 0x80808080u)};
      }
    }
    Cell3 st[AT];
#pragma unroll
    for (int t = 0; t < AT; t++)
      st[t] = Cell3{CELL_MIN, CELL_MIN, CELL_MIN, CELL_NONE, CELL_NONE, CELL_NONE, CELL_MIN};

    TileRegs pf0;
    __syncthreads(); // the previous piece has finished reading both buffers
    fetch_tile(tb, pf0);
    stage_tile(tb, 0, pf0);
    __syncthreads();
    int buf = 0;
    for (uint32_t t0 = tb; t0 < te; t0 += BTT, buf ^= 1)
    {
      const bool more = t0 + BTT < te;
      if (more)
        fetch_tile(t0 + BTT, pf0);
      const uint8_t *s_b = s_b2[buf];
      const int *s_nbh = s_nbh2[buf];
      auto issue = [&](int sub, v16i *acc) {
        const uint8_t *prow = s_b + (sub * 32 + j) * 128;
        v4i bf[4];
#pragma unroll
        for (int s = 0; s < 4; s++)
          bf[s] = *(const v4i *)(prow + swz(j, 2 * s + h) * 16);
        v16i cinit;
#pragma unroll
        for (int b = 0; b < 4; b++)
        {
          const v4i c4 = *(const v4i *)(s_nbh + sub * 32 + 8 * b + 4 * h);
          cinit[4 * b + 0] = c4[0], cinit[4 * b + 1] = c4[1], cinit[4 * b + 2] = c4[2], cinit[4 * b + 3] = c4[3];
        }
#pragma unroll
        for (int t = 0; t < AT; t++)
          acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[0], afrag[t][0], cinit, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < 4; s++)
#pragma unroll
          for (int t = 0; t < AT; t++)
            acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[s], afrag[t][s], acc[t], 0, 0, 0);
      };
      v16i accp[2][AT];
      issue(0, accp[0]);
      const uint32_t cell0 = t0 / 32u;
#pragma unroll
      for (int sub = 0; sub < BTT / 32; sub++)
      {
        if (sub + 1 < BTT / 32)
          issue(sub + 1, accp[(sub + 1) & 1]); // the matrix pipe works on the next sub-block while the VALU folds this one
        const v16i *acc = accp[sub & 1];
#pragma unroll
        for (int t = 0; t < AT; t++)
        {
          // the largest of the 16 accumulators: 8 three-operand maxima
          int m = max(max(acc[t][0], acc[t][1]), acc[t][2]);
#pragma unroll
          for (int i = 3; i + 1 < 16; i += 2)
            m = max(max(m, acc[t][i]), acc[t][i + 1]);
          m = max(m, acc[t][15]);
          cell_update(st[t], m, cell0 + (uint32_t)sub);
        }
        if (PIPE)
        {
          // Pin the software pipeline the source spells out: the 4 AT MFMAs of sub-block sub + 1 spread over the fold of sub-block sub.
          // Left alone the compiler sinks all four folds of a tile behind the staging of the next tile and issues the tile's 32 MFMAs in
          // one run: ~180 VALU instructions with nothing of this wave in the matrix pipe. The empty asm ties the fold's result to this
          // point of the program (nothing sinks past it, and with the memory clobber no LDS read of a later sub-block rises above it);
          // the group barriers order the instructions between two such points: one MFMA, its share of the fold's VALU, the next MFMA, ...
#pragma unroll
          for (int t = 0; t < AT; t++)
            asm volatile("" : "+v"(st[t].m1), "+v"(st[t].m2), "+v"(st[t].m3), "+v"(st[t].m4), "+v"(st[t].i1), "+v"(st[t].i2), "+v"(st[t].i3)::"memory");
#pragma unroll
          for (int g = 0; g < 4 * AT; g++)
          {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                 // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, (22 * AT + 4 * AT - 1) / (4 * AT), 0); // its share of the fold's VALU
          }
        }
      }
      if (more)
        stage_tile(t0 + BTT, buf ^ 1, pf0);
      __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < AT; t++)
    {
      const uint32_t r = row_base + t * 32 + j;
      if (r < na)
      {
        uint4 *pp = (uint4 *)(partial + (((size_t)r * nchunks + chunk) * 2 + h) * 8);
        pp[0] = make_uint4((uint32_t)st[t].m1, st[t].i1, (uint32_t)st[t].m2, st[t].i2);
        pp[1] = make_uint4((uint32_t)st[t].m3, st[t].i3, (uint32_t)st[t].m4, 0u);
      }
    }
  }
}

// The exact finish of k_match_scan32: per row, merge the pieces' cell lists of each lane (in column order, same rule), take the
// cells at or above the row's second largest cell maximum T, recompute the exact d2 of their columns from the descriptor bytes and
// keep the two smallest (d2, column) pairs — the reference's strict '<' scan order. A lane keeps THREE cells and the value of its
// fourth best: a cell it dropped can only matter when it equals T, and then the lane's second, third and fourth maxima all equal T
// (~1e-5 of the rows of random descriptors; a two-way tie at T, ~1 % of the rows, is simply evaluated). That and a second distance of
// 2^22 or more (quirk Q8) send the row to the exact replay. 32 rows per workgroup: 64 threads merge, then each wave finishes 8 rows with one column per lane (4 cells x 16 columns
// per round; a second round when more than four cells reach T).
__global__ void __launch_bounds__(256) k_match_fix(const uint32_t *__restrict__ partial, const uint32_t *__restrict__ desc_a, const uint32_t *__restrict__ norm_a,
                                                   uint32_t na, uint32_t a_index_base, const uint32_t *__restrict__ desc_b, const uint32_t *__restrict__ norm_b,
                                                   uint32_t nb, uint32_t stream_grid, uint32_t tile_rows, uint32_t block_rows, uint32_t *__restrict__ matches,
                                                   uint32_t *__restrict__ redo_list)
{
  __shared__ int s_m[32][6];      // cell maxima of a row: lane 0's three, lane 1's three
  __shared__ uint32_t s_i[32][6]; // their cell indices
  __shared__ int s_m4[32][2];     // the lanes' fourth largest maxima
  const uint32_t row0 = blockIdx.x * 32u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 64)
  {
    const uint32_t r = row0 + (threadIdx.x >> 1), hh = threadIdx.x & 1u;
    Cell3 s{CELL_MIN, CELL_MIN, CELL_MIN, CELL_NONE, CELL_NONE, CELL_NONE, CELL_MIN};
    int v4[4] = {CELL_MIN, CELL_MIN, CELL_MIN, CELL_MIN}; // the four largest cell maxima of the lane over all pieces (values only)
    auto val_insert = [&](int v) {
#pragma unroll
      for (int q = 0; q < 4; q++)
      {
        const int hi = max(v4[q], v), lo = min(v4[q], v);
        v4[q] = hi, v = lo;
      }
    };
    if (r < na)
    {
      const uint32_t tiles = (nb + tile_rows - 1u) / tile_rows;
      const uint32_t span = stream_span((na + block_rows - 1u) / block_rows, tiles, stream_grid);
      const uint32_t rb = r / block_rows;
      const uint32_t n = ((rb + 1u) * tiles - 1u) / span - (rb * tiles) / span + 1u;
      const uint4 *pp = (const uint4 *)(partial + (size_t)r * VKSIFT_HIP_MATCH_CHUNKS * 16u);
      for (uint32_t c = 0; c < n; c++)
      {
        const uint4 v = pp[(c * 2u + hh) * 2u], w = pp[(c * 2u + hh) * 2u + 1u];
        if ((int)v.x != CELL_MIN)
          cell_update(s, (int)v.x, v.y);
        if ((int)v.z != CELL_MIN)
          cell_update(s, (int)v.z, v.w);
        if ((int)w.x != CELL_MIN)
          cell_update(s, (int)w.x, w.y);
        val_insert((int)v.x), val_insert((int)v.z), val_insert((int)w.x), val_insert((int)w.z);
      }
    }
    const int lr = threadIdx.x >> 1;
    s_m4[lr][hh] = v4[3];
    s_m[lr][3 * hh + 0] = s.m1, s_m[lr][3 * hh + 1] = s.m2, s_m[lr][3 * hh + 2] = s.m3;
    s_i[lr][3 * hh + 0] = s.i1, s_i[lr][3 * hh + 1] = s.i2, s_i[lr][3 * hh + 2] = s.i3;
  }
  __syncthreads();
  for (int k = 0; k < 8; k++)
  {
    const int lr = wave * 8 + k;
    const uint32_t r = row0 + (uint32_t)lr;
    if (r >= na)
      break;
    // T = second largest of the six maxima (each lane's list is sorted: the two largest are among m1, m2 of both lanes)
    const int a1 = s_m[lr][0], a2 = s_m[lr][1], b1 = s_m[lr][3], b2 = s_m[lr][4];
    const int T = max(min(a1, b1), max(a2, b2));
    // a cell a lane did not record reaches T (then the lane's second, third and fourth maxima all equal T)
    const bool tie = (s_m4[lr][0] != CELL_MIN && s_m4[lr][0] >= T) || (s_m4[lr][1] != CELL_MIN && s_m4[lr][1] >= T);
    // the cells at or above T, in slot order (uniform): at most six
    int on_slot[6];
    int n_on = 0;
#pragma unroll
    for (int c = 0; c < 6; c++)
      if (s_m[lr][c] != CELL_MIN && s_m[lr][c] >= T && s_i[lr][c] != CELL_NONE)
        on_slot[n_on++] = c;
    Top2 s{QMAX, QMAX, QMAX, QMAX};
    uint32_t sw = 0;
    const int cq = lane >> 4, i = lane & 15;
    for (int round = 0; round * 4 < n_on; round++)
    {
      const int which = round * 4 + cq;
      const int slot = which < n_on ? on_slot[which < 6 ? which : 5] : -1;
      const int hh = slot >= 3 ? 1 : 0;
      const uint32_t col = slot >= 0 ? s_i[lr][slot] * 32u + 8u * (uint32_t)(i >> 2) + 4u * (uint32_t)hh + (uint32_t)(i & 3) : 0u;
      const bool on = slot >= 0 && col < nb;
      uint32_t q = QMAX;
      {
        const uint4 *pa = (const uint4 *)(desc_a + (size_t)r * 32);
        const uint4 *pb = (const uint4 *)(desc_b + (size_t)(on ? col : 0u) * 32);
        int dot = 0;
#pragma unroll
        for (int c = 0; c < 8; c++)
        {
          const uint4 va = pa[c], vb = pb[c];
          dot = __builtin_amdgcn_sdot4((int)(va.x ^ 0x80808080u), (int)(vb.x ^ 0x80808080u), dot, false);
          dot = __builtin_amdgcn_sdot4((int)(va.y ^ 0x80808080u), (int)(vb.y ^ 0x80808080u), dot, false);
          dot = __builtin_amdgcn_sdot4((int)(va.z ^ 0x80808080u), (int)(vb.z ^ 0x80808080u), dot, false);
          dot = __builtin_amdgcn_sdot4((int)(va.w ^ 0x80808080u), (int)(vb.w ^ 0x80808080u), dot, false);
        }
        if (on)
          q = norm_a[r] + norm_b[col] - 2u * (uint32_t)dot;
      }
      // quirk Q7: d2(b0) == d2(b1) (columns 0 and 1 share a cell: evaluated together or not at all)
      const unsigned long long m0 = __ballot(on && col == 0u), m1 = __ballot(on && col == 1u);
      if (m0 != 0ull && m1 != 0ull)
      {
        const uint32_t q0 = __shfl(q, __ffsll((long long)m0) - 1, 64), q1 = __shfl(q, __ffsll((long long)m1) - 1, 64);
        sw = q0 == q1 ? 1u : 0u;
      }
      Top2 t2{q, on ? col : QMAX, QMAX, QMAX};
#pragma unroll
      for (int x = 1; x < 64; x <<= 1)
      {
        Top2 o;
        o.q1 = __shfl_xor(t2.q1, x, 64), o.k1 = __shfl_xor(t2.k1, x, 64);
        o.q2 = __shfl_xor(t2.q2, x, 64), o.k2 = __shfl_xor(t2.k2, x, 64);
        t2 = merge2(t2, o);
      }
      s = merge2(s, t2);
    }
    if (lane == 0)
    {
      uint32_t *m = matches + (size_t)r * 5;
      m[0] = a_index_base + r;
      m[1] = (sw && s.k1 < 2) ? (s.k1 ^ 1u) : s.k1;
      m[2] = (sw && s.k2 < 2) ? (s.k2 ^ 1u) : s.k2;
      m[3] = __float_as_uint(sqrtf((float)s.q1));
      m[4] = __float_as_uint(sqrtf((float)s.q2));
      if (tie || s.q2 >= Q_EXACT || s.k2 == QMAX)
        redo_list[1u + atomicAdd(redo_list, 1u)] = r;
    }
  }
}

// Exact replay of Get2NearestNeighbors.comp:43-103 for the rows listed by k_match_fix: one workgroup per row, a thread scans the
// columns t, t + 256, ... in increasing order with the reference's float loop, the 256 partial results are merged by
// (distance, column) — the order a strict '<' scan produces; quirk Q7 relabels columns 0 and 1 when their float distances are equal.
__global__ void __launch_bounds__(256) k_match_redo_rows(const uint32_t *__restrict__ desc_a, uint32_t a_index_base, const uint32_t *__restrict__ desc_b, uint32_t nb,
                                                         uint32_t *__restrict__ matches, const uint32_t *__restrict__ redo_list)
{
  __shared__ float s_d[256][2];
  __shared__ uint32_t s_i[256][2];
  __shared__ float s_d01[2];
  const uint32_t n = redo_list[0];
  for (uint32_t k = blockIdx.x; k < n; k += gridDim.x)
  {
    const uint32_t row = redo_list[1u + k];
    uint32_t a[32];
    uint32_t na2 = 0;
#pragma unroll
    for (int jj = 0; jj < 32; jj++)
    {
      a[jj] = desc_a[(size_t)row * 32 + jj];
      na2 = __builtin_amdgcn_udot4(a[jj], a[jj], na2, false);
    }
    float bd = __builtin_inff(), sd = __builtin_inff();
    uint32_t bi = QMAX, si = QMAX;
    for (uint32_t c = threadIdx.x; c < nb; c += 256u)
    {
      const uint4 *pb = (const uint4 *)(desc_b + (size_t)c * 32);
      uint32_t dot = 0, nb2 = 0;
#pragma unroll
      for (int jj = 0; jj < 8; jj++)
      {
        const uint4 v = pb[jj];
        dot = __builtin_amdgcn_udot4(a[4 * jj + 0], v.x, dot, false), nb2 = __builtin_amdgcn_udot4(v.x, v.x, nb2, false);
        dot = __builtin_amdgcn_udot4(a[4 * jj + 1], v.y, dot, false), nb2 = __builtin_amdgcn_udot4(v.y, v.y, nb2, false);
        dot = __builtin_amdgcn_udot4(a[4 * jj + 2], v.z, dot, false), nb2 = __builtin_amdgcn_udot4(v.z, v.z, nb2, false);
        dot = __builtin_amdgcn_udot4(a[4 * jj + 3], v.w, dot, false), nb2 = __builtin_amdgcn_udot4(v.w, v.w, nb2, false);
      }
      const float d = sqrtf((float)(na2 + nb2 - 2u * dot));
      if (c < 2u)
        s_d01[c] = d;
      if (d < bd)
        sd = bd, si = bi, bd = d, bi = c;
      else if (d < sd)
        sd = d, si = c;
    }
    s_d[threadIdx.x][0] = bd, s_d[threadIdx.x][1] = sd;
    s_i[threadIdx.x][0] = bi, s_i[threadIdx.x][1] = si;
    __syncthreads();
    for (uint32_t step = 128; step >= 1; step >>= 1)
    {
      if (threadIdx.x < step)
      {
        const uint32_t o = threadIdx.x + step;
        float a1 = s_d[threadIdx.x][0], a2 = s_d[threadIdx.x][1], b1 = s_d[o][0], b2 = s_d[o][1];
        uint32_t ai1 = s_i[threadIdx.x][0], ai2 = s_i[threadIdx.x][1], bi1 = s_i[o][0], bi2 = s_i[o][1];
        auto less = [](float x, uint32_t xi, float y, uint32_t yi) { return x < y || (x == y && xi < yi); };
        const bool af = less(a1, ai1, b1, bi1);
        const float w1 = af ? a1 : b1, w2 = af ? a2 : b2, l1 = af ? b1 : a1;
        const uint32_t wi1 = af ? ai1 : bi1, wi2 = af ? ai2 : bi2, li1 = af ? bi1 : ai1;
        const bool w2f = less(w2, wi2, l1, li1);
        s_d[threadIdx.x][0] = w1, s_i[threadIdx.x][0] = wi1;
        s_d[threadIdx.x][1] = w2f ? w2 : l1, s_i[threadIdx.x][1] = w2f ? wi2 : li1;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0)
    {
      const bool sw = s_d01[0] == s_d01[1];
      const uint32_t k1 = s_i[0][0], k2 = s_i[0][1];
      uint32_t *m = matches + (size_t)row * 5;
      m[0] = a_index_base + row;
      m[1] = (sw && k1 < 2) ? (k1 ^ 1u) : k1;
      m[2] = (sw && k2 < 2) ? (k2 ^ 1u) : k2;
      m[3] = __float_as_uint(s_d[0][0]);
      m[4] = __float_as_uint(s_d[0][1]);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_match_pk: the matcher for reference sets of up to VKSIFT_HIP_MATCH_PK_NB rows (the pairs of a batched matching; single pairs
// below the size where pruning starts to pay).
//
// At this size pruning never pays: a row sees ~2 ln N_B = 15 insertions in 1.9 k columns, the bound stays loose, and the
// event-driven kernels spend their time in the candidate path (k_match_mfma<1,4,64> on 512 self-matches of 1.9 k x 1.9 k: 0.95 ms
// = 12.8 % of the matrix peak, the MFMAs a sixteenth of the issued instructions). Here every candidate is folded into the row's
// top-2 with THREE branch-free VALU instructions, by packing (d2, index) into one 32-bit key whose unsigned order is the
// reference's order:
//       K = ((2^20 - 1 - d2) << 12) | (4095 - index)      larger K = smaller d2, then smaller index (strict '<': earlier wins)
//   top-2 of a multiset of keys:   K2 = med3(K1, K2, K);  K1 = max(K1, K)       (order independent: a set, not a scan)
// With d2 = |a'|^2 + |b'|^2 - 2 a'.b' the key is linear in the dot product:
//       K = ((dot + Ra) << 13) + ck[col] + (pa << 12),    2 Ra + pa = 2^20 - 1 - |a'|^2,   ck[col] = 4095 - col - (|b'|^2 << 12)
// Ra rides in the MFMA's C operand (one register set per wave, constant over the scan), ck is a per-column word staged beside the
// B tile, pa (a per-row constant below one key step) is added after the scan: per candidate v_lshl_add_u32, v_med3_u32, v_max_u32.
// All arithmetic is modulo 2^32 and exact whenever d2 < 2^20 - 1 (real SIFT descriptors: d2 <= |a|^2 + |b|^2 <= 2^19; a key below
// 4096 — field 0 — is never trusted: columns beyond B are forced to key 0). For anything
// else the result is VERIFIED instead of assumed: the d2 of the two reported columns are recomputed exactly (2 x 128 bytes per
// row) and compared with the keys' fields; because the top-2 of a multiset does not depend on arrival order, two genuine keys at
// the top prove that no wrapped key (d2 >= 2^20, which can only look closer than it is) was ahead of them. A mismatch sends the
// row to k_match_redo, like d2 >= 2^22 in the other kernels. Quirk Q7: the tie of columns 0 and 1 is read off their two keys.
// Layout, staging and swizzle as in k_match32 (32x32x32 i8 MFMA, operand roles swapped, one query row per lane pair).
// Reference sets beyond 4096 rows: the scan goes in SUPER-CHUNKS of 4096 columns (the index field holds the column modulo 4096);
// after each one the four chains are folded into a running exact (d2, index) top-2 and restart from zero — a merge of exact
// top-2 lists, so the multiset argument above carries over to the union. A single large pair is also split over blockIdx.z
// (whole super-chunks, strided) so that the grid fills the chip; every (row, z) piece then leaves a partial list in the format of
// the stream-decomposed kernel and k_match_merge combines them.
template <int NW, int BTT>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(4, 4))) k_match_pk(const uint32_t *__restrict__ desc_a, const uint32_t *__restrict__ norm_a, uint32_t na,
                                                  uint32_t a_index_base, const uint32_t *__restrict__ desc_b, const uint32_t *__restrict__ norm_b,
                                                  uint32_t nb, uint32_t *__restrict__ matches, uint32_t *__restrict__ redo,
                                                  const uint32_t *__restrict__ n_dev, SlotStrides ss, SlotIds ids, uint32_t *__restrict__ partial)
{
  static_assert(BTT % 32 == 0 && 4096 % BTT == 0, "whole 32-column sub-blocks, whole tiles per super-chunk");
  constexpr uint32_t ROWS = 32u * NW, SC = 4096u;
  const uint32_t slot = ss.slot_fast ? blockIdx.x : blockIdx.y;
  const uint32_t rb0 = ss.slot_fast ? blockIdx.y : blockIdx.x, rb_step = ss.slot_fast ? gridDim.y : gridDim.x;
  const uint32_t ea = ss.use_ids ? ids.a[slot] : slot, eb = ss.use_ids ? ids.b[slot] : slot;
  desc_a += (size_t)ea * ss.desc_a, desc_b += (size_t)eb * ss.desc_b;
  norm_a += (size_t)ea * ss.norm_a, norm_b += (size_t)eb * ss.norm_b;
  matches += (size_t)slot * ss.matches;
  redo += (size_t)slot * ss.redo;
  if (n_dev)
  {
    n_dev += (size_t)slot * ss.n;
    na = n_dev[0];
    nb = n_dev[1] < 2u ? 2u : n_dev[1];
    if (nb > ss.pk_nb_max)
      return; // larger reference sets belong to the pruning kernels (which skip what this one takes)
  }
  if (na == 0)
    return;
  __shared__ __attribute__((aligned(16))) uint8_t s_b2[2][BTT * 128];
  __shared__ __attribute__((aligned(16))) uint32_t s_ck2[2][BTT];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const uint32_t nz = gridDim.z, z = blockIdx.z;
  constexpr int NTH = 64 * NW, NLD = (BTT * 8 + NTH - 1) / NTH;
  struct TileRegs
  {
    uint4 d[NLD];
    uint32_t n;
  };
  auto fetch_tile = [&](uint32_t t0, TileRegs &pf) {
#pragma unroll
    for (int q = 0; q < NLD; q++)
    {
      const int i = threadIdx.x + q * NTH;
      const int r = i >> 3, c = i & 7;
      pf.d[q] = make_uint4(0, 0, 0, 0);
      if (i < BTT * 8 && t0 + r < nb)
        pf.d[q] = ((const uint4 *)(desc_b + (size_t)(t0 + r) * 32))[c];
    }
    pf.n = (threadIdx.x < BTT && t0 + threadIdx.x < nb) ? norm_b[t0 + threadIdx.x] : 0u;
  };
  auto stage_tile = [&](uint32_t t0, int bufi, const TileRegs &pf) {
#pragma unroll
    for (int q = 0; q < NLD; q++)
    {
      const int i = threadIdx.x + q * NTH;
      const int r = i >> 3, c = i & 7;
      uint4 v = pf.d[q];
      v.x ^= 0x80808080u, v.y ^= 0x80808080u, v.z ^= 0x80808080u, v.w ^= 0x80808080u;
      if (i < BTT * 8)
        *(uint4 *)(s_b2[bufi] + r * 128 + swz(r, c) * 16) = v;
    }
    if ((int)threadIdx.x < BTT)
      s_ck2[bufi][threadIdx.x] = 4095u - ((t0 + threadIdx.x) & 4095u) - (pf.n << 12);
  };

  for (uint32_t rb = rb0; rb * ROWS < na; rb += rb_step)
  {
    const uint32_t row_base = (rb * NW + wave) * 32u;
    uint32_t r = row_base + j;
    if (r >= na)
      r = na - 1;
    v4i afrag[4];
    {
      const uint4 *p = (const uint4 *)(desc_a + (size_t)r * 32);
#pragma unroll
      for (int s = 0; s < 4; s++)
      {
        const uint4 v = p[2 * s + h];
        afrag[s] = v4i{(int)(v.x ^ 0x80808080u), (int)(v.y ^ 0x80808080u), (int)(v.z ^ 0x80808080u), (int)(v.w ^ 0x80808080u)};
      }
    }
    const uint32_t an = norm_a[r];
    const int ra = (int)((1u << 20) - 1u - an) >> 1;  // arithmetic: floor((2^20 - 1 - an) / 2)
    const uint32_t pa = ((1u << 20) - 1u - an) & 1u;
    v16i cra;
#pragma unroll
    for (int i = 0; i < 16; i++)
      cra[i] = ra;
    Top2 st{QMAX, QMAX, QMAX, QMAX}; // exact (d2, column) pairs of the super-chunks done so far
    uint32_t bad = 0;                // a folded key that cannot be genuine (field below 2): the row is replayed
    uint32_t sw = 0;                 // quirk Q7: d2(b0) == d2(b1)

    for (uint32_t sc0 = z * SC; sc0 < nb; sc0 += nz * SC)
    {
      const uint32_t sc_end = min(nb, sc0 + SC);
      // four independent (best, second) pairs, merged after the super-chunk: the med3 / max updates of one pair form a dependent chain
      // (16 steps per sub-block otherwise); a multiset's top-2 does not care how it was partitioned
      uint32_t k1[4] = {0, 0, 0, 0}, k2[4] = {0, 0, 0, 0};
      TileRegs pf0;
      __syncthreads(); // the previous scan has finished reading both buffers
      fetch_tile(sc0, pf0);
      stage_tile(sc0, 0, pf0);
      __syncthreads();
      int buf = 0;
      for (uint32_t t0 = sc0; t0 < sc_end; t0 += BTT, buf ^= 1)
      {
        const bool more = t0 + BTT < sc_end;
        if (more)
          fetch_tile(t0 + BTT, pf0);
        const uint8_t *s_b = s_b2[buf];
        const uint32_t *s_ck = s_ck2[buf];
        const bool partial_tile = t0 + BTT > nb; // columns beyond B in this tile (staged as zero rows): their keys are forced to 0
        // (issuing the MFMAs of sub-block s+1 between the key updates of sub-block s — a second accumulator set, sched_group_barrier —
        // was tried: 200+ VGPRs whatever the unrolling, two waves per SIMD, 0.55 -> 0.61 ms per 512 pairs; not kept)
#pragma unroll
        for (int sub = 0; sub < BTT / 32; sub++)
        {
          if (t0 + sub * 32 >= nb)
            break;
          const uint8_t *prow = s_b + (sub * 32 + j) * 128;
          v4i bf[4];
#pragma unroll
          for (int s = 0; s < 4; s++)
            bf[s] = *(const v4i *)(prow + swz(j, 2 * s + h) * 16);
          v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[0], afrag[0], cra, 0, 0, 0);
#pragma unroll
          for (int s = 1; s < 4; s++)
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[s], afrag[s], acc, 0, 0, 0);
          uint32_t ck[16];
#pragma unroll
          for (int b = 0; b < 4; b++)
          {
            const uint4 c4 = *(const uint4 *)(s_ck + sub * 32 + 8 * b + 4 * h);
            ck[4 * b + 0] = c4.x, ck[4 * b + 1] = c4.y, ck[4 * b + 2] = c4.z, ck[4 * b + 3] = c4.w;
          }
          uint32_t key[16];
#pragma unroll
          for (int i = 0; i < 16; i++)
            key[i] = ((uint32_t)acc[i] << 13) + ck[i];
          if (partial_tile)
          {
#pragma unroll
            for (int i = 0; i < 16; i++)
              if (t0 + sub * 32 + 8 * (i >> 2) + 4 * h + (i & 3) >= nb)
                key[i] = 0u;
          }
          if (sub == 0 && t0 == 0)
            sw = (h == 0 && (key[0] >> 12) == (key[1] >> 12)) ? 1u : 0u;
          // Four keys per chain and sub-block, two at a time: with K2 <= K1 the second largest of {K1, K2, a, b} is
          // max(K2, med3(K1, a, b)) and the largest max3(K1, a, b); the two K2 updates of a chain share one max3. Five instructions
          // per four candidates instead of eight (+ the key itself: 2.25 per candidate, not 3) — the same multiset top-2.
#pragma unroll
          for (int c = 0; c < 4; c++)
          {
            const uint32_t ta = umed3(k1[c], key[c], key[c + 4]);
            const uint32_t ka = umax3(k1[c], key[c], key[c + 4]);
            const uint32_t tb = umed3(ka, key[c + 8], key[c + 12]);
            k1[c] = umax3(ka, key[c + 8], key[c + 12]);
            k2[c] = umax3(k2[c], ta, tb);
          }
        }
        if (more)
          stage_tile(t0 + BTT, buf ^ 1, pf0);
        __syncthreads();
      }
      // fold the super-chunk: its two best keys (a key below 8192 before the parity term is "no column": field < 2 is never trusted)
      uint32_t kb = k1[0], ks = k2[0];
#pragma unroll
      for (int c = 1; c < 4; c++)
      {
        ks = max(min(kb, k1[c]), max(ks, k2[c]));
        kb = max(kb, k1[c]);
      }
#pragma unroll
      for (int c = 0; c < 2; c++)
      {
        const uint32_t k = c == 0 ? kb : ks;
        if (k >= 8192u)
        {
          const uint32_t kk = k + (pa << 12);
          const uint32_t q = (1u << 20) - 1u - (kk >> 12), idx = sc0 + 4095u - (kk & 4095u);
          // keys arrive best first and super-chunks in index order: equal d2 keeps the earlier column (insert_seq's rule)
          if (q < st.q2 || (q == st.q2 && idx < st.k2))
          {
            if (q < st.q1 || (q == st.q1 && idx < st.k1))
              st.q2 = st.q1, st.k2 = st.k1, st.q1 = q, st.k1 = idx;
            else
              st.q2 = q, st.k2 = idx;
          }
        }
        else if (k != 0u) // 0: no column (this lane saw fewer than two columns of the super-chunk)
          bad = 1u;
      }
    }

    // the two lanes of a row
    {
      Top2 o;
      o.q1 = __shfl_xor(st.q1, 32, 64), o.k1 = __shfl_xor(st.k1, 32, 64);
      o.q2 = __shfl_xor(st.q2, 32, 64), o.k2 = __shfl_xor(st.k2, 32, 64);
      st = merge2(st, o);
      sw |= __shfl_xor(sw, 32, 64);
      bad |= __shfl_xor(bad, 32, 64);
    }
    // verification: exact d2 of the two reported columns from the descriptor bytes (this lane's half of K, the partner's by shuffle).
    // A piece (z) that holds fewer than two columns of B reports QMAX entries: nothing to verify there.
    bool ok = bad == 0u;
#pragma unroll 1
    for (int c = 0; c < 2; c++) // not unrolled: the epilogue must not set the kernel's register count
    {
      const uint32_t qc = c == 0 ? st.q1 : st.q2, kc = c == 0 ? st.k1 : st.k2;
      const bool have = qc != QMAX;
      const uint32_t col = have ? min(kc, nb - 1u) : 0u;
      const uint4 *pb = (const uint4 *)(desc_b + (size_t)col * 32);
      int dot = 0;
#pragma unroll
      for (int s = 0; s < 4; s++)
      {
        const uint4 v = pb[2 * s + h];
        dot = __builtin_amdgcn_sdot4(afrag[s][0], (int)(v.x ^ 0x80808080u), dot, false);
        dot = __builtin_amdgcn_sdot4(afrag[s][1], (int)(v.y ^ 0x80808080u), dot, false);
        dot = __builtin_amdgcn_sdot4(afrag[s][2], (int)(v.z ^ 0x80808080u), dot, false);
        dot = __builtin_amdgcn_sdot4(afrag[s][3], (int)(v.w ^ 0x80808080u), dot, false);
      }
      dot += __shfl_xor(dot, 32, 64);
      const uint32_t qt = an + norm_b[col] - 2u * (uint32_t)dot;
      if (have && (kc >= nb || qt != qc))
        ok = false;
    }
    if (st.q1 != QMAX && st.q2 != QMAX && st.k1 == st.k2)
      ok = false;
    const uint32_t rr = row_base + j;
    if (h == 0 && rr < na)
    {
      if (nz > 1)
      {
        uint32_t *pp = partial + ((size_t)rr * nz + z) * 4;
        pp[0] = st.q1, pp[1] = st.k1, pp[2] = st.q2, pp[3] = st.k2;
        partial[(size_t)na * nz * 4 + (size_t)rr * nz + z] = (z == 0 ? sw : 0u) | ((ok ? 0u : 1u) << 1);
      }
      else
      {
        uint32_t *m = matches + (size_t)rr * 5;
        m[0] = a_index_base + rr;
        m[1] = (sw && st.k1 < 2) ? (st.k1 ^ 1u) : st.k1;
        m[2] = (sw && st.k2 < 2) ? (st.k2 ^ 1u) : st.k2;
        m[3] = __float_as_uint(sqrtf((float)st.q1));
        m[4] = __float_as_uint(sqrtf((float)st.q2));
        redo[rr] = (ok && st.q2 != QMAX) ? 0u : 1u;
      }
    }
  }
}

// Exact combination of the per-chunk partial top-2 lists of k_match_mfma (gridDim.z > 1): one thread per A row.
__global__ void __launch_bounds__(256) k_match_merge(const uint32_t *__restrict__ partial, uint32_t na, uint32_t nchunks, uint32_t a_index_base,
                                                     uint32_t *__restrict__ matches, uint32_t *__restrict__ redo, const uint32_t *__restrict__ n_dev,
                                                     uint32_t na_lo, uint32_t na_hi, uint32_t stream_grid, uint32_t nb, uint32_t tile_rows,
                                                     uint32_t block_rows)
{
  if (n_dev)
  {
    na = n_dev[0];
    nb = n_dev[1] < 2u ? 2u : n_dev[1];
    if (na <= na_lo || na > na_hi)
      return;
  }
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  if (r >= na)
    return;
  const uint32_t *pp = partial + (size_t)r * nchunks * 4;
  const uint32_t *fl = partial + (size_t)na * nchunks * 4 + (size_t)r * nchunks;
  Top2 s{pp[0], pp[1], pp[2], pp[3]};
  uint32_t flags = fl[0];
  if (stream_grid) // stream decomposition (k_match_mfma): the row block's pieces are the workgroups whose runs intersect its tiles
  {
    const uint32_t tiles = (nb + tile_rows - 1u) / tile_rows;
    const uint32_t span = stream_span((na + block_rows - 1u) / block_rows, tiles, stream_grid);
    const uint32_t rb = r / block_rows;
    nchunks = ((rb + 1u) * tiles - 1u) / span - (rb * tiles) / span + 1u; // the stride stays VKSIFT_HIP_MATCH_CHUNKS
  }
  for (uint32_t c = 1; c < nchunks; c++)
  {
    Top2 o{pp[c * 4 + 0], pp[c * 4 + 1], pp[c * 4 + 2], pp[c * 4 + 3]};
    s = merge2(s, o);
    flags |= fl[c] & 2u;
  }
  const bool sw = flags & 1u;
  uint32_t *m = matches + (size_t)r * 5;
  m[0] = a_index_base + r;
  m[1] = (sw && s.k1 < 2) ? (s.k1 ^ 1u) : s.k1;
  m[2] = (sw && s.k2 < 2) ? (s.k2 ^ 1u) : s.k2;
  m[3] = __float_as_uint(sqrtf((float)s.q1));
  m[4] = __float_as_uint(sqrtf((float)s.q2));
  redo[r] = (flags >> 1) & 1u;
}

// Small-problem variant: one workgroup = 16 A rows; its 4 waves each take one 16-row slice of every staged 64-row B
// tile (wave w sees B indices t0 + 16w + col, increasing over tiles, so the in-lane pre-filter stays valid), and the
// four partial top-2 lists are merged through LDS. 4x more waves than the row-per-wave kernel for a few thousand
// features, 4x shorter dependent chain per wave.
__global__ void __launch_bounds__(256) k_match_mfma_split(const uint32_t *__restrict__ desc_a, const uint32_t *__restrict__ norm_a, uint32_t na,
                                                          uint32_t a_index_base, const uint32_t *__restrict__ desc_b,
                                                          const uint32_t *__restrict__ norm_b, uint32_t nb, uint32_t *__restrict__ matches,
                                                          uint32_t *__restrict__ redo, const uint32_t *__restrict__ n_dev, uint32_t na_lo, uint32_t na_hi,
                                                          SlotStrides ss, SlotIds ids)
{
  const uint32_t ea = ss.use_ids ? ids.a[blockIdx.y] : blockIdx.y, eb = ss.use_ids ? ids.b[blockIdx.y] : blockIdx.y;
  desc_a += (size_t)ea * ss.desc_a, desc_b += (size_t)eb * ss.desc_b;
  norm_a += (size_t)ea * ss.norm_a, norm_b += (size_t)eb * ss.norm_b;
  matches += (size_t)blockIdx.y * ss.matches;
  redo += (size_t)blockIdx.y * ss.redo;
  if (n_dev)
  {
    n_dev += (size_t)blockIdx.y * ss.n;
    na = n_dev[0];
    nb = n_dev[1] < 2u ? 2u : n_dev[1];
    if (na <= na_lo || na > na_hi || nb <= ss.pk_nb_max)
      return;
  }
  if (blockIdx.x * 16u >= na)
    return;
  __shared__ __attribute__((aligned(16))) uint8_t s_b[2][BT * B_STRIDE];
  __shared__ uint32_t s_nb[2][BT];
  __shared__ uint32_t s_part[4][16][4];
  __shared__ uint32_t s_swap, s_risky;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, grp = lane >> 4;
  const uint32_t row_base = blockIdx.x * 16u;

  v4i afrag[2];
  uint32_t an[4];
  {
    uint32_t r = row_base + col;
    if (r >= na)
      r = na - 1;
    const uint4 *p = (const uint4 *)(desc_a + (size_t)r * 32);
    uint4 v0 = p[grp], v1 = p[4 + grp];
    afrag[0] = v4i{(int)(v0.x ^ 0x80808080u), (int)(v0.y ^ 0x80808080u), (int)(v0.z ^ 0x80808080u), (int)(v0.w ^ 0x80808080u)};
    afrag[1] = v4i{(int)(v1.x ^ 0x80808080u), (int)(v1.y ^ 0x80808080u), (int)(v1.z ^ 0x80808080u), (int)(v1.w ^ 0x80808080u)};
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
      uint32_t rr = row_base + grp * 4 + j;
      an[j] = norm_a[rr < na ? rr : na - 1];
    }
  }
  Top2 st[4];
#pragma unroll
  for (int j = 0; j < 4; j++)
    st[j] = Top2{QMAX, QMAX, QMAX, QMAX};
  uint32_t swap_bits = 0, risky_bits = 0;

  constexpr int NLD = BT * 8 / 256;
  uint4 pfb[NLD];
  uint32_t pfn = 0;
  auto fetch_tile = [&](uint32_t t0) {
#pragma unroll
    for (int q = 0; q < NLD; q++)
    {
      int i = threadIdx.x + q * 256;
      int r = i >> 3, c = i & 7;
      pfb[q] = make_uint4(0, 0, 0, 0);
      if (t0 + r < nb)
        pfb[q] = ((const uint4 *)(desc_b + (size_t)(t0 + r) * 32))[c];
    }
    pfn = (threadIdx.x < BT && t0 + threadIdx.x < nb) ? norm_b[t0 + threadIdx.x] : 0u;
  };
  fetch_tile(0);
  if (threadIdx.x == 0)
    s_swap = 0, s_risky = 0;

  int buf = 0;
  for (uint32_t t0 = 0; t0 < nb; t0 += BT, buf ^= 1)
  {
    // double-buffered staging: one barrier per tile
#pragma unroll
    for (int q = 0; q < NLD; q++)
    {
      int i = threadIdx.x + q * 256;
      int r = i >> 3, c = i & 7;
      uint4 v = pfb[q];
      v.x ^= 0x80808080u, v.y ^= 0x80808080u, v.z ^= 0x80808080u, v.w ^= 0x80808080u;
      *(uint4 *)(s_b[buf] + r * B_STRIDE + c * 16) = v;
    }
    if (threadIdx.x < BT)
      s_nb[buf][threadIdx.x] = pfn;
    __syncthreads();
    if (t0 + BT < nb)
      fetch_tile(t0 + BT);

    const uint32_t sub0 = t0 + 16u * wave;
    if (sub0 < nb)
    {
      const uint32_t bcol = sub0 + col;
      const uint8_t *pb = s_b[buf] + (16 * wave + col) * B_STRIDE + grp * 16;
      const v4i b0 = *(const v4i *)pb;
      const v4i b1 = *(const v4i *)(pb + 64);
      const uint32_t bn = s_nb[buf][16 * wave + col];
      v4i acc = v4i{0, 0, 0, 0};
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[0], b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[1], b1, acc, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; j++)
      {
        const uint32_t q = bcol < nb ? an[j] + bn - 2u * (uint32_t)acc[j] : QMAX;
        uint32_t key = bcol;
        if (sub0 == 0)
        {
          const uint32_t other = __shfl_xor(q, 1, 64);
          const bool sw = col < 2 && q == other;
          if (sw)
            swap_bits |= 1u << j;
          if (col < 2 && (q >= Q_EXACT || other >= Q_EXACT))
            risky_bits |= 1u << j;
          key = sw ? (uint32_t)(col ^ 1) : bcol;
        }
        if (q < st[j].q2)
        {
          if (q >= Q_EXACT)
            risky_bits |= 1u << j;
          insert_seq(st[j], q, key);
        }
      }
    }
  }

  // merge across the 16 lanes of a row group, then across the 4 waves
#pragma unroll
  for (int m = 1; m < 16; m <<= 1)
  {
    risky_bits |= __shfl_xor(risky_bits, m, 64);
    swap_bits |= __shfl_xor(swap_bits, m, 64);
  }
#pragma unroll
  for (int j = 0; j < 4; j++)
  {
    Top2 s = st[j];
#pragma unroll
    for (int m = 1; m < 16; m <<= 1)
    {
      Top2 o;
      o.q1 = __shfl_xor(s.q1, m, 64), o.k1 = __shfl_xor(s.k1, m, 64);
      o.q2 = __shfl_xor(s.q2, m, 64), o.k2 = __shfl_xor(s.k2, m, 64);
      s = merge2(s, o);
    }
    if (col == 0)
    {
      s_part[wave][grp * 4 + j][0] = s.q1, s_part[wave][grp * 4 + j][1] = s.k1;
      s_part[wave][grp * 4 + j][2] = s.q2, s_part[wave][grp * 4 + j][3] = s.k2;
    }
  }
  if (col == 0)
  {
    atomicOr(&s_risky, risky_bits << (grp * 4));
    if (wave == 0)
      atomicOr(&s_swap, swap_bits << (grp * 4));
  }
  __syncthreads();
  if (threadIdx.x < 16)
  {
    const int rr = threadIdx.x;
    Top2 s{s_part[0][rr][0], s_part[0][rr][1], s_part[0][rr][2], s_part[0][rr][3]};
#pragma unroll
    for (int w = 1; w < 4; w++)
    {
      Top2 o{s_part[w][rr][0], s_part[w][rr][1], s_part[w][rr][2], s_part[w][rr][3]};
      s = merge2(s, o);
    }
    const uint32_t r = row_base + rr;
    if (r < na)
    {
      const bool sw = (s_swap >> rr) & 1u;
      uint32_t *m = matches + (size_t)r * 5;
      m[0] = a_index_base + r;
      m[1] = (sw && s.k1 < 2) ? (s.k1 ^ 1u) : s.k1;
      m[2] = (sw && s.k2 < 2) ? (s.k2 ^ 1u) : s.k2;
      m[3] = __float_as_uint(sqrtf((float)s.q1));
      m[4] = __float_as_uint(sqrtf((float)s.q2));
      redo[r] = (s_risky >> rr) & 1u;
    }
  }
}

// Exact replay of Get2NearestNeighbors.comp:43-103 for the rows flagged by the MFMA kernels (candidates with
// d2 >= 2^22, where the reference's float sqrt comparison is not injective). One thread per flagged row, scalar.
// Never taken for real SIFT descriptors; adversarial inputs just run at scalar speed.
__global__ void __launch_bounds__(64) k_match_redo(const uint32_t *__restrict__ desc_a, uint32_t na, uint32_t a_index_base,
                                                   const uint32_t *__restrict__ desc_b, uint32_t nb, uint32_t *__restrict__ matches,
                                                   const uint32_t *__restrict__ redo, const uint32_t *__restrict__ n_dev, SlotStrides ss, SlotIds ids)
{
  const uint32_t ea = ss.use_ids ? ids.a[blockIdx.y] : blockIdx.y, eb = ss.use_ids ? ids.b[blockIdx.y] : blockIdx.y;
  desc_a += (size_t)ea * ss.desc_a, desc_b += (size_t)eb * ss.desc_b;
  matches += (size_t)blockIdx.y * ss.matches;
  redo += (size_t)blockIdx.y * ss.redo;
  if (n_dev)
  {
    n_dev += (size_t)blockIdx.y * ss.n;
    na = n_dev[0];
    nb = n_dev[1] < 2u ? 2u : n_dev[1];
  }
  for (uint32_t row = blockIdx.x * 64 + threadIdx.x; row < na; row += gridDim.x * 64)
  {
    if (!redo[row])
      continue;
    uint32_t a[32];
    uint32_t na2 = 0;
#pragma unroll
    for (int j = 0; j < 32; j++)
    {
      a[j] = desc_a[(size_t)row * 32 + j];
      na2 = __builtin_amdgcn_udot4(a[j], a[j], na2, false);
    }
    auto dist = [&](uint32_t bi) -> float {
      const uint32_t *pb = desc_b + (size_t)bi * 32;
      uint32_t dot = 0, nb2 = 0;
#pragma unroll
      for (int j = 0; j < 32; j++)
      {
        const uint32_t v = pb[j];
        dot = __builtin_amdgcn_udot4(a[j], v, dot, false);
        nb2 = __builtin_amdgcn_udot4(v, v, nb2, false);
      }
      return sqrtf((float)(na2 + nb2 - 2u * dot));
    };
    float d0 = dist(0), d1 = dist(1);
    float best_d, second_d;
    uint32_t best_i, second_i;
    if (d0 < d1)
      best_d = d0, best_i = 0, second_d = d1, second_i = 1;
    else
      best_d = d1, best_i = 1, second_d = d0, second_i = 0;
    for (uint32_t bi = 2; bi < nb; bi++)
    {
      const float d = dist(bi);
      if (d < best_d)
      {
        second_d = best_d, second_i = best_i;
        best_d = d, best_i = bi;
      }
      else if (d < second_d)
        second_d = d, second_i = bi;
    }
    uint32_t *m = matches + (size_t)row * 5;
    m[0] = a_index_base + row;
    m[1] = best_i;
    m[2] = second_i;
    m[3] = __float_as_uint(best_d);
    m[4] = __float_as_uint(second_d);
  }
}

} // namespace

extern "C"
{
  int vksift_hip_shifted_norms(const uint8_t *desc, uint32_t n, uint32_t *norms, vksift_hip_stream s)
  {
    if (n == 0)
      return 0;
    hipLaunchKernelGGL(k_shifted_norms, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)s, (const uint32_t *)desc, n, norms);
    return (int)hipGetLastError();
  }

  int vksift_hip_match_2nn_prenormed(const uint8_t *desc_a, const uint32_t *norm_a, uint32_t na, uint32_t a_index_base, const uint8_t *desc_b,
                                     const uint32_t *norm_b, uint32_t nb, uint32_t *scratch, size_t scratch_u32, uint8_t *matches, vksift_hip_stream s)
  {
    if (na == 0)
      return 0;
    if (nb < 2)
      return (int)hipErrorInvalidValue; /* callers pad B to two rows (quirk Q6) */
    if (scratch == nullptr || scratch_u32 < vksift_hip_match_scratch_u32(na, nb) - 2u * (size_t)na - (size_t)nb)
      return (int)hipErrorInvalidValue; /* a buffer sized by an older formula: nothing is launched */
    hipStream_t hs = (hipStream_t)s;
    uint32_t *redo = scratch;
    const uint32_t *da = (const uint32_t *)desc_a, *db = (const uint32_t *)desc_b;
    const SlotStrides z{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const SlotIds noids{};
    /* A small problem (a few hundred thousand distances): 16 A rows per workgroup with B split over its waves, one launch.
     * Everything else: the stream decomposition (see k_match_mfma) — 8 waves x 32 rows per workgroup, 128-row B tiles, two
     * workgroups per CU, every workgroup the same number of tiles. Measured on MI355X against the three size regimes it
     * replaced (rows x rows, whole call): 2k 0.033 -> 0.028 ms, 8k 0.099 -> 0.062, 16k 0.23 -> 0.126, 32k 0.50 -> 0.28,
     * 50k 0.60 -> 0.51 (31.7 % of the dense int8 peak), 100k 1.78 -> 1.48 (44 %). */
    if (match_use_pk() && nb <= VKSIFT_HIP_MATCH_PK_NB && (uint64_t)na * nb >= 64000000ull)
    {
      /* 8 k x 8 k up to 32 k reference rows: the branch-free packed-key kernel (its rate does not depend on the size: 13 k x 13 k
       * 0.103 -> 0.078 ms, 32 k 0.275 -> 0.243; the pruning kernel passes it beyond 32 k rows, below 8 k the launches dominate). Row blocks x strided pieces of B (whole super-chunks of 4096 columns) sized to put ~2 workgroups on every
       * CU; more than one piece per row block: partial lists + k_match_merge. Scratch: redo[na], then 5 * na * pieces u32
       * (pieces <= 8 <= VKSIFT_HIP_MATCH_CHUNKS). */
      uint32_t *partial = redo + na;
      const uint32_t nsc = (nb + 4095u) / 4096u, want = 2u * device_cus();
      SlotStrides zp = z;
      zp.pk_nb_max = VKSIFT_HIP_MATCH_PK_NB;
      if (na <= 4096u)
      {
        const uint32_t nrb = (na + 63u) / 64u;
        uint32_t nz = (want + nrb - 1u) / nrb;
        nz = nz < nsc ? nz : nsc;
        hipLaunchKernelGGL((k_match_pk<2, 64>), dim3(nrb, 1, nz), dim3(128), 0, hs, da, norm_a, na, a_index_base, db, norm_b, nb, (uint32_t *)matches, redo,
                           (const uint32_t *)nullptr, zp, noids, partial);
        if (nz > 1)
          hipLaunchKernelGGL(k_match_merge, dim3((na + 255u) / 256u), dim3(256), 0, hs, (const uint32_t *)partial, na, nz, a_index_base, (uint32_t *)matches, redo,
                             (const uint32_t *)nullptr, 0u, 0xFFFFFFFFu, 0u, nb, 0u, 0u);
      }
      else
      {
        const uint32_t nrb = (na + 127u) / 128u;
        uint32_t nz = (want + nrb - 1u) / nrb;
        nz = nz < nsc ? nz : nsc;
        hipLaunchKernelGGL((k_match_pk<4, 128>), dim3(nrb, 1, nz), dim3(256), 0, hs, da, norm_a, na, a_index_base, db, norm_b, nb, (uint32_t *)matches, redo,
                           (const uint32_t *)nullptr, zp, noids, partial);
        if (nz > 1)
          hipLaunchKernelGGL(k_match_merge, dim3((na + 255u) / 256u), dim3(256), 0, hs, (const uint32_t *)partial, na, nz, a_index_base, (uint32_t *)matches, redo,
                             (const uint32_t *)nullptr, 0u, 0xFFFFFFFFu, 0u, nb, 0u, 0u);
      }
    }
    else if (na <= VKSIFT_HIP_MATCH_SMALL_NA && nb <= VKSIFT_HIP_MATCH_SMALL_NB)
      hipLaunchKernelGGL(k_match_mfma_split, dim3((na + 15u) / 16u), dim3(256), 0, hs, da, norm_a, na, a_index_base, db, norm_b, nb, (uint32_t *)matches, redo,
                         (const uint32_t *)nullptr, 0u, 0xFFFFFFFFu, z, noids);
    else if (match_use_scan())
    {
      /* large reference sets: the branch-free cell scan, the exact finish from the descriptor bytes, the (rare) replay of tied rows.
       * Scratch: [row list: 1 + na][pad to 16 B][cell lists: 16 * na * VKSIFT_HIP_MATCH_CHUNKS] (VKSIFT_HIP_MATCH_SCRATCH_U32). */
      uint32_t *list = scratch;
      uint32_t *cells = (uint32_t *)(((uintptr_t)(scratch + na + 1u) + 15u) & ~(uintptr_t)15u); /* 16 words per (row, piece) */
      /* Eight waves per workgroup, one workgroup per CU (the two waves of a SIMD share the staged tile: half the staging per MFMA of
       * round 4's 4-wave form), 96 query rows per wave (three MFMAs per B fragment read from LDS), 256-row tiles of B (half the
       * barriers and exposed pipeline ends per MFMA of 128-row tiles; 512 gives nothing more), MFMAs and folds interleaved as the
       * source spells them (k_match_scan32: PIPE). 50 k x 50 k on one box, whole call: 0.430 ms (<2,4,128>, round 4) -> 0.413 (8 waves)
       * -> 0.390 (pinned interleave) -> 0.370 (256-row tiles) -> 0.354 (96 rows per wave; 12 waves x 64 rows: 0.358, 128 rows per
       * wave: 0.360 at 256 VGPRs, 12 waves x 96 rows: 0.50); 100 k x 100 k 1.50 -> 1.18 ms = 55 % of the int8 peak.
       * VKSIFT_TUNE_SCAN_FORM = 1: round 4's form (A/B). */
      const bool r4 = vksift_hip_tune_get(VKSIFT_TUNE_SCAN_FORM) == 1;
      const uint32_t Gs = r4 ? 2u * device_cus() : device_cus(), brows = r4 ? 256u : 768u, trows = r4 ? 128u : 256u;
      if (r4)
        hipLaunchKernelGGL((k_match_scan32<2, 4, 128, 0>), dim3(Gs), dim3(256), 0, hs, da, na, db, norm_b, nb, cells, list);
      else
        hipLaunchKernelGGL((k_match_scan32<3, 8, 256, 1>), dim3(Gs), dim3(512), 0, hs, da, na, db, norm_b, nb, cells, list);
      hipLaunchKernelGGL(k_match_fix, dim3((na + 31u) / 32u), dim3(256), 0, hs, (const uint32_t *)cells, da, norm_a, na, a_index_base, db, norm_b, nb, Gs, trows, brows,
                         (uint32_t *)matches, list);
      hipLaunchKernelGGL(k_match_redo_rows, dim3(512), dim3(256), 0, hs, da, a_index_base, db, nb, (uint32_t *)matches, (const uint32_t *)list);
      return (int)hipGetLastError();
    }
    else
    {
      uint32_t *partial = redo + na;
      const uint32_t G = 2u * device_cus();
      hipLaunchKernelGGL((k_match_mfma<2, 8, 128>), dim3(G), dim3(512), 0, hs, da, norm_a, na, a_index_base, db, norm_b, nb, (uint32_t *)matches, redo,
                         (const uint32_t *)nullptr, 0u, 0xFFFFFFFFu, z, partial, noids, 1u);
      hipLaunchKernelGGL(k_match_merge, dim3((na + 255u) / 256u), dim3(256), 0, hs, (const uint32_t *)partial, na, (uint32_t)VKSIFT_HIP_MATCH_CHUNKS, a_index_base,
                         (uint32_t *)matches, redo, (const uint32_t *)nullptr, 0u, 0xFFFFFFFFu, G, nb, 128u, 256u);
    }
    uint32_t rblocks = (na + 63u) / 64u;
    hipLaunchKernelGGL(k_match_redo, dim3(rblocks > 1024u ? 1024u : rblocks), dim3(64), 0, hs, da, na, a_index_base, db, nb, (uint32_t *)matches,
                       (const uint32_t *)redo, (const uint32_t *)nullptr, z, noids);
    return (int)hipGetLastError();
  }

  size_t vksift_hip_match_scratch_u32(uint32_t na, uint32_t nb) { return VKSIFT_HIP_MATCH_SCRATCH_U32(na, nb); }

  int vksift_hip_match_2nn_desc(const uint8_t *desc_a, uint32_t na, uint32_t a_index_base, const uint8_t *desc_b, uint32_t nb, uint32_t *norm_scratch,
                                size_t scratch_u32, uint8_t *matches, vksift_hip_stream s)
  {
    if (na == 0)
      return 0;
    if (nb < 2)
      return (int)hipErrorInvalidValue; /* callers pad B to two rows (quirk Q6) */
    if (norm_scratch == nullptr || scratch_u32 < vksift_hip_match_scratch_u32(na, nb))
      return (int)hipErrorInvalidValue;
    uint32_t *norm_a = norm_scratch, *norm_b = norm_scratch + na;
    /* both norm arrays in one launch (they are consecutive in the scratch and the rows are independent) when the descriptor sets are too */
    int e = 0;
    if (desc_b == desc_a + (size_t)na * 128u)
      e = vksift_hip_shifted_norms(desc_a, na + nb, norm_a, s);
    else
    {
      e = vksift_hip_shifted_norms(desc_a, na, norm_a, s);
      if (e == 0)
        e = vksift_hip_shifted_norms(desc_b, nb, norm_b, s);
    }
    if (e == 0)
      e = vksift_hip_match_2nn_prenormed(desc_a, norm_a, na, a_index_base, desc_b, norm_b, nb, norm_scratch + na + nb, scratch_u32 - na - nb, matches, s);
    return e;
  }

  int vksift_hip_match_2nn_async(const uint8_t *cache_desc, const uint32_t *cache_norm, const uint32_t *cache_n, const uint32_t *ids_a, const uint32_t *ids_b,
                                 uint32_t max_na, uint32_t max_nb, uint32_t nb_exact, uint32_t *redo, uint32_t *n_dev, uint8_t *matches, uint32_t nslots, uint64_t cache_desc_stride,
                                 uint64_t cache_norm_stride, uint64_t redo_slot_stride, uint64_t match_slot_stride, uint32_t n_slot_stride,
                                 uint32_t *partial_scratch, vksift_hip_stream s)
  {
    if (nslots < 1 || nslots > VKSIFT_HIP_MATCH_SLOTS)
      return (int)hipErrorInvalidValue;
    SlotIds ids;
    for (uint32_t i = 0; i < VKSIFT_HIP_MATCH_SLOTS; i++)
      ids.a[i] = i < nslots ? ids_a[i] : 0u, ids.b[i] = i < nslots ? ids_b[i] : 0u;
    hipLaunchKernelGGL(k_slot_counts, dim3((nslots + 63u) / 64u), dim3(64), 0, (hipStream_t)s, cache_n, 1u, ids, nslots, n_dev, n_slot_stride);
    if (max_na == 0)
      return (int)hipGetLastError();
    const uint8_t *desc_a = cache_desc, *desc_b = cache_desc;
    const uint32_t *norm_a = cache_norm, *norm_b = cache_norm;
    SlotStrides ss;
    ss.desc_a = ss.desc_b = cache_desc_stride / 4;
    ss.norm_a = ss.norm_b = cache_norm_stride;
    ss.matches = match_slot_stride / 4;
    ss.n = n_slot_stride;
    ss.redo = redo_slot_stride;
    ss.slot_fast = 0;
    ss.use_ids = 1;
    ss.pk_nb_max = 0;
    ss.nslots_loop = 0;
    hipStream_t hs = (hipStream_t)s;
    const uint32_t *da = (const uint32_t *)desc_a, *db = (const uint32_t *)desc_b;
    if (nslots == 1 && partial_scratch)
    {
      /* one pair: as vksift_hip_match_2nn_prenormed, but N_A is only known on the device — both kernels are launched, each
       * returns at once unless N_A falls in its range; the stream decomposition sizes its runs from the device-side counts */
      const uint32_t SS = VKSIFT_HIP_MATCH_SMALL_NA;
      const uint32_t n1 = max_na < SS ? max_na : SS;
      hipLaunchKernelGGL(k_match_mfma_split, dim3((n1 + 15u) / 16u, 1), dim3(256), 0, hs, da, norm_a, 0u, 0u, db, norm_b, 0u, (uint32_t *)matches, redo, n_dev, 0u,
                         SS, ss, ids);
      if (max_na > SS)
      {
        const uint32_t G = 2u * device_cus();
        hipLaunchKernelGGL((k_match_mfma<2, 8, 128>), dim3(G), dim3(512), 0, hs, da, norm_a, 0u, 0u, db, norm_b, 0u, (uint32_t *)matches, redo, n_dev, SS,
                           0xFFFFFFFFu, ss, partial_scratch, ids, 1u);
        hipLaunchKernelGGL(k_match_merge, dim3((max_na + 255u) / 256u), dim3(256), 0, hs, (const uint32_t *)partial_scratch, 0u,
                           (uint32_t)VKSIFT_HIP_MATCH_CHUNKS, 0u, (uint32_t *)matches, redo, n_dev, SS, 0xFFFFFFFFu, G, 0u, 128u, 256u);
      }
    }
    else
    {
      /* A batch of pairs (regimes: the table at the top of this file). The row counts are only known on the device: launch for the
       * capacity, surplus workgroups exit at once, a kernel returns immediately unless N_A falls in its range. The B-split kernel keeps
       * every CU busy when few pairs of a few thousand rows are matched; a batch of 8 and more pairs has enough workgroups anyway and
       * runs 24 % faster with 64 rows per workgroup (64 pairs of 1.9k x 1.9k: 0.285 -> 0.217 ms). */
      uint32_t S1 = nslots >= 8 ? 1024u : 8192u;
      const uint32_t S2 = 32768u;
      /* regimes 2/3 loop over their row blocks, so their grids stay small even when only the capacity is known */
      uint32_t lim = nslots >= 8 ? 64u : 1024u;
      /* reference sets of up to 4096 rows — the frames of a batch — take the branch-free packed-key kernel (k_match_pk): 256 query
       * rows per workgroup, looping over the row blocks of its slot; the pruning kernels below skip those slots (ss.pk_nb_max).
       * What is left for them in such a batch is the odd pair with a large reference set: one kernel for all N_A <= 32768 (the
       * B-split kernel is not launched) on a grid of 16 row blocks per slot — every launch of a mostly idle grid costs the batch
       * 4-5 us (4096 workgroups that only read their slot's counts) */
      const bool pk = match_use_pk();
      /* what is left for the pruning kernels: the slots beyond the packed-key range — if the host KNOWS there is one (nb_exact). A batch whose
       * counts are still on the device (the usual case: the matching is queued behind its detection) is served by the packed-key kernel alone,
       * whatever the N_B of a slot turns out to be: no grid is queued for slots that "could" exist (vksift_hip.h) */
      const bool prune = !pk || (max_nb > VKSIFT_HIP_MATCH_PK_NB && nb_exact != 0u);
      if (pk)
      {
        ss.pk_nb_max = prune ? VKSIFT_HIP_MATCH_PK_NB : 0xFFFFFFFFu;
        SlotStrides sp = ss;
        sp.slot_fast = nslots > 1 ? 1u : 0u;
        /* (8 waves x 128-column tiles: swept against <4,128>, <4,64>, <8,64>, <16,128>, <2,64> on 512 self-matches of 1.9 k x 1.9 k:
         * 0.387 ms against 0.48 / 0.49 / 0.397 / 0.418 / 0.51-0.57 ms) */
        uint32_t gp = (max_na + 255u) / 256u;
        gp = gp < 16u ? gp : 16u;
        hipLaunchKernelGGL((k_match_pk<8, 128>), sp.slot_fast ? dim3(nslots, gp) : dim3(gp, nslots), dim3(512), 0, hs, da, norm_a, 0u, 0u, db, norm_b, 0u,
                           (uint32_t *)matches, redo, n_dev, sp, ids, (uint32_t *)nullptr);
        S1 = 0u;
        lim = nslots >= 8 ? 16u : lim;
      }
      auto bounded = [lim](uint32_t blocks) { return blocks < lim ? blocks : lim; };
      const uint32_t n1 = max_na < S1 ? max_na : S1;
      if (n1 > 0)
        hipLaunchKernelGGL(k_match_mfma_split, dim3((n1 + 15u) / 16u, nslots), dim3(256), 0, hs, da, norm_a, 0u, 0u, db, norm_b, 0u, (uint32_t *)matches, redo, n_dev,
                           0u, S1, ss, ids);
      /* What the packed-key kernel leaves (reference sets beyond 32 768 rows) and, without it, everything above S1: two size regimes
       * of the pruning kernel, (S1, 32768] with 16 rows per wave and above with 32 (B fragment reuse). With the packed-key kernel these
       * are the odd slots of a batch, usually none: at most 16 slot positions x 16 row-block positions = 256 workgroups per regime, every
       * one walking the slots of the call (ss.nslots_loop) and the row blocks of a slot that is its own. Without it they carry the
       * whole batch: one workgroup column per slot, slot index fastest so that the busy workgroups are contiguous in dispatch order. */
      SlotStrides s2 = ss;
      s2.slot_fast = nslots > 1 ? 1u : 0u;
      uint32_t gs = nslots;
      if (pk && nslots > 16u)
        gs = 16u, s2.nslots_loop = nslots;
      if (prune && max_na > S1)
      {
        const uint32_t n2 = max_na < S2 ? max_na : S2;
        const uint32_t gb = bounded((n2 + 63u) / 64u);
        hipLaunchKernelGGL(k_match_mfma<1>, s2.slot_fast ? dim3(gs, gb) : dim3(gb, gs), dim3(256), 0, hs, da, norm_a, 0u, 0u, db, norm_b, 0u,
                           (uint32_t *)matches, redo, n_dev, S1, S2, s2, (uint32_t *)nullptr, ids, 0u);
      }
      if (prune && max_na > S2)
      {
        const uint32_t gb = bounded((max_na + 127u) / 128u);
        hipLaunchKernelGGL(k_match_mfma<2>, s2.slot_fast ? dim3(gs, gb) : dim3(gb, gs), dim3(256), 0, hs, da, norm_a, 0u, 0u, db, norm_b, 0u,
                           (uint32_t *)matches, redo, n_dev, S2, 0xFFFFFFFFu, s2, (uint32_t *)nullptr, ids, 0u);
      }
    }
    uint32_t rblocks = (max_na + 63u) / 64u;
    const uint32_t rlim = nslots > 64u ? 16u : 64u; /* grid-stride over the rows: the flags are all zero for real descriptors */
    if (rblocks > rlim)
      rblocks = rlim;
    hipLaunchKernelGGL(k_match_redo, dim3(rblocks, nslots), dim3(64), 0, hs, da, 0u, 0u, db, 0u, (uint32_t *)matches, (const uint32_t *)redo, n_dev, ss, ids);
    return (int)hipGetLastError();
  }
}
