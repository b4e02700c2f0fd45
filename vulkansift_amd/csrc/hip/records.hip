// records.hip — the record-moving launches around the matcher: SIFT buffer sections -> the matcher's dense rows (pack_BufferMemory,
// sift_memory.c:957-1047: the gather pass that serves uploaded buffers and every buffer an instance matches before its cache exists; freshly
// detected buffers get their rows from the descriptor launch, features.hip), sections -> dense 164-byte records for the batched
// download, and the GPU-side cross-check + ratio filter over 2-NN records (src/examples/test_sift_match.cpp:90-107). Split off match.hip
// in round 6: none of this is matcher arithmetic.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "vksift_hip.h"

namespace
{

__global__ void __launch_bounds__(256) k_gather_desc(const uint8_t *__restrict__ feats, uint32_t n, uint32_t *__restrict__ desc)
{
  uint32_t i = blockIdx.x * 256 + threadIdx.x; // dword index
  if (i >= n * 32u)
    return;
  uint32_t row = i >> 5, j = i & 31u;
  desc[i] = *(const uint32_t *)(feats + (size_t)row * 164 + 36 + 4 * j);
}

// Gather the descriptors of a (sectioned or packed) SIFT buffer into dense rows in download order AND compute
// their shifted norms, with the per-section feature counts read on the device (no host round trip):
// row -> section by scanning the <= 16 section counts; eight lanes per row, 16 bytes per lane.
struct SectionTable
{
  uint32_t nsec;
  uint32_t off[16];   // first feature of each section inside the buffer
  uint32_t cap[16];   // capacity (stored = min(found, cap))
  uint32_t fixed[16]; // used instead of found[] when found == nullptr (uploaded / packed buffers)
};

struct SlotMap
{
  uint32_t buf[64]; // SIFT buffer index handled by slot blockIdx.y
};

// (all buffers of a batched detection in ONE launch: 512 buffers = 8 launches of 64 slots x 256 blocks until round 5 — 25 us each alone,
// 330 us each queued behind the next detection's blur launches, 16 384 mostly idle workgroups per launch)
struct GatherMap
{
  uint32_t buf[VKSIFT_HIP_GATHER_SLOTS]; // SIFT buffer index handled by slot blockIdx.y
};

__global__ void __launch_bounds__(256) k_gather_sections(const uint8_t *__restrict__ feats_base, uint64_t buf_stride, GatherMap map, SectionTable tab,
                                                         const uint32_t *__restrict__ found_base, uint32_t found_buf_stride, uint32_t *__restrict__ desc,
                                                         uint64_t desc_slot_stride, uint32_t *__restrict__ norms, uint64_t norm_slot_stride,
                                                         uint32_t *__restrict__ n_out, uint32_t n_slot_stride, uint32_t pad_rows_to)
{
  const uint32_t slot = blockIdx.y;
  const uint32_t bufi = map.buf[slot];
  const uint8_t *feats = feats_base + (size_t)bufi * buf_stride;
  const uint32_t *found = found_base ? found_base + (size_t)bufi * found_buf_stride : nullptr;
  // the dense rows, their norms and the row count land in the per-BUFFER cache entry (reused by every later match of the buffer)
  desc += (size_t)bufi * desc_slot_stride;
  norms += (size_t)bufi * norm_slot_stride;
  n_out += (size_t)bufi * n_slot_stride;

  const uint32_t j = threadIdx.x & 7u; // 16 bytes of a row per lane: a wave moves 8 rows at a time, a workgroup 32
  // stored count of every section (uniform), then a grid-stride walk over the rows that exist
  uint32_t cnt[16];
  uint32_t total = 0;
#pragma unroll
  for (uint32_t o = 0; o < 16; o++)
  {
    uint32_t n = 0;
    if (o < tab.nsec)
    {
      n = found ? found[o] : tab.fixed[o];
      n = n < tab.cap[o] ? n : tab.cap[o];
    }
    cnt[o] = n;
    total += n;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    *n_out = total;
  const uint32_t nrows = total > pad_rows_to ? total : pad_rows_to;
  for (uint32_t row = blockIdx.x * 32 + (threadIdx.x >> 3); row < nrows; row += gridDim.x * 32)
  {
    uint4 v = uint4{0u, 0u, 0u, 0u}; // rows in [total, pad_rows_to): quirk Q6 padding, all-zero descriptors
    if (row < total)
    {
      uint32_t base = 0, src_row = 0;
#pragma unroll
      for (uint32_t o = 0; o < 16; o++)
      {
        if (row >= base && row < base + cnt[o])
          src_row = tab.off[o] + (row - base);
        base += cnt[o];
      }
      // (records are 164 bytes, the descriptor starts at byte 36: dword-aligned 16-byte loads)
      const uint32_t *p = (const uint32_t *)(feats + (size_t)src_row * 164 + 36 + 16 * j);
      v = uint4{p[0], p[1], p[2], p[3]};
    }
    *(uint4 *)(desc + (size_t)row * 32 + 4 * j) = v;
    uint32_t s2 = __builtin_amdgcn_udot4(v.x, v.x, 0u, false), s1 = __builtin_amdgcn_udot4(v.x, 0x01010101u, 0u, false);
    s2 = __builtin_amdgcn_udot4(v.y, v.y, s2, false), s1 = __builtin_amdgcn_udot4(v.y, 0x01010101u, s1, false);
    s2 = __builtin_amdgcn_udot4(v.z, v.z, s2, false), s1 = __builtin_amdgcn_udot4(v.z, 0x01010101u, s1, false);
    s2 = __builtin_amdgcn_udot4(v.w, v.w, s2, false), s1 = __builtin_amdgcn_udot4(v.w, 0x01010101u, s1, false);
#pragma unroll
    for (int d = 4; d >= 1; d >>= 1)
    {
      s2 += __shfl_xor(s2, d, 64);
      s1 += __shfl_xor(s1, d, 64);
    }
    if (j == 0)
      norms[row] = s2 - 256u * s1 + 128u * 128u * 128u;
  }
}

// Download packing (sift_memory.c:957-1047 pack_BufferMemory, here for a whole batch of buffers at once): the stored features of
// slot blockIdx.y's buffer — its sections in order, min(found, capacity) records each — as dense 164-byte records at
// out + out_row[slot] * 164. One dword per thread step; the host then fetches all buffers of a detection with ONE copy.
struct PackOffsets
{
  uint32_t row[64];
};
__global__ void __launch_bounds__(256) k_pack_features(const uint8_t *__restrict__ feats_base, uint64_t buf_stride, SlotMap map, SectionTable tab,
                                                       const uint32_t *__restrict__ found_base, uint32_t found_buf_stride, uint32_t *__restrict__ out,
                                                       PackOffsets offs, uint32_t *__restrict__ found_post)
{
  const uint32_t slot = blockIdx.y;
  const uint32_t bufi = map.buf[slot];
  const uint32_t *feats = (const uint32_t *)(feats_base + (size_t)bufi * buf_stride);
  const uint32_t *found = found_base + (size_t)bufi * found_buf_stride;
  // feature posting: the buffer's counters go to the host mirror with the records (same layout as found_base, mapped pinned memory)
  if (found_post && blockIdx.x == 0 && threadIdx.x < found_buf_stride)
    found_post[(size_t)bufi * found_buf_stride + threadIdx.x] = found[threadIdx.x];
  uint32_t cnt[16];
  uint32_t total = 0;
#pragma unroll
  for (uint32_t o = 0; o < 16; o++)
  {
    uint32_t n = 0;
    if (o < tab.nsec)
    {
      n = found[o];
      n = n < tab.cap[o] ? n : tab.cap[o];
    }
    cnt[o] = n;
    total += n;
  }
  uint32_t *dst = out + (size_t)offs.row[slot] * 41u;
  const uint32_t ndw = total * 41u; // 164-byte records = 41 dwords
  for (uint32_t d = blockIdx.x * 256u + threadIdx.x; d < ndw; d += gridDim.x * 256u)
  {
    const uint32_t row = d / 41u, w = d - row * 41u;
    uint32_t base = 0, src_row = 0;
#pragma unroll
    for (uint32_t o = 0; o < 16; o++)
    {
      if (row >= base && row < base + cnt[o])
        src_row = tab.off[o] + (row - base);
      base += cnt[o];
    }
    dst[d] = feats[(size_t)src_row * 41u + w];
  }
}

// Cross-check + Lowe-ratio filter over the 2-NN records of a forward (A -> B) and, optionally, a reverse (B -> A)
// matching — what every caller of the reference runs on the CPU after vksift_downloadMatches
// (src/examples/test_sift_match.cpp:90-107, src/perf/perf_common.cpp:123-169). One 1024-thread workgroup per pair keeps
// the survivors in increasing idx_a order (ballot + scan compaction, no atomics), 16 B per survivor.
__global__ void __launch_bounds__(1024) k_filter_matches(const uint32_t *__restrict__ fwd, uint64_t fwd_slot_stride, const uint32_t *__restrict__ rev,
                                                         uint64_t rev_slot_stride, const uint32_t *__restrict__ n_fwd, uint32_t n_stride, float ratio,
                                                         uint32_t *__restrict__ out, uint64_t out_slot_stride, uint32_t *__restrict__ out_n)
{
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  const uint32_t slot = blockIdx.x;
  fwd += (size_t)slot * fwd_slot_stride;
  if (rev)
    rev += (size_t)slot * rev_slot_stride;
  out += (size_t)slot * out_slot_stride;
  const uint32_t na = n_fwd[(size_t)slot * n_stride], nb = n_fwd[(size_t)slot * n_stride + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0)
    carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < na; base += 1024)
  {
    const uint32_t i = base + threadIdx.x;
    bool keep = false;
    uint32_t j = 0, d1 = 0, d2 = 0;
    if (i < na)
    {
      const uint32_t *m = fwd + (size_t)i * 5;
      j = m[1], d1 = m[3], d2 = m[4];
      keep = (__uint_as_float(d1) / __uint_as_float(d2)) < ratio;
      if (rev)
      {
        keep = keep && j < nb;
        if (keep)
        {
          const uint32_t *r = rev + (size_t)j * 5;
          keep = r[1] == i && (__uint_as_float(r[3]) / __uint_as_float(r[4])) < ratio;
        }
      }
    }
    const unsigned long long bal = __ballot(keep);
    const uint32_t rank = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0)
      wave_tot[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t wave_base = 0, total = 0;
    for (int wv = 0; wv < 16; wv++)
    {
      if (wv < wave)
        wave_base += wave_tot[wv];
      total += wave_tot[wv];
    }
    const uint32_t carry = carry_s;
    if (keep)
    {
      uint32_t *o = out + (size_t)(carry + wave_base + rank) * 4;
      o[0] = fwd[(size_t)i * 5], o[1] = j, o[2] = d1, o[3] = d2;
    }
    __syncthreads();
    if (threadIdx.x == 0)
      carry_s = carry + total;
    __syncthreads();
  }
  if (threadIdx.x == 0)
    out_n[slot] = carry_s;
}

} // namespace

extern "C"
{
  int vksift_hip_gather_descriptors(const uint8_t *feats, uint32_t n, uint8_t *desc, vksift_hip_stream s)
  {
    if (n == 0)
      return 0;
    uint32_t blocks = (n * 32u + 255u) / 256u;
    hipLaunchKernelGGL(k_gather_desc, dim3(blocks), dim3(256), 0, (hipStream_t)s, feats, n, (uint32_t *)desc);
    return (int)hipGetLastError();
  }

  int vksift_hip_gather_sections(const uint8_t *feats_base, uint64_t buf_stride, const uint32_t *buf_ids, uint32_t nslots, uint32_t nsec,
                                 const uint32_t *sec_off, const uint32_t *sec_cap, const uint32_t *fixed_counts, const uint32_t *found_base,
                                 uint32_t found_buf_stride, uint32_t max_rows, uint32_t pad_rows_to, uint8_t *desc, uint64_t desc_slot_stride,
                                 uint32_t *norms, uint64_t norm_slot_stride, uint32_t *n_out_dev, uint32_t n_slot_stride, vksift_hip_stream s)
  {
    if (nsec > 16 || nslots < 1 || nslots > VKSIFT_HIP_GATHER_SLOTS)
      return (int)hipErrorInvalidValue;
    SectionTable t;
    t.nsec = nsec;
    for (uint32_t o = 0; o < 16; o++)
    {
      t.off[o] = o < nsec ? sec_off[o] : 0u;
      t.cap[o] = o < nsec ? sec_cap[o] : 0u;
      t.fixed[o] = (o < nsec && fixed_counts) ? fixed_counts[o] : 0u;
    }
    GatherMap m;
    for (uint32_t i = 0; i < VKSIFT_HIP_GATHER_SLOTS; i++)
      m.buf[i] = i < nslots ? buf_ids[i] : 0u;
    if (max_rows < pad_rows_to)
      max_rows = pad_rows_to;
    /* grid-stride over the rows that actually exist (count read on the device): ~4096 workgroups per launch whatever the batch — a
     * buffer alone gets up to 256 blocks of 32 rows, 512 buffers 8 each (8 rounds over 1900 rows) */
    uint32_t blocks = (max_rows + 31u) / 32u, cap_blocks = 4096u / nslots;
    cap_blocks = cap_blocks < 8u ? 8u : (cap_blocks > 256u ? 256u : cap_blocks);
    if (blocks > cap_blocks)
      blocks = cap_blocks;
    if (blocks == 0)
      blocks = 1;
    hipLaunchKernelGGL(k_gather_sections, dim3(blocks, nslots), dim3(256), 0, (hipStream_t)s, feats_base, buf_stride, m, t, found_base, found_buf_stride,
                       (uint32_t *)desc, desc_slot_stride / 4, norms, norm_slot_stride, n_out_dev, n_slot_stride, pad_rows_to);
    return (int)hipGetLastError();
  }

  int vksift_hip_pack_features(const uint8_t *feats_base, uint64_t buf_stride, const uint32_t *buf_ids, const uint32_t *out_rows, uint32_t nslots, uint32_t nsec,
                               const uint32_t *sec_off, const uint32_t *sec_cap, const uint32_t *found_base, uint32_t found_buf_stride, uint8_t *out,
                               uint32_t max_rows, uint32_t *found_post, vksift_hip_stream s)
  {
    if (nslots < 1 || nslots > 64 || nsec > 16 || (found_post && found_buf_stride > 256u))
      return (int)hipErrorInvalidValue;
    SectionTable t;
    t.nsec = nsec;
    for (uint32_t o = 0; o < 16; o++)
    {
      t.off[o] = o < nsec ? sec_off[o] : 0u;
      t.cap[o] = o < nsec ? sec_cap[o] : 0u;
      t.fixed[o] = 0u;
    }
    SlotMap m;
    PackOffsets po;
    for (uint32_t i = 0; i < 64; i++)
      m.buf[i] = i < nslots ? buf_ids[i] : 0u, po.row[i] = i < nslots ? out_rows[i] : 0u;
    uint32_t blocks = (uint32_t)(((uint64_t)max_rows * 41u + 1023u) / 1024u); /* four dwords per thread */
    blocks = blocks < 1u ? 1u : (blocks > 128u ? 128u : blocks);
    hipLaunchKernelGGL(k_pack_features, dim3(blocks, nslots), dim3(256), 0, (hipStream_t)s, feats_base, buf_stride, m, t, found_base, found_buf_stride,
                       (uint32_t *)out, po, found_post);
    return (int)hipGetLastError();
  }

  int vksift_hip_filter_matches(const uint8_t *fwd, uint64_t fwd_slot_stride, const uint8_t *rev, uint64_t rev_slot_stride, const uint32_t *n_fwd,
                                uint32_t n_stride, float ratio, uint32_t nslots, uint8_t *out, uint64_t out_slot_stride, uint32_t *out_n, vksift_hip_stream s)
  {
    if (nslots < 1)
      return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_filter_matches, dim3(nslots), dim3(1024), 0, (hipStream_t)s, (const uint32_t *)fwd, fwd_slot_stride / 4, (const uint32_t *)rev,
                       rev_slot_stride / 4, n_fwd, n_stride, ratio, (uint32_t *)out, out_slot_stride / 4, out_n);
    return (int)hipGetLastError();
  }
}
