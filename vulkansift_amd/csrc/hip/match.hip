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
// sqrt(float(d2)) values, and two different d2 can round to the same float. A lane sees its B columns
// in increasing index order, so the integer test d2 < (current second best d2) is a safe pre-filter
// (sqrt is monotone) and the float comparison is evaluated only on that rare path; cross-lane merges
// compare (sqrtf(d2), index) lexicographically. Quirk Q7 (d(b0) == d(b1) makes index 1 the best) is an
// index-priority swap of columns 0 and 1 for that A row. Quirk Q6 (b[0], b[1] read unconditionally):
// callers pad B to two rows.
//
// MFMA operand layout used (16x16x64 i8): lane l supplies 16 consecutive K bytes (l>>4)*16.. of A row
// (l&15) / B row (l&15); since A and B use the same K slicing any K permutation cancels in the dot
// product. Accumulator: lane l holds column (l&15), rows (l>>4)*4 + r, r = 0..3.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vksift_hip.h"

namespace
{

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int BT = 64;          // B rows staged in LDS per iteration
constexpr int B_STRIDE = 144;   // bytes per staged row (128 + 16 pad: ds_read_b128 of 16 rows hits 16 distinct bank quads)
constexpr uint32_t QMAX = 0xFFFFFFFFu;

__global__ void __launch_bounds__(256) k_gather_desc(const uint8_t *__restrict__ feats, uint32_t n, uint32_t *__restrict__ desc)
{
  uint32_t i = blockIdx.x * 256 + threadIdx.x; // dword index
  if (i >= n * 32u)
    return;
  uint32_t row = i >> 5, j = i & 31u;
  desc[i] = *(const uint32_t *)(feats + (size_t)row * 164 + 36 + 4 * j);
}

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

// Gather the descriptors of a (sectioned or packed) SIFT buffer into dense rows in download order AND compute
// their shifted norms, with the per-section feature counts read on the device (no host round trip):
// row -> section by scanning the <= 16 section counts; one half-wave (32 lanes) per row, one dword per lane.
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

__global__ void __launch_bounds__(256) k_gather_sections(const uint8_t *__restrict__ feats_base, uint64_t buf_stride, SlotMap map, SectionTable tab,
                                                         const uint32_t *__restrict__ found_base, uint32_t found_buf_stride, uint32_t *__restrict__ desc,
                                                         uint64_t desc_slot_stride, uint32_t *__restrict__ norms, uint64_t norm_slot_stride,
                                                         uint32_t *__restrict__ n_out, uint32_t n_slot_stride, uint32_t pad_rows_to)
{
  const uint32_t slot = blockIdx.y;
  const uint32_t bufi = map.buf[slot];
  const uint8_t *feats = feats_base + (size_t)bufi * buf_stride;
  const uint32_t *found = found_base ? found_base + (size_t)bufi * found_buf_stride : nullptr;
  desc += (size_t)slot * desc_slot_stride;
  norms += (size_t)slot * norm_slot_stride;
  n_out += (size_t)slot * n_slot_stride;

  const uint32_t j = threadIdx.x & 31u;
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
  for (uint32_t row = blockIdx.x * 8 + (threadIdx.x >> 5); row < nrows; row += gridDim.x * 8)
  {
    uint32_t v = 0u; // rows in [total, pad_rows_to): quirk Q6 padding, all-zero descriptors
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
      v = *(const uint32_t *)(feats + (size_t)src_row * 164 + 36 + 4 * j);
    }
    desc[(size_t)row * 32 + j] = v;
    uint32_t s2 = __builtin_amdgcn_udot4(v, v, 0u, false);
    uint32_t s1 = __builtin_amdgcn_udot4(v, 0x01010101u, 0u, false);
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1)
    {
      s2 += __shfl_xor(s2, d, 64);
      s1 += __shfl_xor(s1, d, 64);
    }
    if (j == 0)
      norms[row] = s2 - 256u * s1 + 128u * 128u * 128u;
  }
}

// Per-slot strides of a batched matching launch (blockIdx.y = slot); all zero for a single pair.
struct SlotStrides
{
  uint64_t desc_a, desc_b; // dwords
  uint64_t norm_a, norm_b; // u32
  uint64_t matches;        // dwords
  uint32_t n;              // u32 between the {N_A, N_B} pairs
};

struct Top2
{
  uint32_t q1, k1, q2, k2; // squared distances and index keys of best / second
};

// Below 2^22 the map q -> sqrtf(float(q)) is injective (gap 1/(2 sqrt q) > ulp), so integer order == float
// order and the correctly rounded sqrtf (a ~20-instruction sequence) is only needed above it.
constexpr uint32_t Q_EXACT = 1u << 22;

// (sqrtf(q), key) lexicographic order — the order the reference's scan realises
__device__ __forceinline__ bool lex_less(uint32_t qa, uint32_t ka, uint32_t qb, uint32_t kb)
{
  if (qa == qb)
    return ka < kb;
  if ((qa | qb) < Q_EXACT)
    return qa < qb;
  float da = sqrtf((float)qa), db = sqrtf((float)qb);
  return da < db || (da == db && ka < kb);
}

// in-lane insertion; keys arrive in increasing order so a tie never displaces a holder.
// Precondition (pre-filter): q < t.q2 as integers.
__device__ __forceinline__ void insert_seq(Top2 &t, uint32_t q, uint32_t key)
{
  if (t.q2 < Q_EXACT)
  {
    // q < q2 < 2^22 and q1 <= q2: pure integer comparison is exact
    if (q < t.q1)
    {
      t.q2 = t.q1, t.k2 = t.k1;
      t.q1 = q, t.k1 = key;
    }
    else
      t.q2 = q, t.k2 = key;
    return;
  }
  float d = sqrtf((float)q);
  if (d < sqrtf((float)t.q1))
  {
    t.q2 = t.q1, t.k2 = t.k1;
    t.q1 = q, t.k1 = key;
  }
  else if (d < sqrtf((float)t.q2))
  {
    t.q2 = q, t.k2 = key;
  }
}

__device__ __forceinline__ Top2 merge2(const Top2 &a, const Top2 &b)
{
  Top2 r;
  if (lex_less(a.q1, a.k1, b.q1, b.k1))
  {
    r.q1 = a.q1, r.k1 = a.k1;
    if (lex_less(a.q2, a.k2, b.q1, b.k1))
      r.q2 = a.q2, r.k2 = a.k2;
    else
      r.q2 = b.q1, r.k2 = b.k1;
  }
  else
  {
    r.q1 = b.q1, r.k1 = b.k1;
    if (lex_less(b.q2, b.k2, a.q1, a.k1))
      r.q2 = b.q2, r.k2 = b.k2;
    else
      r.q2 = a.q1, r.k2 = a.k1;
  }
  return r;
}

// AT = 16-row A tiles per wave. Block = 4 waves = 64*AT A rows; B streams through LDS.
template <int AT>
__global__ void __launch_bounds__(256) k_match_mfma(const uint32_t *__restrict__ desc_a, const uint32_t *__restrict__ norm_a, uint32_t na,
                                                    uint32_t a_index_base, const uint32_t *__restrict__ desc_b, const uint32_t *__restrict__ norm_b,
                                                    uint32_t nb, uint32_t *__restrict__ matches, const uint32_t *__restrict__ n_dev,
                                                    uint32_t na_lo, uint32_t na_hi, SlotStrides ss)
{
  desc_a += (size_t)blockIdx.y * ss.desc_a, desc_b += (size_t)blockIdx.y * ss.desc_b;
  norm_a += (size_t)blockIdx.y * ss.norm_a, norm_b += (size_t)blockIdx.y * ss.norm_b;
  matches += (size_t)blockIdx.y * ss.matches;
  if (n_dev)
  {
    n_dev += (size_t)blockIdx.y * ss.n;
    // asynchronous path: the row counts were produced on the device by k_gather_sections; this instantiation only
    // serves na in (na_lo, na_hi] (the host launches one kernel per regime, the others exit here)
    na = n_dev[0];
    nb = n_dev[1] < 2u ? 2u : n_dev[1];
    if (na <= na_lo || na > na_hi)
      return;
  }
  __shared__ __attribute__((aligned(16))) uint8_t s_b[BT * B_STRIDE];
  __shared__ uint32_t s_nb[BT];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, grp = lane >> 4;
  // the grid may be smaller than the number of 64*AT-row blocks (bounded launch): loop over row blocks
  for (uint32_t rb = blockIdx.x; rb * (64u * AT) < na; rb += gridDim.x)
  {
  const uint32_t row_base = (rb * 4 + wave) * (16 * AT);

  // A fragments (XOR 0x80 -> int8) and norms of the rows this lane accumulates
  v4i afrag[AT][2];
  uint32_t an[AT][4];
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
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
      uint32_t rr = row_base + t * 16 + grp * 4 + j;
      an[t][j] = norm_a[rr < na ? rr : na - 1];
    }
  }

  Top2 st[AT][4];
#pragma unroll
  for (int t = 0; t < AT; t++)
#pragma unroll
    for (int j = 0; j < 4; j++)
      st[t][j] = Top2{QMAX, QMAX, QMAX, QMAX};
  uint32_t swap_bits = 0; // bit (t*4+j): d(b0) == d(b1) for that A row (quirk Q7)

  // B tiles are prefetched one tile ahead into registers (2 x 16 B per thread) so that the global-load latency of
  // tile t+1 hides behind the MFMA + epilogue work of tile t.
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

  for (uint32_t t0 = 0; t0 < nb; t0 += BT)
  {
    __syncthreads();
    // stage the prefetched BT rows (zero beyond nb), converting to int8
#pragma unroll
    for (int q = 0; q < NLD; q++)
    {
      int i = threadIdx.x + q * 256;
      int r = i >> 3, c = i & 7;
      uint4 v = pfb[q];
      v.x ^= 0x80808080u, v.y ^= 0x80808080u, v.z ^= 0x80808080u, v.w ^= 0x80808080u;
      *(uint4 *)(s_b + r * B_STRIDE + c * 16) = v;
    }
    if (threadIdx.x < BT)
      s_nb[threadIdx.x] = pfn;
    __syncthreads();
    if (t0 + BT < nb)
      fetch_tile(t0 + BT);

#pragma unroll
    for (int sub = 0; sub < BT / 16; sub++)
    {
      const uint32_t bcol = t0 + sub * 16 + col; // B index this lane's outputs belong to
      if (t0 + sub * 16 >= nb)
        break;
      const uint8_t *pb = s_b + (sub * 16 + col) * B_STRIDE + grp * 16;
      const v4i b0 = *(const v4i *)pb;
      const v4i b1 = *(const v4i *)(pb + 64);
      const uint32_t bn = s_nb[sub * 16 + col];
      const bool first = (t0 == 0 && sub == 0);
#pragma unroll
      for (int t = 0; t < AT; t++)
      {
        v4i acc = v4i{0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[t][0], b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[t][1], b1, acc, 0, 0, 0);
        uint32_t q[4];
        bool any = false;
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
          q[j] = bcol < nb ? an[t][j] + bn - 2u * (uint32_t)acc[j] : QMAX;
          any = any || (q[j] < st[t][j].q2);
        }
        if (first)
        {
          // quirk Q7: exchange d2(b0) / d2(b1) between the col-0 and col-1 lanes of each row group
#pragma unroll
          for (int j = 0; j < 4; j++)
          {
            uint32_t other = __shfl_xor(q[j], 1, 64);
            bool sw = col < 2 && sqrtf((float)q[j]) == sqrtf((float)other);
            if (sw)
              swap_bits |= 1u << (t * 4 + j);
            uint32_t key = (col < 2 && sw) ? (uint32_t)(col ^ 1) : bcol;
            if (q[j] != QMAX)
              insert_seq(st[t][j], q[j], key);
          }
        }
        else if (any)
        {
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (q[j] < st[t][j].q2)
              insert_seq(st[t][j], q[j], bcol);
        }
      }
    }
  }

  // merge the 16 lanes that share A rows (butterfly over the column bits), then lane col==0 writes
#pragma unroll
  for (int t = 0; t < AT; t++)
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
      Top2 s = st[t][j];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1)
      {
        Top2 o;
        o.q1 = __shfl_xor(s.q1, m, 64), o.k1 = __shfl_xor(s.k1, m, 64);
        o.q2 = __shfl_xor(s.q2, m, 64), o.k2 = __shfl_xor(s.k2, m, 64);
        s = merge2(s, o);
      }
      uint32_t r = row_base + t * 16 + grp * 4 + j;
      if (col == 0 && r < na)
      {
        bool sw = (swap_bits >> (t * 4 + j)) & 1u;
        uint32_t i1 = (sw && s.k1 < 2) ? (s.k1 ^ 1u) : s.k1;
        uint32_t i2 = (sw && s.k2 < 2) ? (s.k2 ^ 1u) : s.k2;
        uint32_t *m = matches + (size_t)r * 5;
        m[0] = a_index_base + r;
        m[1] = i1;
        m[2] = i2;
        m[3] = __float_as_uint(sqrtf((float)s.q1));
        m[4] = __float_as_uint(sqrtf((float)s.q2));
      }
    }
  } // row-block loop
}

// Small-problem variant: one workgroup = 16 A rows; its 4 waves each take one 16-row slice of every staged 64-row B
// tile (wave w sees B indices t0 + 16w + col, increasing over tiles, so the in-lane pre-filter stays valid), and the
// four partial top-2 lists are merged through LDS with the same (sqrt(d2), index) order. 4x more waves than the
// row-per-wave kernel for a few thousand features, 4x shorter dependent chain per wave.
__global__ void __launch_bounds__(256) k_match_mfma_split(const uint32_t *__restrict__ desc_a, const uint32_t *__restrict__ norm_a, uint32_t na,
                                                          uint32_t a_index_base, const uint32_t *__restrict__ desc_b,
                                                          const uint32_t *__restrict__ norm_b, uint32_t nb, uint32_t *__restrict__ matches,
                                                          const uint32_t *__restrict__ n_dev, uint32_t na_lo, uint32_t na_hi, SlotStrides ss)
{
  desc_a += (size_t)blockIdx.y * ss.desc_a, desc_b += (size_t)blockIdx.y * ss.desc_b;
  norm_a += (size_t)blockIdx.y * ss.norm_a, norm_b += (size_t)blockIdx.y * ss.norm_b;
  matches += (size_t)blockIdx.y * ss.matches;
  if (n_dev)
  {
    n_dev += (size_t)blockIdx.y * ss.n;
    na = n_dev[0];
    nb = n_dev[1] < 2u ? 2u : n_dev[1];
    if (na <= na_lo || na > na_hi)
      return;
  }
  if (blockIdx.x * 16u >= na)
    return;
  __shared__ __attribute__((aligned(16))) uint8_t s_b[2][BT * B_STRIDE];
  __shared__ uint32_t s_nb[2][BT];
  __shared__ uint32_t s_part[4][16][4];
  __shared__ uint32_t s_swap;

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
  uint32_t swap_bits = 0;

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
      uint32_t q[4];
      bool any = false;
#pragma unroll
      for (int j = 0; j < 4; j++)
      {
        q[j] = bcol < nb ? an[j] + bn - 2u * (uint32_t)acc[j] : QMAX;
        any = any || (q[j] < st[j].q2);
      }
      if (sub0 == 0)
      {
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
          uint32_t other = __shfl_xor(q[j], 1, 64);
          bool sw = col < 2 && sqrtf((float)q[j]) == sqrtf((float)other);
          if (sw)
            swap_bits |= 1u << j;
          uint32_t key = (col < 2 && sw) ? (uint32_t)(col ^ 1) : bcol;
          if (q[j] != QMAX)
            insert_seq(st[j], q[j], key);
        }
      }
      else if (any)
      {
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (q[j] < st[j].q2)
            insert_seq(st[j], q[j], bcol);
      }
    }
  }

  // merge across the 16 lanes of a row group, then across the 4 waves
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
  if (threadIdx.x == 0)
    s_swap = 0;
  __syncthreads();
  if (wave == 0 && col == 0)
    atomicOr(&s_swap, swap_bits << (grp * 4));
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
      bool sw = (s_swap >> rr) & 1u;
      uint32_t i1 = (sw && s.k1 < 2) ? (s.k1 ^ 1u) : s.k1;
      uint32_t i2 = (sw && s.k2 < 2) ? (s.k2 ^ 1u) : s.k2;
      uint32_t *m = matches + (size_t)r * 5;
      m[0] = a_index_base + r;
      m[1] = i1;
      m[2] = i2;
      m[3] = __float_as_uint(sqrtf((float)s.q1));
      m[4] = __float_as_uint(sqrtf((float)s.q2));
    }
  }
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

  int vksift_hip_match_2nn_desc(const uint8_t *desc_a, uint32_t na, uint32_t a_index_base, const uint8_t *desc_b, uint32_t nb, uint32_t *norm_scratch,
                                uint8_t *matches, vksift_hip_stream s)
  {
    if (na == 0)
      return 0;
    if (nb < 2)
      return (int)hipErrorInvalidValue; /* callers pad B to two rows (quirk Q6) */
    uint32_t *norm_a = norm_scratch, *norm_b = norm_scratch + na;
    hipLaunchKernelGGL(k_shifted_norms, dim3((na + 255u) / 256u), dim3(256), 0, (hipStream_t)s, (const uint32_t *)desc_a, na, norm_a);
    hipLaunchKernelGGL(k_shifted_norms, dim3((nb + 255u) / 256u), dim3(256), 0, (hipStream_t)s, (const uint32_t *)desc_b, nb, norm_b);
    /* Small problems: 16 A rows per wave to fill more CUs; large ones: 64 rows per wave for B-tile reuse. */
    if (na <= 8192u)
    {
      hipLaunchKernelGGL(k_match_mfma_split, dim3((na + 15u) / 16u), dim3(256), 0, (hipStream_t)s, (const uint32_t *)desc_a, norm_a, na, a_index_base,
                         (const uint32_t *)desc_b, norm_b, nb, (uint32_t *)matches, (const uint32_t *)nullptr, 0u, 0xFFFFFFFFu, SlotStrides{0, 0, 0, 0, 0, 0});
    }
    else if (na <= 32768u)
    {
      uint32_t blocks = (na + 63u) / 64u;
      hipLaunchKernelGGL(k_match_mfma<1>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const uint32_t *)desc_a, norm_a, na, a_index_base,
                         (const uint32_t *)desc_b, norm_b, nb, (uint32_t *)matches, (const uint32_t *)nullptr, 0u, 0xFFFFFFFFu, SlotStrides{0, 0, 0, 0, 0, 0});
    }
    else
    {
      uint32_t blocks = (na + 255u) / 256u;
      hipLaunchKernelGGL(k_match_mfma<4>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const uint32_t *)desc_a, norm_a, na, a_index_base,
                         (const uint32_t *)desc_b, norm_b, nb, (uint32_t *)matches, (const uint32_t *)nullptr, 0u, 0xFFFFFFFFu, SlotStrides{0, 0, 0, 0, 0, 0});
    }
    return (int)hipGetLastError();
  }

  int vksift_hip_gather_sections(const uint8_t *feats_base, uint64_t buf_stride, const uint32_t *buf_ids, uint32_t nslots, uint32_t nsec,
                                 const uint32_t *sec_off, const uint32_t *sec_cap, const uint32_t *fixed_counts, const uint32_t *found_base,
                                 uint32_t found_buf_stride, uint32_t max_rows, uint32_t pad_rows_to, uint8_t *desc, uint64_t desc_slot_stride,
                                 uint32_t *norms, uint64_t norm_slot_stride, uint32_t *n_out_dev, uint32_t n_slot_stride, vksift_hip_stream s)
  {
    if (nsec > 16 || nslots < 1 || nslots > 64)
      return (int)hipErrorInvalidValue;
    SectionTable t;
    t.nsec = nsec;
    for (uint32_t o = 0; o < 16; o++)
    {
      t.off[o] = o < nsec ? sec_off[o] : 0u;
      t.cap[o] = o < nsec ? sec_cap[o] : 0u;
      t.fixed[o] = (o < nsec && fixed_counts) ? fixed_counts[o] : 0u;
    }
    SlotMap m;
    for (uint32_t i = 0; i < 64; i++)
      m.buf[i] = i < nslots ? buf_ids[i] : 0u;
    if (max_rows < pad_rows_to)
      max_rows = pad_rows_to;
    uint32_t blocks = (max_rows + 7u) / 8u;
    if (blocks > 256u)
      blocks = 256u; /* grid-stride over the rows that actually exist (count read on the device) */
    if (blocks == 0)
      blocks = 1;
    hipLaunchKernelGGL(k_gather_sections, dim3(blocks, nslots), dim3(256), 0, (hipStream_t)s, feats_base, buf_stride, m, t, found_base, found_buf_stride,
                       (uint32_t *)desc, desc_slot_stride / 4, norms, norm_slot_stride, n_out_dev, n_slot_stride, pad_rows_to);
    return (int)hipGetLastError();
  }

  int vksift_hip_match_2nn_async(const uint8_t *desc_a, const uint32_t *norm_a, uint32_t max_na, const uint8_t *desc_b, const uint32_t *norm_b,
                                 const uint32_t *n_dev, uint8_t *matches, uint32_t nslots, uint64_t desc_slot_stride, uint64_t norm_slot_stride,
                                 uint64_t match_slot_stride, uint32_t n_slot_stride, vksift_hip_stream s)
  {
    if (max_na == 0)
      return 0;
    if (nslots < 1)
      return (int)hipErrorInvalidValue;
    SlotStrides ss;
    ss.desc_a = ss.desc_b = desc_slot_stride / 4;
    ss.norm_a = ss.norm_b = norm_slot_stride;
    ss.matches = match_slot_stride / 4;
    ss.n = n_slot_stride;
    /* The row count is only known on the device: launch for the capacity (surplus workgroups exit at once), one
     * kernel per size regime, each of which returns immediately unless N_A falls in its range:
     *   N_A <= 8192        B-split kernel, 16 A rows per workgroup (keeps a few thousand rows busy on every CU)
     *   8192 < N_A <= 32768 16 A rows per wave
     *   N_A > 32768         64 A rows per wave (B tile reuse) */
    const uint32_t S1 = 8192u, S2 = 32768u;
    hipStream_t hs = (hipStream_t)s;
    /* regimes 2/3 loop over their row blocks, so their grids stay small even when only the capacity is known */
    auto bounded = [](uint32_t blocks, uint32_t slots) { uint32_t lim = slots >= 8 ? 64u : 1024u; return blocks < lim ? blocks : lim; };
    const uint32_t n1 = max_na < S1 ? max_na : S1;
    hipLaunchKernelGGL(k_match_mfma_split, dim3((n1 + 15u) / 16u, nslots), dim3(256), 0, hs, (const uint32_t *)desc_a, norm_a, 0u, 0u,
                       (const uint32_t *)desc_b, norm_b, 0u, (uint32_t *)matches, n_dev, 0u, S1, ss);
    if (max_na > S1)
    {
      const uint32_t n2 = max_na < S2 ? max_na : S2;
      hipLaunchKernelGGL(k_match_mfma<1>, dim3(bounded((n2 + 63u) / 64u, nslots), nslots), dim3(256), 0, hs, (const uint32_t *)desc_a, norm_a, 0u, 0u,
                         (const uint32_t *)desc_b, norm_b, 0u, (uint32_t *)matches, n_dev, S1, S2, ss);
    }
    if (max_na > S2)
      hipLaunchKernelGGL(k_match_mfma<4>, dim3(bounded((max_na + 255u) / 256u, nslots), nslots), dim3(256), 0, hs, (const uint32_t *)desc_a, norm_a, 0u, 0u,
                         (const uint32_t *)desc_b, norm_b, 0u, (uint32_t *)matches, n_dev, S2, 0xFFFFFFFFu, ss);
    return (int)hipGetLastError();
  }
}
