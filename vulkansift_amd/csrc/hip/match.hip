// match.hip — brute-force 2-nearest-neighbour descriptor matcher (gfx950).
//
// Replaces Get2NearestNeighbors.comp (dispatch sift_matcher.c:246-279): for every row of A the two
// closest rows of B under the L2 distance of the 128 uint8 descriptor bytes.
//
// Exactness: squared distances are computed in integers, d2 = |a|^2 + |b|^2 - 2 a.b with
// v_dot4_u32_u8 (all terms < 2^24, exact). The reference compares sqrt(float(d2)) values with
// strict '<' while scanning B in index order; integer d2 order equals float sqrt order except when
// two different d2 round to the same float, so the float comparison is evaluated (with a correctly
// rounded sqrtf) only when the integer test says "closer than the current second best" — rare,
// O(log nb) times per row — which reproduces the reference's choice bit for bit (quirk Q8), as well
// as the unconditional b[0]/b[1] initialisation (Q6) and its tie rule (Q7).
//
// Layout: descriptors are first gathered from the 164-byte feature records into dense 128-byte rows
// (16-byte aligned) so that A rows load as 8 x dwordx4 and B tiles stream through LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vksift_hip.h"

namespace
{

constexpr int B_TILE = 128; // B rows per LDS tile (16 KiB)

__global__ void __launch_bounds__(256) k_gather_desc(const uint8_t *__restrict__ feats, uint32_t n, uint32_t *__restrict__ desc)
{
  uint32_t i = blockIdx.x * 256 + threadIdx.x; // dword index
  if (i >= n * 32u)
    return;
  uint32_t row = i >> 5, j = i & 31u;
  desc[i] = *(const uint32_t *)(feats + (size_t)row * 164 + 36 + 4 * j);
}

__global__ void __launch_bounds__(256) k_zero_rows(uint32_t *desc, uint32_t first_row, uint32_t nrows)
{
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < nrows * 32u)
    desc[(size_t)first_row * 32 + i] = 0u;
}

struct Best2
{
  float d1, d2f;  // float distances of best / second
  uint32_t i1, i2;
  uint32_t q1, q2; // their integer squared distances
};

__device__ __forceinline__ void consider(Best2 &r, uint32_t q, uint32_t idx)
{
  if (q < r.q2)
  {
    float d = sqrtf((float)q);
    if (d < r.d1)
    {
      r.d2f = r.d1, r.i2 = r.i1, r.q2 = r.q1;
      r.d1 = d, r.i1 = idx, r.q1 = q;
    }
    else if (d < r.d2f)
    {
      r.d2f = d, r.i2 = idx, r.q2 = q;
    }
  }
}

// One thread per A row; B streamed through LDS in tiles, every lane reads the same B row (LDS
// broadcast). nb_eff = max(nb, 2): rows beyond nb are zero-filled by the host wrapper (Q6).
__global__ void __launch_bounds__(256) k_match_2nn(const uint32_t *__restrict__ desc_a, uint32_t na, uint32_t a_index_base,
                                                   const uint32_t *__restrict__ desc_b, uint32_t nb, uint32_t *__restrict__ matches)
{
  __shared__ uint4 s_b[B_TILE * 8];
  __shared__ uint32_t s_nb2[B_TILE];
  const uint32_t row = blockIdx.x * 256 + threadIdx.x;
  const bool active = row < na;

  uint32_t a[32];
  {
    const uint4 *pa = (const uint4 *)(desc_a + (size_t)(active ? row : 0) * 32);
#pragma unroll
    for (int j = 0; j < 8; j++)
    {
      uint4 v = pa[j];
      a[4 * j + 0] = v.x, a[4 * j + 1] = v.y, a[4 * j + 2] = v.z, a[4 * j + 3] = v.w;
    }
  }
  uint32_t na2 = 0;
#pragma unroll
  for (int j = 0; j < 32; j++)
    na2 = __builtin_amdgcn_udot4(a[j], a[j], na2, false);

  Best2 r;
  r.d1 = r.d2f = 0.f;
  r.i1 = r.i2 = 0;
  r.q1 = r.q2 = 0;

  for (uint32_t t0 = 0; t0 < nb; t0 += B_TILE)
  {
    const uint32_t rows = nb - t0 < (uint32_t)B_TILE ? nb - t0 : (uint32_t)B_TILE;
    __syncthreads();
    // stage tile: rows*8 uint4, coalesced
    for (uint32_t i = threadIdx.x; i < rows * 8; i += 256)
      s_b[i] = ((const uint4 *)(desc_b + (size_t)t0 * 32))[i];
    __syncthreads();
    // row norms: 2 threads per row would conflict on banks; use one thread per (row) with a skewed walk
    if (threadIdx.x < rows)
    {
      uint32_t acc = 0;
#pragma unroll
      for (int j = 0; j < 8; j++)
      {
        uint4 v = s_b[threadIdx.x * 8 + ((j + threadIdx.x) & 7)];
        acc = __builtin_amdgcn_udot4(v.x, v.x, acc, false);
        acc = __builtin_amdgcn_udot4(v.y, v.y, acc, false);
        acc = __builtin_amdgcn_udot4(v.z, v.z, acc, false);
        acc = __builtin_amdgcn_udot4(v.w, v.w, acc, false);
      }
      s_nb2[threadIdx.x] = acc;
    }
    __syncthreads();

    for (uint32_t bi = 0; bi < rows; bi++)
    {
      uint32_t dot = 0;
#pragma unroll
      for (int j = 0; j < 8; j++)
      {
        uint4 v = s_b[bi * 8 + j];
        dot = __builtin_amdgcn_udot4(a[4 * j + 0], v.x, dot, false);
        dot = __builtin_amdgcn_udot4(a[4 * j + 1], v.y, dot, false);
        dot = __builtin_amdgcn_udot4(a[4 * j + 2], v.z, dot, false);
        dot = __builtin_amdgcn_udot4(a[4 * j + 3], v.w, dot, false);
      }
      const uint32_t q = na2 + s_nb2[bi] - 2u * dot;
      const uint32_t gb = t0 + bi;
      if (gb >= 2)
        consider(r, q, gb);
      else if (gb == 0)
      {
        r.q1 = q, r.d1 = sqrtf((float)q), r.i1 = 0; // provisional: holds b[0] until b[1] is seen
      }
      else
      {
        // Get2NearestNeighbors.comp:66-80
        float d0 = r.d1, d1 = sqrtf((float)q);
        uint32_t q0 = r.q1;
        if (d0 < d1)
        {
          r.d1 = d0, r.i1 = 0, r.q1 = q0;
          r.d2f = d1, r.i2 = 1, r.q2 = q;
        }
        else
        {
          r.d1 = d1, r.i1 = 1, r.q1 = q;
          r.d2f = d0, r.i2 = 0, r.q2 = q0;
        }
      }
    }
  }
  if (active)
  {
    uint32_t *m = matches + (size_t)row * 5;
    m[0] = a_index_base + row;
    m[1] = r.i1;
    m[2] = r.i2;
    m[3] = __float_as_uint(r.d1);
    m[4] = __float_as_uint(r.d2f);
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

  int vksift_hip_match_2nn_desc(const uint8_t *desc_a, uint32_t na, uint32_t a_index_base, const uint8_t *desc_b, uint32_t nb, uint8_t *matches,
                                vksift_hip_stream s)
  {
    if (na == 0)
      return 0;
    if (nb < 2)
      return (int)hipErrorInvalidValue; /* callers pad B to two rows (quirk Q6) */
    uint32_t blocks = (na + 255u) / 256u;
    hipLaunchKernelGGL(k_match_2nn, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const uint32_t *)desc_a, na, a_index_base, (const uint32_t *)desc_b, nb,
                       (uint32_t *)matches);
    return (int)hipGetLastError();
  }

  int vksift_hip_match_2nn(const uint8_t *feats_a, uint32_t na, const uint8_t *feats_b, uint32_t nb, uint8_t *desc_a, uint8_t *desc_b, uint8_t *matches,
                           vksift_hip_stream s)
  {
    if (na == 0)
      return 0;
    int e = vksift_hip_gather_descriptors(feats_a, na, desc_a, s);
    if (e)
      return e;
    e = vksift_hip_gather_descriptors(feats_b, nb, desc_b, s);
    if (e)
      return e;
    uint32_t nb_eff = nb;
    if (nb < 2)
    {
      /* The shader reads b[0] and b[1] unconditionally (stale memory when nb < 2, quirk Q6); this build
       * defines the missing rows as all-zero descriptors. */
      hipLaunchKernelGGL(k_zero_rows, dim3(1), dim3(256), 0, (hipStream_t)s, (uint32_t *)desc_b, nb, 2u - nb);
      nb_eff = 2;
    }
    return vksift_hip_match_2nn_desc(desc_a, na, 0u, desc_b, nb_eff, matches, s);
  }
}
