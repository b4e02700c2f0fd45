// multi.h — one launch for several octaves of a detection.
//
// The reference records the dispatches of ALL octaves of a stage into one command buffer (sift_detector.c:1106-1259); the
// per-octave launch chains this build started with turn that into 11 launches per octave and stage set, most of them far too
// small to fill 256 CUs (octave 3 of a 640x480 frame is 80x60 texels). Here a stage is ONE launch: the workgroups of all
// octaves are laid out back to back in a flat 1-D grid, largest octave first, and every workgroup looks up its octave and its
// position inside that octave's own (virtual) 3-D grid — the grid the single-octave launch would have had, in the same
// x-fastest dispatch order, so everything the kernels derive from the dispatch order (XCD-contiguous ranges, image-fastest
// refinement chunks) holds per octave as before.
#pragma once
#include <stdint.h>
#include <stdlib.h>

#include "vksift_hip.h"

constexpr int MULTI_MAX = 8; // octaves per launch (a 1920x1080 frame with up-sampling has 7); longer lists are cut in runs

template <typename A>
struct Multi
{
  A oct[MULTI_MAX];
  uint32_t start[MULTI_MAX + 1];          // first flat workgroup of entry i; start[n] = grid size
  uint32_t gx[MULTI_MAX], gy[MULTI_MAX];  // virtual grid extents of entry i (x fastest); z follows from its workgroup count
  int n;
};

struct VBlock
{
  int o;                 // entry (octave) this workgroup belongs to — wave-uniform
  uint32_t x, y, z;      // its virtual blockIdx
  uint32_t gx, gy, gz;   // the virtual gridDim
};

template <typename A>
__device__ __forceinline__ VBlock vblock(const Multi<A> &m)
{
  const uint32_t id = blockIdx.x;
  int o = 0;
#pragma unroll
  for (int i = 1; i < MULTI_MAX; i++)
    if (i < m.n && id >= m.start[i])
      o = i;
  VBlock v;
  v.o = o;
  const uint32_t first = m.start[o], cnt = m.start[o + 1] - first, l = id - first;
  v.gx = m.gx[o], v.gy = m.gy[o];
  v.gz = cnt / (v.gx * v.gy);
  v.x = l % v.gx;
  const uint32_t r = l / v.gx;
  v.y = r % v.gy;
  v.z = r / v.gy;
  return v;
}

// host side: octaves per launch (vksift_hip_tune(VKSIFT_TUNE_MULTI_MAX, 1..8) lowers it: tests of the cutting of longer octave lists)
static inline uint32_t multi_run_max()
{
  const int n = vksift_hip_tune_get(VKSIFT_TUNE_MULTI_MAX);
  return (uint32_t)(n >= 1 && n <= MULTI_MAX ? n : MULTI_MAX);
}

// host side: append entry i with its virtual grid; returns false if the flat grid would overflow 2^31 workgroups
template <typename A>
static inline bool multi_add(Multi<A> &m, const A &a, uint32_t gx, uint32_t gy, uint32_t gz)
{
  if (m.n == 0)
    m.start[0] = 0;
  const uint64_t cnt = (uint64_t)gx * gy * gz, end = (uint64_t)m.start[m.n] + cnt;
  if (m.n >= MULTI_MAX || cnt == 0 || end >= 0x7FFFFFFFull)
    return false;
  m.oct[m.n] = a;
  m.gx[m.n] = gx, m.gy[m.n] = gy;
  m.start[m.n + 1] = (uint32_t)end;
  m.n++;
  return true;
}
