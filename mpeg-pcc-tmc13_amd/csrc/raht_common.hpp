// raht_common.hpp -- device-side data layout shared by the RAHT kernels.
//
// Layout in HBM (one batch = S slices laid back to back, N points):
//
//   points      pos[N] int64 (ascending per slice), attrs[N][C] int32
//   level li    nodes of the octree level `li` (key = pos >> 3*li) of ALL
//               slices, slice after slice, Morton order inside a slice:
//                 key[li][M]    int64   pos >> 3*li
//                 fp [li][M+1]  int32   first point of the node
//                               (weight = fp[j+1] - fp[j])
//                 fc [li][M+1]  int32   first child in level li-1
//                 soff[li][S+1] int32   first node of each slice
//   P[N+1][C]   exclusive prefix sums of the attributes (int32, modular):
//               a node's attribute sum is P[fp[j+1]] - P[fp[j]]
//   rec*/nn*    reconstruction ping-pong buffers indexed by
//               (slice point offset + slice-local node index)
//
// The reference keeps one std::vector<UrahtNode> (40 B per node) that is
// repeatedly compacted and re-expanded (tmc3/RAHT.cpp:95-264); here each
// level exists once, as 16 B of index data per node.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gpcc_attr_mi355.h"
#include "gpcc_primitives.hpp"

namespace gpcc {

constexpr int kMaxLevels = 22;   // ceil(63 / 3) + root
// points per wavefront-tile of the tree build (a multiple of 64)
#ifndef GPCC_TILE_POINTS
#define GPCC_TILE_POINTS 1024
#endif
constexpr int kTilePoints = GPCC_TILE_POINTS;
constexpr int kWave = 64;

// value slots of the compact level pass (cx_tree.hpp): hold = slot | top level << 27
constexpr int kCxSlotBits = 27;
constexpr uint32_t kCxSlotMask = (1u << kCxSlotBits) - 1;
constexpr int32_t kCxMaxPoints = 1 << (kCxSlotBits - 1);

// payload of one 16-byte buffer load / store (mailbox granules)
#ifdef GPCC_EMU  // (tests/emu: a plain struct with the same layout and member names)
typedef emu_u32x4 u32x4;
#else
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#endif

struct TreeView {
  int32_t nlev;         // level arrays in use (top one has 1 node / slice)
  int32_t num_slices;
  int32_t n_total;
  int32_t num_tiles;
  const int64_t* pos;
  const int32_t* pt_off;  // [S+1] device copy of the slice point offsets
  int64_t* key[kMaxLevels];
  int32_t* fp[kMaxLevels];
  int32_t* fc[kMaxLevels];
  int32_t* soff[kMaxLevels];
  int32_t cap[kMaxLevels];  // nodes the arrays of each level were sized for
  int32_t* error;           // the context's sticky error word (0 = fine)
};

// A kernel of a call whose tree does not fit its arrays (or of any call behind
// an unreported failure) must not touch memory through that tree's indices:
// every kernel behind tree_scan leaves at once while the error word is set.
__device__ __forceinline__ bool
tree_failed(const TreeView& tv)
{
  return __hip_atomic_load(tv.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// Per (slice, level) processing schedule, derived on the device from the
// node counts (tmc3/RAHT.cpp:1165-1265: which levels run the block loop,
// which quantisation layer they use, where their coefficients start).
struct LevelSched {
  int32_t coeff_base;  // first coefficient of the level, slice relative
  uint8_t processed;
  uint8_t is_root;     // first processed level: DC is coded, no prediction
  uint8_t qp_layer;
  int8_t ac_layer;
  uint8_t parity;      // reconstruction buffer written by this level
  uint8_t coarse;      // processed inside raht_coarse_kernel (raht_tile.hpp), not by a level launch
  uint8_t pad[2];
};

// What the host needs from the tree to size and skip the level launches,
// written by schedule_kernel into pinned host memory (the host waits for an
// event recorded right behind it while the coarse kernel runs).
struct TreeStats {
  int32_t fine_levels;       // levels li < fine_levels still need a launch for some slice
  int32_t max_top;           // largest top_level of a slice
  int32_t nodes[kMaxLevels]; // nodes per level, all slices
};

struct SliceSched {
  int32_t num_unique;
  int32_t top_level;       // smallest li with one node
  int32_t final_qp_layer;  // qpLayer when the level loop ends
  int32_t final_parity;    // buffer holding the leaf reconstruction
  LevelSched lvl[kMaxLevels];
};

// slice of a node / point index: off[] is ascending with off[0] == 0
__device__ __forceinline__ int
find_slice(const int32_t* __restrict__ off, int num_slices, int idx)
{
  int lo = 0, hi = num_slices;  // answer in [lo, hi)
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (off[mid] <= idx)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// ---- wave / 8-lane group helpers --------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// Exchange with lane ^ mask inside an 8-lane group (mask 1, 2 or 4) as DPP
// moves: quad permutes, and lane ^ 4 = 7 - (lane ^ 3) = half-row mirror then
// quad reverse.  A few cycles each, where __shfl_xor compiles to
// ds_bpermute_b32 -- an LDS-crossbar round trip (~100 cycles with its
// s_waitcnt) that would sit on every hop of the dependency chains.  `mask`
// folds to a constant once the caller's stage loop is unrolled.
__device__ __forceinline__ int
lane_xor8(int v, int mask)
{
  if (mask == 1)
    return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);  // quad_perm:[1,0,3,2]
  if (mask == 2)
    return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);  // quad_perm:[2,3,0,1]
  if (mask == 4) {
    const int m = __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    return __builtin_amdgcn_update_dpp(m, m, 0x1B, 0xF, 0xF, false);          // quad_perm:[3,2,1,0]
  }
  return __shfl_xor(v, mask);
}

__device__ __forceinline__ int64_t
shfl_xor_i64(int64_t v, int mask)
{
  int lo = lane_xor8((int)(uint32_t)v, mask);
  int hi = lane_xor8((int)(v >> 32), mask);
  return ((int64_t)hi << 32) | (uint32_t)lo;
}

__device__ __forceinline__ int64_t
shfl_i64(int64_t v, int src)
{
  int lo = __shfl((int)(uint32_t)v, src);
  int hi = __shfl((int)(v >> 32), src);
  return ((int64_t)hi << 32) | (uint32_t)lo;
}

// the same for a value of either arithmetic back end (raht_arith.hpp): int64_t or double
template<class T>
__device__ __forceinline__ T
shfl_xor_v(T v, int mask)
{
  static_assert(sizeof(T) == 8, "64-bit values");
  return __builtin_bit_cast(T, shfl_xor_i64(__builtin_bit_cast(int64_t, v), mask));
}
template<class T>
__device__ __forceinline__ T
shfl_v(T v, int src)
{
  static_assert(sizeof(T) == 8, "64-bit values");
  return __builtin_bit_cast(T, shfl_i64(__builtin_bit_cast(int64_t, v), src));
}

// OR / sum / max over the 8 lanes of a group (lanes g*8 .. g*8+7)
__device__ __forceinline__ uint32_t
group8_or(uint32_t v)
{
  v |= (uint32_t)lane_xor8((int)v, 1);
  v |= (uint32_t)lane_xor8((int)v, 2);
  v |= (uint32_t)lane_xor8((int)v, 4);
  return v;
}
__device__ __forceinline__ int
group8_sum(int v)
{
  v += lane_xor8(v, 1);
  v += lane_xor8(v, 2);
  v += lane_xor8(v, 4);
  return v;
}
__device__ __forceinline__ int
group8_max(int v)
{
#pragma unroll
  for (int d = 1; d < 8; d <<= 1) {
    const int o = lane_xor8(v, d);
    v = o > v ? o : v;
  }
  return v;
}
// bit u = p of lane (group base | u): one compare into an SGPR pair and a
// shift.  The lanes of a group are active together wherever this is used.
__device__ __forceinline__ uint32_t
group8_bits(bool p)
{
  return (uint32_t)(__ballot(p) >> (lane_id() & 56)) & 0xffu;
}
__device__ __forceinline__ bool
group8_any(bool p)
{
  return group8_bits(p) != 0;
}

// value of lane `src` (wave-uniform) in every lane: v_readlane, where __shfl is an
// LDS-crossbar permute
__device__ __forceinline__ int
wave_bcast(int v, int src)
{
  return __builtin_amdgcn_readlane(v, src);
}

__device__ __forceinline__ uint32_t
wave_incl_scan_u32(uint32_t v)
{
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    uint32_t o = __shfl_up(v, d);
    if (lane >= d)
      v += o;
  }
  return v;
}

// XCD-aware work split: workgroup b runs on XCD b % 8 (observed dispatch,
// used for L2 locality only).  Give every XCD one contiguous eighth of the
// work so neighbouring nodes are looked up in the same L2.
__device__ __forceinline__ void
xcd_chunk(int64_t total, int64_t* begin, int64_t* end)
{
  const int nb = gridDim.x;
  const int b = blockIdx.x;
  const int per_xcd = (nb + 7) >> 3;
  const int slot = (b & 7) * per_xcd + (b >> 3);  // position in XCD-major order
  const int64_t chunk = (total + (int64_t)per_xcd * 8 - 1) / ((int64_t)per_xcd * 8);
  int64_t s = (int64_t)slot * chunk;
  int64_t e = s + chunk;
  *begin = s < total ? s : total;
  *end = e < total ? e : total;
}

}  // namespace gpcc
