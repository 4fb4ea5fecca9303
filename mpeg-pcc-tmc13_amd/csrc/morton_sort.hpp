// morton_sort.hpp -- Morton encode + stable LSD radix sort of (code, index).
//
// Replaces the prologue of encode/decode{Colors,Reflectances}TransformRaht
// and of buildPredictorsFast (tmc3/AttributeEncoder.cpp:1225-1229,
// 1316-1321; AttributeDecoder.cpp:538-542, 624-628; PCCTMC3Common.h:2323):
// mortonAddr per point, then std::sort of MortonCodeWithIndex whose
// operator< orders by code and breaks ties by original index
// (PCCTMC3Common.h:184-190).  A STABLE sort of the codes started from the
// identity permutation gives exactly that order, so no tie handling exists.
//
// 8-bit digits, only as many passes as the codes have significant bits.
// Each pass: per-tile digit histograms (one workgroup per tile of 4096
// keys) -> one scan kernel over the [digit][tile] table -> scatter with
// tile-local stable ranks (wave ballots per digit bit).  Slices of a batch
// are sorted independently: the slice index is the most significant digit
// of the scatter key, so one launch sequence serves the whole batch.
#pragma once

#include "raht_common.hpp"

namespace gpcc {

constexpr int kSortTile = 4096;    // keys per workgroup
constexpr int kSortThreads = 256;
constexpr int kSortPerThread = kSortTile / kSortThreads;  // 16, blocked layout

struct SortCtx {
  int32_t n;
  int32_t num_tiles;
  int32_t num_slices;
  const int32_t* pt_off;     // [S+1] slice boundaries; tiles never straddle
  const int32_t* tile_first; // [num_tiles] first key of the tile
  const int32_t* tile_count; // [num_tiles] keys in the tile
  const int32_t* tile_slice; // [num_tiles]
  uint32_t* hist;            // [256][num_tiles] -> exclusive offsets
  const int64_t* key_in;
  const int32_t* val_in;
  int64_t* key_out;
  int32_t* val_out;
  int32_t shift;
};

__global__ __launch_bounds__(256) void
morton_encode_kernel(
  const int32_t* __restrict__ xyz, const int32_t* __restrict__ pt_off,
  int num_slices, int n, int64_t* key, int32_t* val)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    key[i] = morton_addr(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    const int s = find_slice(pt_off, num_slices, i);
    val[i] = i - pt_off[s];  // index local to the slice
  }
}

__global__ __launch_bounds__(kSortThreads) void
sort_hist_kernel(SortCtx cx)
{
  __shared__ uint32_t h[256];
  for (int tile = blockIdx.x; tile < cx.num_tiles; tile += gridDim.x) {
    h[threadIdx.x] = 0;
    __syncthreads();
    const int first = cx.tile_first[tile], cnt = cx.tile_count[tile];
    for (int i = threadIdx.x; i < cnt; i += kSortThreads)
      atomicAdd(&h[(uint32_t)(cx.key_in[first + i] >> cx.shift) & 255u], 1u);
    __syncthreads();
    cx.hist[(size_t)threadIdx.x * cx.num_tiles + tile] = h[threadIdx.x];
    __syncthreads();
  }
}

// Exclusive scan of hist in (slice, digit, tile) order: a key goes after all
// keys of earlier slices, then smaller digits of its slice, then earlier
// tiles with the same digit.  One workgroup; slices x 256 digits x tiles.
__global__ __launch_bounds__(1024) void
sort_scan_kernel(SortCtx cx)
{
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  if (threadIdx.x == 0)
    carry_s = 0;
  __syncthreads();
  // tiles of a slice are contiguous: [t0, t1)
  int t0 = 0;
  for (int s = 0; s < cx.num_slices; s++) {
    int t1 = t0;
    while (t1 < cx.num_tiles && cx.tile_slice[t1] == s)
      t1++;
    const int nt = t1 - t0;
    const int64_t total = (int64_t)256 * nt;
    for (int64_t b = 0; b < total; b += 1024) {
      const int64_t e = b + threadIdx.x;
      uint32_t v = 0;
      size_t addr = 0;
      if (e < total) {
        const int d = (int)(e / nt), t = t0 + (int)(e % nt);
        addr = (size_t)d * cx.num_tiles + t;
        v = cx.hist[addr];
      }
      uint32_t inc = wave_incl_scan_u32(v);
      if (lane == kWave - 1)
        wave_tot[wave] = inc;
      __syncthreads();
      uint32_t off = carry_s;
      for (int w = 0; w < wave; w++)
        off += wave_tot[w];
      if (e < total)
        cx.hist[addr] = off + inc - v;
      __syncthreads();
      if (threadIdx.x == 1023)
        carry_s = off + inc;
      __syncthreads();
    }
    t0 = t1;
  }
}

__global__ __launch_bounds__(kSortThreads) void
sort_scatter_kernel(SortCtx cx)
{
  __shared__ uint32_t digit_base[256];
  __shared__ uint32_t wave_cnt[4][256];  // per wave digit counts of a row
  __shared__ uint32_t run[256];          // digits placed by earlier rows
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const unsigned long long lt = (1ull << lane) - 1;
  for (int tile = blockIdx.x; tile < cx.num_tiles; tile += gridDim.x) {
    const int first = cx.tile_first[tile], cnt = cx.tile_count[tile];
    digit_base[threadIdx.x] = cx.hist[(size_t)threadIdx.x * cx.num_tiles + tile];
    run[threadIdx.x] = 0;
    __syncthreads();
    // rows of 256 consecutive keys keep the input order: wave w holds keys
    // [row*256 + w*64, +64), so rank = earlier rows + earlier waves + lanes
    for (int row = 0; row < cnt; row += kSortThreads) {
      const int i = row + threadIdx.x;
      const bool in = i < cnt;
      int64_t k = 0;
      int32_t v = 0;
      uint32_t d = 0;
      if (in) {
        k = cx.key_in[first + i];
        v = cx.val_in[first + i];
        d = (uint32_t)(k >> cx.shift) & 255u;
      }
      // lanes of this wave with the same digit, by 8 ballots
      unsigned long long same = in ? ~0ull : 0ull;
#pragma unroll
      for (int b = 0; b < 8; b++) {
        const unsigned long long bb = __ballot(in && ((d >> b) & 1));
        same &= ((d >> b) & 1) ? bb : ~bb;
      }
      same &= __ballot(in);
      const uint32_t before = __popcll(same & lt);
      for (int q = threadIdx.x; q < 4 * 256; q += kSortThreads)
        (&wave_cnt[0][0])[q] = 0;
      __syncthreads();
      if (in && before == 0)
        wave_cnt[wave][d] = __popcll(same);
      __syncthreads();
      if (in) {
        uint32_t off = digit_base[d] + run[d] + before;
        for (int w = 0; w < wave; w++)
          off += wave_cnt[w][d];
        cx.key_out[off] = k;
        cx.val_out[off] = v;
      }
      __syncthreads();
      run[threadIdx.x] += wave_cnt[0][threadIdx.x] + wave_cnt[1][threadIdx.x]
        + wave_cnt[2][threadIdx.x] + wave_cnt[3][threadIdx.x];
      __syncthreads();
    }
  }
}

// marshalling of the RAHT drivers (AttributeEncoder.cpp:1331-1338,
// 1364-1375): attributes into Morton order / clipped reconstruction back by
// original point index
__global__ __launch_bounds__(256) void
attr_gather_kernel(
  int n, int c, const int32_t* __restrict__ order, const int32_t* __restrict__ src,
  int32_t* __restrict__ dst)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const size_t o = (size_t)order[i] * c;
    for (int k = 0; k < c; k++)
      dst[(size_t)i * c + k] = src[o + k];
  }
}

__global__ __launch_bounds__(256) void
attr_clip_scatter_kernel(
  int n, int c, int32_t clip_max, const int32_t* __restrict__ order,
  const int32_t* __restrict__ src, int32_t* __restrict__ dst)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const size_t o = (size_t)order[i] * c;
    for (int k = 0; k < c; k++) {
      const int32_t v = src[(size_t)i * c + k];
      dst[o + k] = v < 0 ? 0 : (v > clip_max ? clip_max : v);
    }
  }
}

}  // namespace gpcc
