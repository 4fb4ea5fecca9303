// residual_bins.hpp -- binarisation of the attribute residual symbols: the part
// of PCCResidualsEncoder (tmc3/AttributeEncoder.cpp:57-307) that is not the
// adaptive arithmetic coder.
//
// The reference codes every symbol of the entropy loops -- a zero-run length
// (encodeRunLength :227-254) followed by a coefficient tuple (encode :278-307
// through encodeSymbol :259-272 and the exp-Golomb binarisations of
// entropyutils.h:142-183) -- as a sequence of binary decisions, each with
// either one of 31 adaptive contexts or in bypass.  WHICH context a decision
// uses depends on the symbol alone (its own magnitudes), never on what was
// coded before; only the context's probability state is sequential.  So the
// decisions of all symbols can be produced in parallel where the symbols are,
// and the host's loop shrinks to
//     for (b : bins) b.ctx == kBinBypass ? enc.encode(b.bin) : enc.encode(b.bin, model[b.ctx]);
// with the reference's own coder and context memory -- the bitstream is
// identical (tests/test_bins.py).
//
// One byte per decision: (context id << 1) | bin, context ids in the order the
// reference declares the models (tmc3/AttributeCommon.h:54-57):
//     0..4    ctxRunLen[5]
//     5..18   ctxCoeffGtN[2][7]
//    19..24   ctxCoeffRemPrefix[2][3]
//    25..30   ctxCoeffRemSuffix[2][3]
//    31       bypass
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gpcc_primitives.hpp"

namespace gpcc {

constexpr int kBinCtxRunLen = 0;
constexpr int kBinCtxGtN = 5;
constexpr int kBinCtxRemPrefix = 19;
constexpr int kBinCtxRemSuffix = 25;
constexpr int kBinBypass = 31;

// counts, or writes, the decisions of one symbol
struct BinSink {
  uint8_t* out;  // nullptr: count only
  int n;
  GPCC_HD void put(int ctx, int bin)
  {
    if (out)
      out[n] = (uint8_t)((ctx << 1) | (bin & 1));
    n++;
  }
};

// encodeRunLength (tmc3/AttributeEncoder.cpp:227-254)
GPCC_HD void
bins_run_length(BinSink& s, int run)
{
  int ctx = kBinCtxRunLen;
  for (int i = 0; i < (run < 3 ? run : 3); i++, ctx++)
    s.put(ctx, 1);
  if (run < 3) {
    s.put(ctx, 0);
    return;
  }
  run -= 3;
  const int prefix = run >> 1;
  for (int i = 0; i < (prefix < 4 ? prefix : 4); i++)
    s.put(ctx, 1);
  if (run < 8) {
    s.put(ctx, 0);
    s.put(kBinBypass, run & 1);
    return;
  }
  run -= 8;
  // encodeExpGolomb(run, 2, one context) (entropyutils.h:142-157)
  ctx++;
  uint32_t sym = (uint32_t)run;
  int k = 2;
  while (sym >= (1u << k)) {
    s.put(ctx, 1);
    sym -= 1u << k;
    k++;
  }
  s.put(ctx, 0);
  while (k--)
    s.put(kBinBypass, (sym >> k) & 1);
}

// encodeSymbol (tmc3/AttributeEncoder.cpp:259-272)
GPCC_HD void
bins_symbol(BinSink& s, uint32_t value, int k1, int k2, int k3)
{
  s.put(kBinCtxGtN + k1, value > 0);
  if (!value)
    return;
  --value;
  s.put(kBinCtxGtN + 7 + k2, value > 0);
  if (!value)
    return;
  --value;
  // encodeExpGolomb(value, 1, prefix contexts, suffix contexts) (entropyutils.h:164-183)
  int k = 1;
  while (value >= (1u << k)) {
    const int p = k - 1 < 2 ? k - 1 : 2;
    s.put(kBinCtxRemPrefix + 3 * k3 + p, 1);
    value -= 1u << k;
    k++;
  }
  {
    const int p = k - 1 < 2 ? k - 1 : 2;
    s.put(kBinCtxRemPrefix + 3 * k3 + p, 0);
  }
  while (k--)
    s.put(kBinCtxRemSuffix + 3 * k3 + (k < 2 ? k : 2), (value >> k) & 1);
}

// encode(value0, value1, value2) (:278-300) / encode(value) (:304-307)
GPCC_HD void
bins_values(BinSink& s, const int32_t* v, int c)
{
  if (c == 3) {
    const int mag0 = v[0] < 0 ? -v[0] : v[0];
    const int mag1 = v[1] < 0 ? -v[1] : v[1];
    const int mag2 = v[2] < 0 ? -v[2] : v[2];
    const int b0 = mag1 == 0, b1 = mag1 <= 1, b2 = mag2 == 0, b3 = mag2 <= 1;
    bins_symbol(s, (uint32_t)mag1, 0, 0, 1);
    bins_symbol(s, (uint32_t)mag2, 1 + b0, 1 + b1, 1);
    const int mag0m = (b0 && b2) ? mag0 - 1 : mag0;
    bins_symbol(s, (uint32_t)mag0m, 3 + (b0 << 1) + b2, 3 + (b1 << 1) + b3, 0);
    if (mag0)
      s.put(kBinBypass, v[0] < 0);
    if (mag1)
      s.put(kBinBypass, v[1] < 0);
    if (mag2)
      s.put(kBinBypass, v[2] < 0);
  } else {
    // (two components do not occur: attributes are colour or a scalar)
    const int mag = (v[0] < 0 ? -v[0] : v[0]) - 1;
    bins_symbol(s, (uint32_t)mag, 0, 0, 0);
    s.put(kBinBypass, v[0] < 0);
  }
}

// symbol k of the stream: its run, then its values; symbol `num_symbols` is the
// trailing run (present when non-zero)
GPCC_HD void
bins_of_symbol(
  BinSink& s, int k, int num_symbols, const int32_t* runs, const int32_t* values, int trailing_run, int c)
{
  if (k < num_symbols) {
    bins_run_length(s, runs[k]);
    bins_values(s, values + (size_t)k * c, c);
  } else if (trailing_run) {
    bins_run_length(s, trailing_run);
  }
}

constexpr int kBinBlock = 256;

// pass 1: decisions per symbol, summed per block of 256 symbols
__global__ __launch_bounds__(kBinBlock) void
bins_count_kernel(
  int num_symbols, const int32_t* __restrict__ runs, const int32_t* __restrict__ values,
  int trailing_run, int c, int32_t* __restrict__ counts, int32_t* __restrict__ block_sum)
{
  __shared__ int wsum[kBinBlock / 64];
  const int k = blockIdx.x * kBinBlock + threadIdx.x;
  int cnt = 0;
  if (k <= num_symbols) {
    BinSink s{nullptr, 0};
    bins_of_symbol(s, k, num_symbols, runs, values, trailing_run, c);
    cnt = s.n;
    counts[k] = cnt;
  }
  int v = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
    v += __shfl_xor(v, d);
  if ((threadIdx.x & 63) == 0)
    wsum[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0)
    block_sum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// pass 2: exclusive scan of the block sums (one workgroup)
__global__ __launch_bounds__(1024) void
bins_scan_kernel(int nblocks, const int32_t* __restrict__ block_sum, long long* __restrict__ block_base)
{
  // (a block holds at most 256 symbols' decisions: its sum fits 32 bits; the running
  // total of a long colour stream does not -- offsets and total are 64 bit)
  __shared__ long long part[1024];
  const int per = (nblocks + 1023) / 1024;
  const int b0 = threadIdx.x * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
  long long sum = 0;
  for (int b = b0; b < b1; b++)
    sum += block_sum[b];
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long run = 0;
    for (int i = 0; i < 1024; i++) {
      const long long v = part[i];
      part[i] = run;
      run += v;
    }
    block_base[nblocks] = run;  // the total
  }
  __syncthreads();
  long long run = part[threadIdx.x];
  for (int b = b0; b < b1; b++) {
    block_base[b] = run;
    run += block_sum[b];
  }
}

// pass 3: every symbol writes its decisions at its offset
__global__ __launch_bounds__(kBinBlock) void
bins_emit_kernel(
  int num_symbols, const int32_t* __restrict__ runs, const int32_t* __restrict__ values,
  int trailing_run, int c, const int32_t* __restrict__ counts,
  const long long* __restrict__ block_base, uint8_t* __restrict__ bins)
{
  __shared__ int wsum[kBinBlock / 64];
  const int k = blockIdx.x * kBinBlock + threadIdx.x;
  const int cnt = k <= num_symbols ? counts[k] : 0;
  // exclusive scan of the block's counts
  int inc = cnt;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d);
    if (lane >= d)
      inc += o;
  }
  if (lane == 63)
    wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  long long off = block_base[blockIdx.x] + inc - cnt;
  for (int w = 0; w < (int)(threadIdx.x >> 6); w++)
    off += wsum[w];
  if (k <= num_symbols && cnt) {
    BinSink s{bins + off, 0};
    bins_of_symbol(s, k, num_symbols, runs, values, trailing_run, c);
  }
}

}  // namespace gpcc
