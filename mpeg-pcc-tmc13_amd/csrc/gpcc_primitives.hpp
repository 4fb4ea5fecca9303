// gpcc_primitives.hpp -- integer primitives of the G-PCC attribute path for
// gfx950 device code (also callable on the host for unit tests).
//
// Every function names the reference code whose RESULT it reproduces
// (paths relative to the reference tree, TMC13 release-23.0-rc2).  The
// formulations are branch-light closed forms for a 64-lane SIMD machine
// (clz-based normalisation instead of shift loops, 32x32->64 multiplies
// where the operand ranges allow), not transcriptions.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define GPCC_HD __host__ __device__ __forceinline__

// A floor of 64 vector registers for a kernel's wavefronts.  finish_kernel in the 56-register
// allocation its code arrives at gave wrong duplicate-chain coefficients on the MI355X in large
// batches -- the SAME machine code is clean with 64 / 72 / 80 registers allocated, no instruction
// changed (profiles/r04_finish_lds_root_cause.txt: asm re-assembly with only the descriptor edited).
// Round 4 saw it again after an unrelated edit had changed the kernel's instruction stream (the
// table form that had "never failed" did, in the pinned batch 144), so every kernel whose allocation
// falls into that class (49..56 registers) and that runs long per-lane loops is raised to 64; it costs
// nothing (eight wavefronts per SIMD either way).  tools/isa_audit.py lists the kernels of the class.
#if defined(__HIP_DEVICE_COMPILE__)
#define GPCC_VGPR_FLOOR_64() asm volatile("" ::: "v63")
#else
#define GPCC_VGPR_FLOOR_64() ((void)0)
#endif

// an ordered group of `n` instructions of the classes in `mask` for the machine scheduler
// (__builtin_amdgcn_sched_group_barrier: 0x002 VALU, 0x100 LDS read); nothing on the host and under the emulator
#if defined(__HIP_DEVICE_COMPILE__)
#define GPCC_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#else
#define GPCC_SCHED_GROUP(mask, n) ((void)0)
#endif

namespace gpcc {

constexpr int kFpFrac = 15;                 // FixedPoint::kFracBits
constexpr int64_t kFpHalf = 1 << (kFpFrac - 1);

// ---- Q15 fixed point (tmc3/FixedPoint.h:78-123) -----------------------
// FixedPoint::operator=(int64): sign-symmetric << 15
GPCC_HD int64_t fp_from_int(int64_t v)
{
  uint64_t m = (uint64_t)(v < 0 ? -v : v) << kFpFrac;
  return v < 0 ? -(int64_t)m : (int64_t)m;
}

// Sign-symmetric "round half away from zero, then >> s" without the
// abs/negate pair: for p < 0,  -((h - p) >> s) == (p + h - 1) >> s  with an
// arithmetic shift (h = 2^(s-1)), so one conditional -1 replaces four
// 64-bit operations.
GPCC_HD int64_t round_shift_sym(int64_t p, int s)
{
  return (p + ((int64_t)1 << (s - 1)) + (p >> 63)) >> s;
}

// FixedPoint::round(): sign-symmetric round to nearest, >> 15
GPCC_HD int64_t fp_round(int64_t v) { return round_shift_sym(v, kFpFrac); }

// FixedPoint::operator*=: 64-bit product, sign-symmetric rounding >> 15
GPCC_HD int64_t fp_mul(int64_t a, int64_t b)
{
  return round_shift_sym((int64_t)((uint64_t)a * (uint64_t)b), kFpFrac);
}

// a * c for a multiplier known to lie in [0, 2^32): two 32-bit multiplies
// instead of the three of a 64 x 64 product (32-bit integer multiplies are
// quarter rate, and these sit on the dependency chains).  Every constant of
// the transform qualifies: butterfly a/b <= 2^15, 1/sqrt(w) <= 2^15,
// sqrt(w) < 2^30 for w <= GPCC_MAX_POINTS, the prediction divisor, the
// quantiser step and reciprocal.
GPCC_HD int64_t mul_i64_u32(int64_t a, int64_t c)
{
  const uint32_t cl = (uint32_t)c;
  const uint64_t lo = (uint64_t)(uint32_t)a * cl;
  const uint32_t hi = (uint32_t)((uint64_t)a >> 32) * cl + (uint32_t)(lo >> 32);
  return (int64_t)(((uint64_t)hi << 32) | (uint32_t)lo);
}

// FixedPoint::operator*= by such a constant
GPCC_HD int64_t fp_mul_c(int64_t a, int64_t c)
{
  return round_shift_sym(mul_i64_u32(a, c), kFpFrac);
}

// ---- bit helpers --------------------------------------------------------
GPCC_HD int clz64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __clzll((long long)x);
#else
  return x ? __builtin_clzll(x) : 64;
#endif
}
GPCC_HD int clz32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __clz((int)x);
#else
  return x ? __builtin_clz(x) : 32;
#endif
}
GPCC_HD int popc32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __popc(x);
#else
  return __builtin_popcount(x);
#endif
}

// number of significant bits (0 for x == 0); ilog2(x) == bitlen(x) - 1
// (tmc3/PCCMisc.h:150-165)
GPCC_HD int bitlen64(uint64_t x) { return 64 - clz64(x); }
GPCC_HD int ilog2_u64(uint64_t x) { return 63 - clz64(x); }

// morton3dAdd (tmc3/PCCMisc.h:245-256): per-axis addition of two
// interleaved 3-D addresses.
GPCC_HD uint64_t morton3d_add(uint64_t a, uint64_t b)
{
  constexpr uint64_t mz = 0x9249249249249249ull;
  constexpr uint64_t my = mz << 1, mx = mz << 2;
  return (((a | ~mz) + (b & mz)) & mz) | (((a | ~my) + (b & my)) & my)
    | (((a | ~mx) + (b & mx)) & mx);
}

// mortonAddr (tmc3/PCCMath.h:606-616): x -> bit 2, y -> bit 1, z -> bit 0
// of each triplet, 21 bits per axis.
GPCC_HD uint64_t spread3_21(uint32_t v)
{
  uint64_t x = v & 0x1fffffu;
  x = (x | x << 32) & 0x001f00000000ffffull;
  x = (x | x << 16) & 0x001f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
GPCC_HD int64_t morton_addr(int32_t x, int32_t y, int32_t z)
{
  return (int64_t)((spread3_21((uint32_t)x) << 2)
                   | (spread3_21((uint32_t)y) << 1) | spread3_21((uint32_t)z));
}

// divExp2RoundHalfUp (tmc3/PCCMath.h:651-658)
GPCC_HD int64_t div_exp2_round_half_up(int64_t x, int s)
{
  return s ? (x + ((int64_t)1 << (s - 1))) >> s : x;
}

// ---- inverse square root (tmc3/misc.cpp:152-222) ----------------------
// Normative look-up tables of the fixed-point rsqrt, stored compactly:
// k3timesR[i] == kRsqrt3R[i] << 20, kRcubed[i] == kRsqrtR3[i] << 10.
struct RsqrtLut {
  uint16_t r3[96];
  uint32_t rc[96];
};

#define GPCC_RSQRT_R3                                                        \
  0xbe8, 0xbb8, 0xb94, 0xb64, 0xb40, 0xb10, 0xaec, 0xac8, 0xab0, 0xa8c,      \
    0xa68, 0xa50, 0xa2c, 0xa14, 0x9f0, 0x9d8, 0x9c0, 0x9a8, 0x990, 0x978,    \
    0x960, 0x948, 0x930, 0x918, 0x90c, 0x8f4, 0x8dc, 0x8d0, 0x8b8, 0x8ac,    \
    0x894, 0x888, 0x870, 0x864, 0x858, 0x840, 0x834, 0x828, 0x810, 0x804,    \
    0x7f8, 0x7ec, 0x7e0, 0x7d4, 0x7c8, 0x7bc, 0x7a4, 0x798, 0x78c, 0x780,    \
    0x774, 0x774, 0x768, 0x750, 0x750, 0x744, 0x738, 0x72c, 0x720, 0x714,    \
    0x714, 0x708, 0x6fc, 0x6f0, 0x6e4, 0x6e4, 0x6d8, 0x6cc, 0x6c0, 0x6c0,    \
    0x6b4, 0x6a8, 0x6a8, 0x69c, 0x690, 0x690, 0x684, 0x678, 0x678, 0x66c,    \
    0x66c, 0x660, 0x654, 0x654, 0x648, 0x648, 0x63c, 0x630, 0x630, 0x624,    \
    0x624, 0x618, 0x618, 0x60c, 0x60c, 0x600
#define GPCC_RSQRT_RC                                                        \
  0x3e82f7, 0x3b9abd, 0x397bfe, 0x36bc9e, 0x34bbfd, 0x32242b, 0x3040d9,      \
    0x2e69c6, 0x2d368b, 0x2b739f, 0x29bcac, 0x289e69, 0x26fad5, 0x25e971,    \
    0x2458c1, 0x2353f4, 0x225407, 0x2158f6, 0x2062b1, 0x1f713f, 0x1e8488,    \
    0x1d9c6f, 0x1cb912, 0x1bda3e, 0x1b6c8a, 0x1a9498, 0x19c106, 0x1958e5,    \
    0x188bff, 0x18272e, 0x1760df, 0x16ff3f, 0x163f51, 0x15e0e1, 0x158384,    \
    0x14cc02, 0x1471cb, 0x141886, 0x136920, 0x13130b, 0x12bde2, 0x1269a2,    \
    0x121684, 0x11c43e, 0x11730b, 0x1122d3, 0x108541, 0x1037d7, 0xfeb71,     \
    0xf9ff8, 0xf5572, 0xf5576, 0xf0bd0, 0xe7b76, 0xe7b70, 0xe34a0, 0xdeeb8,  \
    0xda9b0, 0xd6575, 0xd223c, 0xd223f, 0xcdfef, 0xc9e77, 0xc5dd2, 0xc1e0c,  \
    0xc1e1e, 0xbdf40, 0xba137, 0xb6401, 0xb6405, 0xb27b0, 0xaec28, 0xaec3d,  \
    0xab185, 0xa77b4, 0xa77b2, 0xa3eaf, 0xa067a, 0xa067b, 0x9cf0f, 0x9cf0d,  \
    0x99877, 0x962b4, 0x962bc, 0x92dc4, 0x92dac, 0x8f981, 0x8c604, 0x8c61c,  \
    0x89368, 0x89376, 0x86183, 0x86189, 0x83064, 0x8306c, 0x80005

// irsqrt(a) ~ 2^40 / sqrt(a).  `lut` may live in LDS (device) or in host
// memory.  The reference's two normalisation loops (shift right by 2 while
// the value needs more than 32 bits, then left by 2 while the top two bits
// are clear) are closed forms of the bit length.
GPCC_HD uint64_t irsqrt(uint64_t a64, const RsqrtLut& lut)
{
  if (!a64)
    return 0;
  int n = bitlen64(a64);
  int s1 = n > 32 ? (n - 31) >> 1 : 0;  // ceil((n - 32) / 2)
  uint32_t a = (uint32_t)(a64 >> (2 * s1));
  int s2 = clz32(a) >> 1;
  a <<= 2 * s2;
  int shift = -3 - s1 + s2;

  int idx = (int)(a >> 25) - 32;
  // all intermediates fit 32 bits: 32x32->64 multiplies only
  uint64_t r = ((uint64_t)lut.r3[idx] << 20)
    - ((((uint64_t)lut.rc[idx] << 10) * a) >> 32);
  uint64_t ar = (r * a) >> 32;
  uint64_t s = 0x30000000ull - ((r * ar) >> 32);
  r = (r * s) >> 32;
  return shift > 0 ? r << shift : r >> -shift;
}

// isqrt (tmc3/misc.cpp:139-146)
GPCC_HD uint32_t isqrt(uint64_t x, const RsqrtLut& lut)
{
  if (x <= ((uint64_t)1 << 46))
    return (uint32_t)(1 + ((x * irsqrt(x, lut)) >> 40));
  uint64_t x0 = (x + 65536) >> 16;
  return (uint32_t)(1 + ((x0 * irsqrt(x0, lut)) >> 32));
}

// ---- quantiser (tmc3/quantization.h:79-102, quantization.cpp:46-52) ---
struct Quantizer {
  int32_t step;
  int32_t recip;
};

GPCC_HD Quantizer make_quantizer(int qp)
{
  // kQpStep / kQpStepRecip (tmc3/tables.cpp:478-481) as selects: the
  // index is data dependent per lane, a table would be a scattered load.
  qp = qp < 4 ? 4 : qp;
  int sh = qp / 6, m = qp - 6 * sh;
  int32_t st = m == 0 ? 161 : m == 1 ? 181 : m == 2 ? 203 : m == 3 ? 228
    : m == 4                                                        ? 256
                                                                    : 287;
  int32_t rc = m == 0 ? 416825 : m == 1 ? 370767 : m == 2 ? 330586
    : m == 3                                              ? 294337
    : m == 4                                              ? 262144
                                                          : 233829;
  return Quantizer{st << sh, rc >> sh};
}

GPCC_HD int64_t quantize(Quantizer q, int64_t x)
{
  // x >= 0: (x r + off) >> 26;  x < 0: -((off - x r) >> 26)
  //                                  == (x r + 2^26 - 1 - off) >> 26
  constexpr int64_t off = ((int64_t)1 << 26) / 3;
  constexpr int64_t noff = ((int64_t)1 << 26) - 1 - off;
  return (mul_i64_u32(x, q.recip) + (x < 0 ? noff : off)) >> 26;
}

GPCC_HD int64_t dequantize(Quantizer q, int64_t c)
{
  // Quantizer::scale followed by divExp2RoundHalfUp(.., 8)
  // (tmc3/RAHT.cpp:1702-1703)
  return (mul_i64_u32(c, q.step) + 128) >> 8;
}

GPCC_HD int clip(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// round(32768 / d): kDivisors[d - 1] of intraDcPred (tmc3/RAHT.cpp:445-451)
GPCC_HD int64_t pred_divisor(int d) { return (32768 + (d >> 1)) / d; }

}  // namespace gpcc
