/* primitives.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Plain-C restatement of the integer primitives the attribute-transform
 * path of TMC13 is built from.  Each function cites the reference code it
 * follows (paths relative to the reference tree).  Pinned against the
 * compiled reference (oracle/_ref/libtmc3_ref.so) by
 * tests/test_oracle_primitives.py.
 */
#ifndef GPCC_ORACLE_PRIMITIVES_H
#define GPCC_ORACLE_PRIMITIVES_H

#include <stdint.h>

/* ---- FixedPoint (tmc3/FixedPoint.h:44-124): Q15 in int64 -------------- */
#define FP_FRAC 15
#define FP_HALF (1 << (FP_FRAC - 1))

/* FixedPoint::operator=(int64) FixedPoint.h:88-94 */
static inline int64_t
fp_from_int(int64_t v)
{
  return v > 0 ? (int64_t)((uint64_t)v << FP_FRAC)
               : -(int64_t)((uint64_t)(-v) << FP_FRAC);
}

/* FixedPoint::round FixedPoint.h:78-83 */
static inline int64_t
fp_round(int64_t v)
{
  return v > 0 ? (FP_HALF + v) >> FP_FRAC : -((FP_HALF - v) >> FP_FRAC);
}

/* FixedPoint::operator*= FixedPoint.h:115-123 */
static inline int64_t
fp_mul(int64_t a, int64_t b)
{
  int64_t p = (int64_t)((uint64_t)a * (uint64_t)b);
  return p < 0 ? -((FP_HALF - p) >> FP_FRAC) : (FP_HALF + p) >> FP_FRAC;
}

/* ---- bit helpers ------------------------------------------------------ */
/* ilog2 PCCMisc.h:150-165: floor(log2 x), ilog2(0) = -1 */
static inline int
ilog2_u64(uint64_t x)
{
  return x ? 63 - __builtin_clzll(x) : -1;
}
static inline int
ilog2_u32(uint32_t x)
{
  return x ? 31 - __builtin_clz(x) : -1;
}

/* morton3dAdd PCCMisc.h:245-256: per-axis add of interleaved addresses */
static inline uint64_t
morton3d_add(uint64_t a, uint64_t b)
{
  uint64_t mask = 0x9249249249249249ull, val = 0;
  for (int i = 0; i < 3; i++) {
    val |= ((a | ~mask) + (b & mask)) & mask;
    mask <<= 1;
  }
  return val;
}

/* mortonAddr PCCMath.h:606-616: x -> bit 2, y -> bit 1, z -> bit 0 of
 * every triplet; 24 bits per axis survive the three table look-ups. */
static inline uint64_t
spread3(uint32_t v)
{
  uint64_t x = v & 0x1fffffu;
  x = (x | x << 32) & 0x001f00000000ffffull;
  x = (x | x << 16) & 0x001f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x9249249249249249ull;
  return x;
}
static inline int64_t
morton_addr(int32_t x, int32_t y, int32_t z)
{
  return (int64_t)((spread3((uint32_t)x) << 2) | (spread3((uint32_t)y) << 1)
                   | spread3((uint32_t)z));
}

/* divExp2RoundHalfUp PCCMath.h:651-658 */
static inline int64_t
div_exp2_round_half_up(int64_t x, int s)
{
  return s ? (x + ((int64_t)1 << (s - 1))) >> s : x;
}

/* divExp2RoundHalfInf PCCMath.h:665-673 */
static inline int64_t
div_exp2_round_half_inf(int64_t x, int s)
{
  if (!s)
    return x;
  int64_t h = (int64_t)1 << (s - 1);
  return x >= 0 ? (h + x) >> s : -((h - x) >> s);
}

/* ---- irsqrt / isqrt (tmc3/misc.cpp:139-222) --------------------------- */
/* k3timesR[i] = R3_12[i] << 20, kRcubed[i] = RC_22[i] << 10
 * (misc.cpp:152-186, normative look-up tables of the G-PCC fixed-point
 * inverse square root). */
static const uint16_t kR3_12[96] = {
  0xbe8, 0xbb8, 0xb94, 0xb64, 0xb40, 0xb10, 0xaec, 0xac8, 0xab0, 0xa8c, 0xa68,
  0xa50, 0xa2c, 0xa14, 0x9f0, 0x9d8, 0x9c0, 0x9a8, 0x990, 0x978, 0x960, 0x948,
  0x930, 0x918, 0x90c, 0x8f4, 0x8dc, 0x8d0, 0x8b8, 0x8ac, 0x894, 0x888, 0x870,
  0x864, 0x858, 0x840, 0x834, 0x828, 0x810, 0x804, 0x7f8, 0x7ec, 0x7e0, 0x7d4,
  0x7c8, 0x7bc, 0x7a4, 0x798, 0x78c, 0x780, 0x774, 0x774, 0x768, 0x750, 0x750,
  0x744, 0x738, 0x72c, 0x720, 0x714, 0x714, 0x708, 0x6fc, 0x6f0, 0x6e4, 0x6e4,
  0x6d8, 0x6cc, 0x6c0, 0x6c0, 0x6b4, 0x6a8, 0x6a8, 0x69c, 0x690, 0x690, 0x684,
  0x678, 0x678, 0x66c, 0x66c, 0x660, 0x654, 0x654, 0x648, 0x648, 0x63c, 0x630,
  0x630, 0x624, 0x624, 0x618, 0x618, 0x60c, 0x60c, 0x600};
static const uint32_t kRC_22[96] = {
  0x3e82f7, 0x3b9abd, 0x397bfe, 0x36bc9e, 0x34bbfd, 0x32242b, 0x3040d9,
  0x2e69c6, 0x2d368b, 0x2b739f, 0x29bcac, 0x289e69, 0x26fad5, 0x25e971,
  0x2458c1, 0x2353f4, 0x225407, 0x2158f6, 0x2062b1, 0x1f713f, 0x1e8488,
  0x1d9c6f, 0x1cb912, 0x1bda3e, 0x1b6c8a, 0x1a9498, 0x19c106, 0x1958e5,
  0x188bff, 0x18272e, 0x1760df, 0x16ff3f, 0x163f51, 0x15e0e1, 0x158384,
  0x14cc02, 0x1471cb, 0x141886, 0x136920, 0x13130b, 0x12bde2, 0x1269a2,
  0x121684, 0x11c43e, 0x11730b, 0x1122d3, 0x108541, 0x1037d7, 0xfeb71,
  0xf9ff8,  0xf5572,  0xf5576,  0xf0bd0,  0xe7b76,  0xe7b70,  0xe34a0,
  0xdeeb8,  0xda9b0,  0xd6575,  0xd223c,  0xd223f,  0xcdfef,  0xc9e77,
  0xc5dd2,  0xc1e0c,  0xc1e1e,  0xbdf40,  0xba137,  0xb6401,  0xb6405,
  0xb27b0,  0xaec28,  0xaec3d,  0xab185,  0xa77b4,  0xa77b2,  0xa3eaf,
  0xa067a,  0xa067b,  0x9cf0f,  0x9cf0d,  0x99877,  0x962b4,  0x962bc,
  0x92dc4,  0x92dac,  0x8f981,  0x8c604,  0x8c61c,  0x89368,  0x89376,
  0x86183,  0x86189,  0x83064,  0x8306c,  0x80005};

/* irsqrt misc.cpp:191-222: ~ 2^40 / sqrt(a) */
static inline uint64_t
irsqrt_u64(uint64_t a64)
{
  if (!a64)
    return 0;
  int shift = -3;
  while (a64 & 0xffffffff00000000ull) {
    a64 >>= 2;
    shift--;
  }
  uint32_t a = (uint32_t)a64;
  while (!(a & 0xc0000000u)) {
    a <<= 2;
    shift++;
  }
  int idx = (int)(a >> 25) - 32;
  uint64_t r = ((uint64_t)kR3_12[idx] << 20)
    - ((((uint64_t)kRC_22[idx] << 10) * a) >> 32);
  uint64_t ar = (r * a) >> 32;
  uint64_t s = 0x30000000ull - ((r * ar) >> 32);
  r = (r * s) >> 32;
  return shift > 0 ? r << shift : r >> -shift;
}

/* isqrt misc.cpp:139-146 */
static inline uint32_t
isqrt_u64(uint64_t x)
{
  if (x <= ((uint64_t)1 << 46))
    return (uint32_t)(1 + ((x * irsqrt_u64(x)) >> 40));
  uint64_t x0 = (x + 65536) >> 16;
  return (uint32_t)(1 + ((x0 * irsqrt_u64(x0)) >> 32));
}

/* ---- quantiser (tmc3/quantization.h:79-102, quantization.cpp:46-52,
 *      tables.cpp:478-481) --------------------------------------------- */
static const int32_t kQpStepTab[6] = {161, 181, 203, 228, 256, 287};
static const int32_t kQpStepRecipTab[6] = {416825, 370767, 330586,
                                           294337, 262144, 233829};
typedef struct {
  int32_t step;
  int32_t recip;
} quantizer_t;

static inline quantizer_t
quantizer_make(int qp)
{
  if (qp < 4)
    qp = 4;
  quantizer_t q = {kQpStepTab[qp % 6] << (qp / 6),
                   kQpStepRecipTab[qp % 6] >> (qp / 6)};
  return q;
}

/* Quantizer::quantize quantization.h:79-93 (fracBits = 18 + 8) */
static inline int64_t
quantizer_quantize(quantizer_t q, int64_t x)
{
  const int64_t off = ((int64_t)1 << 26) / 3;
  return x >= 0 ? (x * q.recip + off) >> 26 : -((off - x * q.recip) >> 26);
}

/* Quantizer::scale quantization.h:97-102 */
static inline int64_t
quantizer_scale(quantizer_t q, int64_t x)
{
  return x * q.step;
}

static inline int
clip_int(int v, int lo, int hi)
{
  return v < lo ? lo : (v > hi ? hi : v);
}

/* ---- divApprox (tmc3/PCCMath.h:715-737; LUT misc.cpp:313: the table is
 *      round(65536/(i+1)) - 1, checked against the exported reference
 *      symbol by the tests) -------------------------------------------- */
static inline int64_t
div_approx(int64_t a, uint64_t b, int log2scale)
{
  int n = ilog2_u64(b) + 1 - 8;
  if (n < 0)
    n = 0;
  uint64_t index = (b + (((uint64_t)1 << n) >> 1)) >> n;
  int64_t inv = (int64_t)((2 * 65536 + index) / (2 * index) - 1) + 1;
  return (inv * a) >> (n + 16 - log2scale);
}

#endif
