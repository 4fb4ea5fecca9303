// raht_arith.hpp -- the two arithmetic back ends of the RAHT dependency kernels.
//
// The transform is Q15 fixed point in int64 (tmc3/FixedPoint.h:44-124): every step is
// "multiply by a constant below 2^32, round half away from zero, >> 15".  On gfx950 a
// 64 x 32-bit integer multiply is three quarter-rate instructions plus the rounding
// sequence; measured (tools/ubench/mulround.hip, MI355X): one butterfly step -- four such
// products and two additions -- 224 cycles as a dependent chain of one wavefront, 148
// cycles of issue with the chip full.  Doubles hold every integer below 2^53 exactly and
// v_fma_f64 / v_trunc_f64 / v_floor_f64 are plain VALU instructions, so
//     round_half_away(a * c / 2^15) == trunc(fma(a, c * 2^-15, copysign(0.5, a)))
// EXACTLY whenever |a * c| + 2^14 < 2^53: the same butterfly step 100 cycles as a
// chain, 63 of issue.  ArithF64 is that back end: values are doubles that hold integers,
// results are bit-identical to ArithI64 while every product stays below 2^53, and the
// kernels check the magnitudes that bound the products (below()): a call whose values leave the
// range reports it and is redone with ArithI64 -- never a silently different result.
// With B-bit attributes and n points per slice the largest product is about
// 2^(B + 30) * sqrt(n) * 4, so 8-bit colour / reflectance slices of 2^20 points have a
// factor 2^6 to spare; the dispatcher only picks ArithF64 for attributes of at most 10 bits.
#pragma once

#include "gpcc_primitives.hpp"

namespace gpcc {

struct ArithI64 {
  static constexpr bool kF64 = false;
  static constexpr int kFwdLimit = 62, kInvLimit = 62, kRecLimit = 62;
  typedef int64_t T;     // a Q15 value (or an integer, where the reference holds one)
  typedef int64_t Coef;  // a multiplier in [0, 2^32): butterfly a / b, sqrt(w), 1 / sqrt(w), divisor
  struct Quant {
    Quantizer q;
  };
  GPCC_HD static T zero() { return 0; }
  GPCC_HD static T from_i64(int64_t v) { return v; }
  GPCC_HD static int64_t to_i64(T v) { return v; }
  GPCC_HD static int64_t to_small(T v) { return v; }   // an integer known to be below 2^31
  GPCC_HD static T from_int(int32_t v) { return fp_from_int(v); }   // FixedPoint = int
  GPCC_HD static Coef coef(int64_t c) { return c; }
  GPCC_HD static T mulc(T a, Coef c) { return fp_mul_c(a, c); }      // FixedPoint *= constant
  GPCC_HD static T muli(T a, int m) { return a * m; }                // prediction weight
  GPCC_HD static T round_int(T v) { return fp_round(v); }            // FixedPoint::round()
  GPCC_HD static T shr(T v, int s) { return v >> s; }                // arithmetic shift (floor)
  GPCC_HD static bool lt0(T v) { return v < 0; }
  GPCC_HD static Quant quant(Quantizer q) { return Quant{q}; }
  // Quantizer::quantize(coefficient << 8), coefficient = an integer
  GPCC_HD static int32_t quantize(const Quant& q, T co) { return (int32_t)gpcc::quantize(q.q, co * 256); }
  // FixedPoint(divExp2RoundHalfUp(Quantizer::scale(c), 8))
  GPCC_HD static T dequant_fp(const Quant& q, int32_t c) { return fp_from_int(dequantize(q.q, c)); }
  GPCC_HD static bool below(T, int) { return true; }
};

struct ArithF64 {
  static constexpr bool kF64 = true;
  typedef double T;
  typedef double Coef;  // c * 2^-15
  struct Quant {
    double recip18;  // recip * 2^-18: (|co| << 8) * recip / 2^26
    double off;      // (2^26 / 3) * 2^-26
    double step8;    // step / 256
  };
  // What the kernels check (below()): the values that ENTER the forward butterflies stay under
  // 2^34 and those that enter the inverse ones (transformed prediction + de-quantised residual,
  // inherited DC) under 2^35.  A block's transform is orthonormal up to rounding, so nothing inside
  // exceeds sqrt(8) times its inputs and every product with a coefficient <= 2^16 stays below 2^52.5;
  // a product that were inexact (>= 2^53) yields a result >= 2^38 and fails the next check, so a
  // wrong value cannot pass unnoticed.  Reconstructed means are held under 2^30 (their weighted
  // sum times the divisor: 2^6 * 2^30 * 2^15).
  static constexpr int kFwdLimit = 34, kInvLimit = 35, kRecLimit = 30;
  GPCC_HD static T zero() { return 0.0; }
  GPCC_HD static T from_i64(int64_t v) { return (double)v; }
  GPCC_HD static int64_t to_i64(T v) { return (int64_t)v; }
  GPCC_HD static int64_t to_small(T v) { return (int64_t)(int32_t)v; }
  GPCC_HD static T from_int(int32_t v) { return (double)v * 32768.0; }
  GPCC_HD static Coef coef(int64_t c) { return (double)c * (1.0 / 32768.0); }
  GPCC_HD static T mulc(T a, Coef c)
  {
    return __builtin_trunc(__builtin_fma(a, c, __builtin_copysign(0.5, a)));
  }
  GPCC_HD static T muli(T a, int m) { return a * (double)m; }
  GPCC_HD static T round_int(T v) { return mulc(v, 1.0 / 32768.0); }
  GPCC_HD static T shr(T v, int s) { return __builtin_floor(__builtin_ldexp(v, -s)); }
  GPCC_HD static bool lt0(T v) { return v < 0.0; }
  GPCC_HD static Quant quant(Quantizer q)
  {
    return Quant{(double)q.recip * (1.0 / 262144.0), (double)(((int64_t)1 << 26) / 3) * (1.0 / 67108864.0),
                 (double)q.step * (1.0 / 256.0)};
  }
  GPCC_HD static int32_t quantize(const Quant& q, T co)
  {
    // sign(co) * floor((|co << 8| * recip + 2^26 / 3) / 2^26)
    const double m = __builtin_floor(__builtin_fma(__builtin_fabs(co), q.recip18, q.off));
    return (int32_t)__builtin_copysign(m, co);
  }
  GPCC_HD static T dequant_fp(const Quant& q, int32_t c)
  {
    // ((c * step + 128) >> 8) << 15, sign-symmetric shift left
    return __builtin_floor(__builtin_fma((double)c, q.step8, 0.5)) * 32768.0;
  }
  GPCC_HD static bool below(T v, int log2_limit) { return __builtin_fabs(v) < (double)((int64_t)1 << log2_limit); }
};

}  // namespace gpcc
