// tests/emu/arith_check.cpp -- TEST INFRASTRUCTURE: the two arithmetic back ends of the RAHT
// dependency kernels (csrc/raht_arith.hpp) against each other on the CPU.  ArithF64 must give
// ArithI64's bits wherever its range conditions hold; prints the number of differences per primitive.
//   g++ -O1 -std=c++17 -I tests/emu -I include -I mpeg-pcc-tmc13_amd/csrc tests/emu/arith_check.cpp
#include <cstdio>
#include <cstdlib>
#include <random>

#include "hip/hip_runtime.h"

#include "raht_arith.hpp"

using namespace gpcc;

int
main(int argc, char** argv)
{
  const long n = argc > 1 ? atol(argv[1]) : 2000000;
  std::mt19937_64 rng(argc > 2 ? atol(argv[2]) : 12345);
  long bad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t edge[] = {0, 1, -1, 16383, 16384, 16385, -16383, -16384, -16385, 32767, 32768, -32768, 49152, -49152};
  for (long it = 0; it < n; it++) {
    // a value of up to 37 bits, a coefficient of up to 16 bits (butterfly, 1 / sqrt(w)) or 26 (sqrt(w))
    const int vb = 1 + (int)(rng() % 37);
    int64_t v = (int64_t)(rng() & (((uint64_t)1 << vb) - 1));
    if (rng() & 1)
      v = -v;
    if (it < 14 * 14)
      v = edge[it % 14] * (1 + (it / 14) * 977);
    const int cbits = (rng() % 4) ? 16 : 26;
    int64_t c = (int64_t)(rng() & (((uint64_t)1 << cbits) - 1));
    if (bitlen64((uint64_t)(v < 0 ? -v : v)) + bitlen64((uint64_t)c) > 52)
      c &= 0xffff;
    const double vd = ArithF64::from_i64(v);
    // FixedPoint *= constant
    bad[0] += ArithF64::to_i64(ArithF64::mulc(vd, ArithF64::coef(c))) != ArithI64::mulc(v, ArithI64::coef(c));
    // FixedPoint::round()
    bad[1] += ArithF64::to_i64(ArithF64::round_int(vd)) != ArithI64::round_int(v);
    // arithmetic shift
    const int sh = (int)(rng() % 12);
    bad[2] += ArithF64::to_i64(ArithF64::shr(vd, sh)) != ArithI64::shr(v, sh);
    // FixedPoint = int
    const int32_t iv = (int32_t)(rng() % 2000001) - 1000000;
    bad[3] += ArithF64::to_i64(ArithF64::from_int(iv)) != ArithI64::from_int(iv);
    // quantiser: every qp, coefficients of up to 22 bits
    const Quantizer q = make_quantizer(4 + (int)(rng() % 96));
    int64_t co = (int64_t)(rng() & (((uint64_t)1 << (1 + rng() % 22)) - 1));
    if (rng() & 1)
      co = -co;
    bad[4] += ArithF64::quantize(ArithF64::quant(q), (double)co) != ArithI64::quantize(ArithI64::quant(q), co);
    // de-quantiser: levels whose scaled value stays below 2^35
    int32_t lv = (int32_t)(rng() & 0xfffff) - 0x80000;
    while ((double)(lv < 0 ? -lv : lv) * q.step > 8.0e9)
      lv /= 2;
    bad[5] += ArithF64::to_i64(ArithF64::dequant_fp(ArithF64::quant(q), lv)) != ArithI64::dequant_fp(ArithI64::quant(q), lv);
    // small multipliers of the prediction
    const int m = 1 + (int)(rng() % 25);
    bad[6] += ArithF64::to_i64(ArithF64::muli(vd, m)) != ArithI64::muli(v, m);
    bad[7] += ArithF64::to_small((double)iv) != (int64_t)iv;
  }
  long total = 0;
  const char* name[8] = {"mulc", "round_int", "shr", "from_int", "quantize", "dequant_fp", "muli", "to_small"};
  for (int i = 0; i < 8; i++) {
    printf("%s %ld\n", name[i], bad[i]);
    total += bad[i];
  }
  return total ? 1 : 0;
}
