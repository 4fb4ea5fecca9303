/* symbols_oracle.c -- TEST INFRASTRUCTURE: plain-C restatement of the zero-run
 * formation in the reference's entropy loops (the part of the loops that is
 * not the arithmetic coder itself):
 *   RAHT     tmc3/AttributeEncoder.cpp:1279-1291 (c = 1), :1347-1362 (c = 3),
 *            coefficients planar [c][n]
 *   lifting  tmc3/AttributeEncoder.cpp:1458-1474 (c = 3), :1617-1633 (c = 1),
 *            values interleaved [n][c] in coding order
 * A position whose c values are all zero extends the current run; any other
 * position emits (run, values) and resets the run; a non-empty run at the end
 * is emitted alone.  Pinned to the reference by feeding the result to the
 * reference's own PCCResidualsEncoder (oracle/ref_entropy_harness.cpp) and
 * comparing the bytes with the reference operator's payload
 * (tests/test_symbols.py). */
#include <stdint.h>

int
oracle_zero_run_pack(
  const int32_t* coeffs, int32_t n, int32_t c, int32_t planar, int32_t* runs,
  int32_t* values, int32_t* trailing_run)
{
  int m = 0, run = 0;
  for (int i = 0; i < n; i++) {
    int32_t v[3] = {0, 0, 0};
    int any = 0;
    for (int d = 0; d < c; d++) {
      v[d] = planar ? coeffs[(int64_t)n * d + i] : coeffs[(int64_t)i * c + d];
      any |= v[d] != 0;
    }
    if (!any) {
      run++;
    } else {
      runs[m] = run;
      for (int d = 0; d < c; d++)
        values[(int64_t)m * c + d] = v[d];
      m++;
      run = 0;
    }
  }
  *trailing_run = run;
  return m;
}

/* ---- binarisation of the symbols --------------------------------------------
 * Plain-C restatement of the decisions PCCResidualsEncoder makes for a symbol
 * stream (tmc3/AttributeEncoder.cpp:227-307, exp-Golomb forms
 * tmc3/entropyutils.h:142-183): one byte per decision, (context << 1) | bin,
 * contexts numbered in declaration order (tmc3/AttributeCommon.h:54-57):
 * 0..4 ctxRunLen, 5..18 ctxCoeffGtN[2][7], 19..24 ctxCoeffRemPrefix[2][3],
 * 25..30 ctxCoeffRemSuffix[2][3], 31 bypass.  Pinned by feeding the decisions
 * to the reference's own arithmetic coder and context models
 * (ref_entropy_encode_bins) and comparing the bytes with what the reference
 * class produces for the same symbols (tests/test_bins.py). */
static int64_t
ob_put(uint8_t* out, int64_t cap, int64_t n, int ctx, int bin)
{
  if (out && n < cap)
    out[n] = (uint8_t)((ctx << 1) | (bin & 1));
  return n + 1;
}

static int64_t
ob_run(uint8_t* out, int64_t cap, int64_t n, int run)
{
  int ctx = 0;
  int ones = run < 3 ? run : 3;
  for (int i = 0; i < ones; i++)
    n = ob_put(out, cap, n, ctx++, 1);
  if (run < 3)
    return ob_put(out, cap, n, ctx, 0);
  run -= 3;
  int prefix = run >> 1;
  if (prefix > 4)
    prefix = 4;
  for (int i = 0; i < prefix; i++)
    n = ob_put(out, cap, n, 3, 1);
  if (run < 8) {
    n = ob_put(out, cap, n, 3, 0);
    return ob_put(out, cap, n, 31, run & 1);
  }
  uint32_t sym = (uint32_t)(run - 8);
  int k = 2;
  for (; sym >= (1u << k); k++) {
    n = ob_put(out, cap, n, 4, 1);
    sym -= 1u << k;
  }
  n = ob_put(out, cap, n, 4, 0);
  for (int b = k - 1; b >= 0; b--)
    n = ob_put(out, cap, n, 31, (sym >> b) & 1);
  return n;
}

static int64_t
ob_symbol(uint8_t* out, int64_t cap, int64_t n, uint32_t value, int k1, int k2, int k3)
{
  n = ob_put(out, cap, n, 5 + k1, value > 0);
  if (value == 0)
    return n;
  n = ob_put(out, cap, n, 12 + k2, value > 1);
  if (value == 1)
    return n;
  value -= 2;
  int k = 1;
  for (; value >= (1u << k); k++) {
    int p = k - 1;
    n = ob_put(out, cap, n, 19 + 3 * k3 + (p > 2 ? 2 : p), 1);
    value -= 1u << k;
  }
  n = ob_put(out, cap, n, 19 + 3 * k3 + (k - 1 > 2 ? 2 : k - 1), 0);
  for (int b = k - 1; b >= 0; b--)
    n = ob_put(out, cap, n, 25 + 3 * k3 + (b > 2 ? 2 : b), (value >> b) & 1);
  return n;
}

int64_t
oracle_binarise_symbols(
  const int32_t* runs, const int32_t* values, int32_t num_symbols, int32_t trailing_run,
  int32_t c, uint8_t* bins, int64_t cap)
{
  int64_t n = 0;
  for (int s = 0; s < num_symbols; s++) {
    n = ob_run(bins, cap, n, runs[s]);
    if (c == 3) {
      const int32_t* v = values + 3 * (int64_t)s;
      uint32_t m0 = (uint32_t)(v[0] < 0 ? -v[0] : v[0]);
      uint32_t m1 = (uint32_t)(v[1] < 0 ? -v[1] : v[1]);
      uint32_t m2 = (uint32_t)(v[2] < 0 ? -v[2] : v[2]);
      int b0 = m1 == 0, b1 = m1 <= 1, b2 = m2 == 0, b3 = m2 <= 1;
      n = ob_symbol(bins, cap, n, m1, 0, 0, 1);
      n = ob_symbol(bins, cap, n, m2, 1 + b0, 1 + b1, 1);
      n = ob_symbol(bins, cap, n, (b0 && b2) ? m0 - 1 : m0, 3 + 2 * b0 + b2, 3 + 2 * b1 + b3, 0);
      for (int d = 0; d < 3; d++)
        if (v[d])
          n = ob_put(bins, cap, n, 31, v[d] < 0);
    } else {
      int32_t v = values[s];
      n = ob_symbol(bins, cap, n, (uint32_t)((v < 0 ? -v : v) - 1), 0, 0, 0);
      n = ob_put(bins, cap, n, 31, v < 0);
    }
  }
  if (trailing_run)
    n = ob_run(bins, cap, n, trailing_run);
  return n;
}
