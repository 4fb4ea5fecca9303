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
