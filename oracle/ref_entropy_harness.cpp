// ref_entropy_harness.cpp -- TEST INFRASTRUCTURE, not product code.
//
// The reference keeps its residual entropy coder (class PCCResidualsEncoder)
// private to tmc3/AttributeEncoder.cpp.  This translation unit INCLUDES that
// source file where it lies under /root/reference (nothing is copied), so
// that the tests can feed a symbol stream -- zero runs and coefficient
// tuples, exactly the calls of the entropy loops at
// AttributeEncoder.cpp:1279-1291 / 1347-1362 (RAHT) and :1458-1474 / :1617-1633
// (lifting) -- to the reference's own arithmetic coder and compare the bytes
// with the payload of the reference operator.  Built into its own shared
// object (oracle/_ref/libtmc3_entropy.so) together with the other reference
// objects, AttributeEncoder.o excluded.
#include "AttributeEncoder.cpp"

#include <cstdint>
#include <cstring>

extern "C" {

// returns the number of arithmetic-coded bytes written to out (<= cap), or
// -1 if out is too small
int
ref_entropy_encode_symbols(
  int32_t c, int32_t num_points, const int32_t* runs, const int32_t* values,
  int32_t num_symbols, int32_t trailing_run, uint8_t* out, int32_t cap)
{
  using namespace pcc;
  SequenceParameterSet sps;
  sps.cabac_bypass_stream_enabled_flag = false;
  sps.entropy_continuation_enabled_flag = false;
  sps.bypass_bin_coding_without_prob_update = false;
  AttributeParameterSet aps;
  aps.max_num_direct_predictors = 0;
  aps.direct_avg_predictor_disabled_flag = false;
  AttributeBrickHeader abh;
  AttributeContexts ctx;
  ctx.reset();
  PCCResidualsEncoder encoder(aps, abh, ctx);
  encoder.start(sps, num_points);
  for (int k = 0; k < num_symbols; k++) {
    encoder.encodeRunLength(runs[k]);
    if (c == 3)
      encoder.encode(values[3 * k], values[3 * k + 1], values[3 * k + 2]);
    else
      encoder.encode(values[k]);
  }
  if (trailing_run)
    encoder.encodeRunLength(trailing_run);
  const int len = encoder.stop();
  if (len > cap)
    return -1;
  memcpy(out, encoder.arithmeticEncoder.buffer(), size_t(len));
  return len;
}

}  // extern "C"
