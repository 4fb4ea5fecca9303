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


// The reference's arithmetic coder and context models driven by a stream of
// binary decisions, one byte each: (context << 1) | bin, contexts in the
// declaration order of AttributeContexts (AttributeCommon.h:54-57), 31 = bypass
// -- what gpcc_binarise_symbols / oracle_binarise_symbols produce.
namespace {
struct BinsEncoder : pcc::PCCResidualsEncoder {
  using pcc::PCCResidualsEncoder::PCCResidualsEncoder;
  pcc::AdaptiveBitModel& model(int id)
  {
    if (id < 5)
      return ctxRunLen[id];
    if (id < 19)
      return ctxCoeffGtN[(id - 5) / 7][(id - 5) % 7];
    if (id < 25)
      return ctxCoeffRemPrefix[(id - 19) / 3][(id - 19) % 3];
    return ctxCoeffRemSuffix[(id - 25) / 3][(id - 25) % 3];
  }
};
}  // namespace

int
ref_entropy_encode_bins(
  const uint8_t* bins, int64_t num_bins, int32_t num_points, uint8_t* out, int32_t cap)
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
  BinsEncoder encoder(aps, abh, ctx);
  encoder.start(sps, num_points);
  for (int64_t i = 0; i < num_bins; i++) {
    const int id = bins[i] >> 1, bin = bins[i] & 1;
    if (id == 31)
      encoder.arithmeticEncoder.encode(bin);
    else
      encoder.arithmeticEncoder.encode(bin, encoder.model(id));
  }
  const int len = encoder.stop();
  if (len > cap)
    return -1;
  memcpy(out, encoder.arithmeticEncoder.buffer(), size_t(len));
  return len;
}

}  // extern "C"
