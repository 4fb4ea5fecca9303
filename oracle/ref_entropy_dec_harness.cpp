// ref_entropy_dec_harness.cpp -- TEST INFRASTRUCTURE, not product code.
//
// The reference keeps its residual entropy DEcoder (class PCCResidualsDecoder)
// private to tmc3/AttributeDecoder.cpp.  This translation unit INCLUDES that
// source file where it lies under /root/reference (nothing is copied) so that
// the tests can recover the symbol stream -- the `values` of every predictor
// in coding order, zero runs expanded, exactly what the loops at
// AttributeDecoder.cpp:356-366 / 480-490 read -- from the payload the
// reference operator wrote.  That pins the encoder side of the predicting
// transform's oracle (symbols, not only the reconstruction).  Built into its
// own shared object (oracle/_ref/libtmc3_entropy_dec.so), AttributeDecoder.o
// excluded.
#include "AttributeDecoder.cpp"

#include <cstdint>
#include <cstring>

extern "C" {

// buf/len: the arithmetic-coded part of an attribute brick payload (behind
// the brick header).  values [num_points][c] out.
int
ref_entropy_decode_symbols(
  int32_t c, int32_t num_points, const uint8_t* buf, int32_t len, int32_t* values)
{
  using namespace pcc;
  SequenceParameterSet sps;
  sps.cabac_bypass_stream_enabled_flag = false;
  sps.entropy_continuation_enabled_flag = false;
  sps.bypass_bin_coding_without_prob_update = false;
  AttributeBrickHeader abh;
  AttributeContexts ctx;
  ctx.reset();
  PCCResidualsDecoder decoder(abh, ctx);
  decoder.start(sps, reinterpret_cast<const char*>(buf), len);
  int zeroRunRem = 0;
  for (int i = 0; i < num_points; i++) {
    int32_t v[3] = {0, 0, 0};
    if (--zeroRunRem < 0)
      zeroRunRem = decoder.decodeRunLength();
    if (!zeroRunRem) {
      if (c == 3)
        decoder.decode(v);
      else
        v[0] = decoder.decode();
    }
    for (int k = 0; k < c; k++)
      values[size_t(i) * c + k] = v[k];
  }
  decoder.stop();
  return 0;
}

}  // extern "C"
