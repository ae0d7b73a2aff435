// Reference-side harness (test infrastructure, built only where /root/reference exists): calls the
// reference's OWN dependency-free headers where they lie, third_party/half_sourceforge_net/half.hpp
// and runtime/hash_table/compressor/fake_quantizer.h, so the oracle's restatement of the serving
// compressors (float_compressor.cc:63-109) and of stochastic_round (stochastic_rounding.h:27-37)
// can be checked against them. No reference source is copied.
#include <cstdint>
#include <limits>

#include "monolith/native_training/runtime/hash_table/compressor/fake_quantizer.h"
#include "third_party/half_sourceforge_net/half.hpp"

extern "C" {

// float_compressor.cc:69-82: Encode (half_float::half x(num)) then Decode (float(half))
void ref_fp16_roundtrip(const float* in, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    half_float::half x(in[i]);
    out[i] = x;
  }
}

// stochastic_rounding.h:28-31
void ref_half_up_down(const float* in, float* up, float* down, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    unsigned int hu = half_float::detail::float2half<std::round_toward_infinity>(in[i]);
    unsigned int hd = half_float::detail::float2half<std::round_toward_neg_infinity>(in[i]);
    up[i] = half_float::detail::half2float<float>(hu);
    down[i] = half_float::detail::half2float<float>(hd);
  }
}

// float_compressor.cc:93-105
void ref_fixed_r8(float r, const float* in, float* out, int64_t n) {
  monolith::hash_table::FakeQuantizer q(r);
  for (int64_t i = 0; i < n; ++i) out[i] = q.IntegerToFloat(q.QuantizeToInteger(in[i]));
}

}  // extern "C"
