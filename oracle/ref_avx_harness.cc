// Thin C wrapper (our code) around the reference's dependency-free header
// monolith/native_training/runtime/hash_table/optimizer/avx_utils.h, compiled from where it lies
// under /root/reference (never copied). Output: oracle/_ref/libavx_ref.so. Build container only.
#include <cmath>
#include <cstddef>
#include "monolith/native_training/runtime/hash_table/optimizer/avx_utils.h"

extern "C" {
void ref_baseline_adagrad(float* num, float* norm, const float* grad, size_t len, float lr, float wd) {
  monolith::hash_table::BaselineAdagradOptimize(num, norm, grad, len, lr, wd);
}
void ref_adagrad(float* num, float* norm, const float* grad, size_t len, float lr, float wd) {
  // With -D_ENABLE_AVX -mavx -mfma this is Avx256AdagradOptimize (the production build flags,
  // optimizer/BUILD:94-109).
  monolith::hash_table::AdagradOptimize(num, norm, grad, len, lr, wd);
}
void ref_reduce_sum(const float* a, const float* b, float* out, size_t len) {
  monolith::hash_table::ReduceSum(a, b, out, len);
}
}
