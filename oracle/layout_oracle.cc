// layout_oracle.cc — TEST INFRASTRUCTURE ONLY (see oracle.cc header).
//
// Sequential restatement of MonolithEmbeddingToLayout(V2) and its gradient:
//   monolith/native_training/runtime/ops/fused_embedding_to_layout.cc:403-695 (forward),
//   :703-976 (gradient). Layout / slice flattening (the constructor, :246-334, and the Layout
//   classes' GetSlice, :92-242) is done by the caller into OrcLayoutSlice records, in the
//   reference's order (layouts by sorted name, slices in config order).
// Parity: pinned by tests/test_layout_oracle.py against the reference test's own truth
// algorithm (fused_embedding_to_layout_test.py:90-115,435-481) on seeded inputs, and the
// gradient by the reference's "grad == use count" assertions (:682-694).
#include <cstdint>
#include <cstring>
#include <vector>

extern "C" {

struct OrcLayoutSlice {  // == mono_layout_slice
  int32_t feature_idx, start, dim, pooling, max_seq, out_index, out_stride, out_offset, addn, first;
};
enum { POOL_SUM = 0, POOL_MEAN = 1, POOL_FIRSTN = 2 };

}  // extern "C"

namespace {

struct Nfl {
  bool shared;
  int offset, feature_num;
};

// GetFeatureInfo (:369-383)
Nfl feature_info(int idx, const uint32_t* nfl, int n_nfl, int64_t n_features) {
  Nfl r;
  r.shared = nfl[idx] >> 31;
  r.offset = (int)(nfl[idx] & 0x7fffffffu);
  if (idx < n_nfl - 1)
    r.feature_num = (int)(nfl[idx + 1] & 0x7fffffffu) - r.offset;
  else
    r.feature_num = (int)n_features - r.offset;
  return r;
}

inline int fid_count(int f, const int32_t* fo, int64_t n_features, int64_t n_fids) {
  return f < n_features - 1 ? fo[f + 1] - fo[f] : (int)n_fids - fo[f];
}

// OptimizedSumpooling (:385-400): first touch copies, later touches add
inline void sum_pool(const float* src, int dim, bool* init, float* dst) {
  if (*init) {
    std::memcpy(dst, src, sizeof(float) * dim);
    *init = false;
  } else {
    for (int i = 0; i < dim; ++i) dst[i] = src[i] + dst[i];
  }
}

}  // namespace

extern "C" {

// Forward. Outputs are zero-filled here (:509).
void orc_embedding_to_layout(const OrcLayoutSlice* slices, int n_slices,
                             const float* const* emb_ptrs, const int32_t* emb_dims, int n_bufs,
                             const uint64_t* fid_offset, int64_t n_fids,
                             const int32_t* feature_offset, int64_t n_features,
                             const uint32_t* nfl_offset, int n_nfl, int32_t batch,
                             float* const* outs, const int64_t* out_floats, int n_outs) {
  (void)n_bufs;
  for (int o = 0; o < n_outs; ++o) std::memset(outs[o], 0, sizeof(float) * out_floats[o]);
  for (int si = 0; si < n_slices; ++si) {  // gather_emb_fn (:535-601), layouts then slices
    const OrcLayoutSlice& sc = slices[si];
    const int dim = sc.dim;
    float* base = outs[sc.out_index] + sc.out_offset;
    Nfl nf = feature_info(sc.feature_idx, nfl_offset, n_nfl, n_features);
    if (!nf.feature_num) continue;
    std::vector<float> tmp;
    const bool use_tmp = sc.addn && (sc.pooling == POOL_MEAN || nf.shared);  // :554-559
    if (use_tmp) tmp.assign(dim, 0.f);
    int feature_idx = nf.offset;
    for (int index = 0; index < batch; ++index) {
      float* dst = base + (int64_t)index * sc.out_stride;
      if (!nf.shared || index == 0) {
        bool init = !sc.addn || use_tmp;
        float* out_ptr = use_tmp ? tmp.data() : dst;
        // GatherEmb (:624-676)
        const int fnum = fid_count(feature_idx, feature_offset, n_features, n_fids);
        if (fnum != 0) {
          const int p0 = feature_offset[feature_idx];
          int seq_idx = 0;
          for (int q = 0; q < fnum; ++q) {
            const uint64_t fo = fid_offset[p0 + q];
            const int32_t i1 = (int32_t)(fo >> 32), i2 = (int32_t)(fo << 32 >> 32);
            const float* src = emb_ptrs[i1] + (int64_t)i2 * emb_dims[i1] + sc.start;
            if (sc.pooling == POOL_FIRSTN) {
              if (seq_idx < sc.max_seq) std::memcpy(out_ptr + seq_idx * dim, src, sizeof(float) * dim);
              seq_idx++;
            } else {
              sum_pool(src, dim, &init, out_ptr);
            }
          }
          if (sc.pooling == POOL_MEAN)
            for (int mi = 0; mi < dim; ++mi) out_ptr[mi] = out_ptr[mi] / fnum;
        }
        if (use_tmp) {
          bool init_tmp = sc.first;
          sum_pool(tmp.data(), dim, &init_tmp, dst);
        }
        feature_idx++;
      } else {
        if (use_tmp) {
          bool init_tmp = sc.first;
          sum_pool(tmp.data(), dim, &init_tmp, dst);
        } else {  // CopyEmb (:678-694)
          const int n = sc.pooling == POOL_FIRSTN ? dim * sc.max_seq : dim;
          std::memcpy(dst, base, sizeof(float) * n);
        }
      }
    }
  }
}

// Gradient. emb_grads are zero-filled here (:764,801). slice_idx_of[si] = index of the slice
// inside its feature (init flags are per (unique fid, slice_idx), :815,922).
void orc_embedding_to_layout_grad(const OrcLayoutSlice* slices, const int32_t* slice_idx_of,
                                  int n_slices, int max_slice_num, float* const* emb_grads,
                                  const int32_t* emb_dims, const int64_t* emb_floats, int n_bufs,
                                  const uint64_t* fid_offset, int64_t n_fids,
                                  const int32_t* feature_offset, int64_t n_features,
                                  const uint32_t* nfl_offset, int n_nfl, int32_t batch,
                                  const float* const* out_grads, int mean_divides) {
  std::vector<int64_t> ufid_base(n_bufs);
  int64_t init_counter = 0;
  for (int i = 0; i < n_bufs; ++i) {
    std::memset(emb_grads[i], 0, sizeof(float) * emb_floats[i]);
    ufid_base[i] = init_counter;
    init_counter += emb_dims[i] ? emb_floats[i] / emb_dims[i] : 0;
  }
  std::vector<char> init((size_t)init_counter * max_slice_num, 1);  // GroupA (:59-65)
  for (int si = 0; si < n_slices; ++si) {  // scatter_grad_fn (:844-880)
    const OrcLayoutSlice& sc = slices[si];
    const int dim = sc.dim;
    const float* base = out_grads[sc.out_index] + sc.out_offset;
    Nfl nf = feature_info(sc.feature_idx, nfl_offset, n_nfl, n_features);
    if (!nf.feature_num) continue;
    int feature_idx = nf.offset;
    for (int index = 0; index < batch; ++index) {
      const float* grad_ptr = base + (int64_t)index * sc.out_stride;
      // ScatterGrad (:898-975)
      const int fnum = fid_count(feature_idx, feature_offset, n_features, n_fids);
      if (fnum != 0) {
        const int p0 = feature_offset[feature_idx];
        int seq_idx = 0;
        std::vector<float> scaled(dim);
        for (int q = 0; q < fnum; ++q) {
          const uint64_t fo = fid_offset[p0 + q];
          const int32_t i1 = (int32_t)(fo >> 32), i2 = (int32_t)(fo << 32 >> 32);
          float* dst = emb_grads[i1] + (int64_t)i2 * emb_dims[i1] + sc.start;
          char* ip = &init[(size_t)(ufid_base[i1] + i2) * max_slice_num + slice_idx_of[si]];
          bool flag = *ip != 0;
          if (sc.pooling == POOL_SUM) {
            sum_pool(grad_ptr, dim, &flag, dst);
          } else if (sc.pooling == POOL_MEAN) {
            // the reference computes grad / fid_num into a temporary and then scatters the
            // UNDIVIDED gradient (:929-937); mean_divides != 0 gives the mathematical gradient
            const float* g = grad_ptr;
            if (mean_divides) {
              for (int mi = 0; mi < dim; ++mi) scaled[mi] = grad_ptr[mi] / fnum;
              g = scaled.data();
            }
            sum_pool(g, dim, &flag, dst);
          } else {
            if (seq_idx < sc.max_seq) sum_pool(grad_ptr + seq_idx * dim, dim, &flag, dst);
            seq_idx++;
          }
          *ip = flag ? 1 : 0;
        }
      }
      if (!nf.shared) feature_idx++;
    }
  }
}

}  // extern "C"
