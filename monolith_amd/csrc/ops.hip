// ops.hip — routing / exchange / pooling kernels of the hot path for MI355X (gfx950).
//
//   F1  fused_reorder_by_indices      ops/fused_reorder_by_indices.cc:36-112
//   F2  fused_gather_embeddings(+grad) ops/map_id_to_embedding_op.cc:239-343, .cu.cc:30-119
//   E1  reduce_sum / reduce_mean(+grad) ops/reduce_op.cc:29-180
//   E2  fused_reduce_sum_and_split(+grad) ops/reduce_op.cc:232-320
//   E3  fused_sorted_segment_sum      ops/reduce_op.cu.cc:53-161
//   E4  first_n                       embedding_combiners.py:71-100
//
// All of it is HBM-bound integer / row-copy work. Every reduction is deterministic and sums in
// input order (the reference CPU order): duplicates are grouped with a stable radix sort and a
// segment is summed left to right, instead of the float atomics of the reference's CUDA kernels.
#include <algorithm>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <vector>

#include "mono_common.h"

namespace mono {

static Workspace g_ws;       // scratch of the stateless ops (one host thread per GPU)
static Workspace g_ws_sort;  // radix-sort / scan temporaries

constexpr int kMaxSplit = 64;

static int bits_for_u(uint64_t max_value) {
  int b = 1;
  while (b < 64 && (max_value >> b) != 0) ++b;
  return b;
}

// F2-gradient plan. The gradient of fused_gather_embeddings_by_input sums the gradient rows of
// every occurrence into the row of its unique id, in input order: a stable sort of (offset,
// occurrence) plus the segment classification, ~120 us of latency-bound work that depends on
// nothing but the offsets F1 has just produced. F1 therefore runs it on a stream of its own
// while the host waits for F1's sizes (the GPU would otherwise idle there), and hands out a token;
// the gradient call that presents the token and the very same offset arrays only launches the
// sum kernels. Two plans are kept, so an F1 issued one step ahead does not displace the plan of
// the step still to be differentiated. MONO_GRAD_PLAN=0 switches it off.
struct GradPlan {
  uint64_t token = 0;
  int M = 0;
  std::vector<const int32_t*> offs;
  std::vector<int64_t> lens;
  std::vector<int32_t> dims;
  std::vector<size_t> o_skey, o_sidx, o_q;  // byte offsets into ws, per slot
  int64_t covered = 0;  // floats [0, covered) each belong to the row of some occurrence
  Workspace ws;
  hipEvent_t ready = nullptr;
  // behind the last gradient that read this plan (another stream than the one that rebuilds it)
  hipEvent_t consumed = nullptr;
  bool consumed_set = false;
};
static GradPlan g_plans[2];
static uint64_t g_plan_seq = 0;
static hipStream_t g_plan_stream = nullptr;
static hipEvent_t g_plan_src = nullptr, g_ev_sizes = nullptr;
static int32_t* g_pinned = nullptr;
static size_t g_pinned_words = 0;

static bool grad_plan_enabled() {
  static const bool v = [] {
    const char* e = getenv("MONO_GRAD_PLAN");
    return !(e && atoi(e) == 0);
  }();
  return v;
}

static int pinned_words(size_t words, int32_t** out) {
  if (words > g_pinned_words) {
    if (g_pinned) MONO_HIP(hipHostFree(g_pinned));
    g_pinned = nullptr;
    g_pinned_words = 0;
    const size_t want = std::max<size_t>(words, 1024);
    MONO_HIP(hipHostMalloc((void**)&g_pinned, want * sizeof(int32_t), hipHostMallocDefault));
    g_pinned_words = want;
  }
  *out = g_pinned;
  return 0;
}

// Sort + classify every slot's offsets on the plan stream, behind `st`'s work so far.
static int grad_plan_build(const std::vector<int32_t*>& offs, const int64_t* h_lens,
                           const int32_t* h_dims, int M, int64_t total, hipStream_t st,
                           GradPlan** out) {
  *out = nullptr;
  if (!grad_plan_enabled()) return 0;
  int32_t max_dim = 1;
  for (int m = 0; m < M; ++m) max_dim = std::max(max_dim, h_dims[m]);
  if ((uint64_t)total * (uint64_t)max_dim >= 0x7FFFFFF0ull) return 0;
  if (!g_plan_stream) {
    MONO_HIP(hipStreamCreateWithFlags(&g_plan_stream, hipStreamNonBlocking));
    MONO_HIP(hipEventCreateWithFlags(&g_plan_src, hipEventDisableTiming));
  }
  GradPlan& p = g_plans[(g_plan_seq + 1) & 1];
  p.token = 0;
  if (!p.ready) MONO_HIP(hipEventCreateWithFlags(&p.ready, hipEventDisableTiming));
  const int end_bit = bits_for_u((uint64_t)total * (uint64_t)max_dim);
  p.M = M;
  p.offs.assign(offs.begin(), offs.end());
  p.lens.assign(h_lens, h_lens + M);
  p.dims.assign(h_dims, h_dims + M);
  p.o_skey.assign(M, 0);
  p.o_sidx.assign(M, 0);
  p.o_q.assign(M, 0);
  size_t bytes = 0, tmp_max = 0;
  for (int m = 0; m < M; ++m) {
    const size_t n_al = ((size_t)h_lens[m] + 63) & ~(size_t)63;
    p.o_skey[m] = bytes;
    bytes += n_al * sizeof(uint32_t);
    p.o_sidx[m] = bytes;
    bytes += n_al * sizeof(uint32_t);
    bytes = (bytes + 255) & ~(size_t)255;
    p.o_q[m] = bytes;
    bytes += h_lens[m] ? segment_plan_bytes(h_lens[m], h_dims[m]) : 0;
    if (h_lens[m]) tmp_max = std::max(tmp_max, sort_pairs_tmp_bytes((size_t)h_lens[m], end_bit));
  }
  tmp_max = (tmp_max + 255) & ~(size_t)255;
  MONO_TRY(p.ws.ensure(bytes + tmp_max + 256));
  char* base = (char*)p.ws.ptr;
  void* tmp = base + bytes;
  MONO_HIP(hipEventRecord(g_plan_src, st));
  MONO_HIP(hipStreamWaitEvent(g_plan_stream, g_plan_src, 0));
  // the slot is recycled: the gradient that used the plan two reorders back may still be reading it
  if (p.consumed_set) {
    MONO_HIP(hipStreamWaitEvent(g_plan_stream, p.consumed, 0));
    p.consumed_set = false;
  }
  for (int m = 0; m < M; ++m) {
    if (h_lens[m] == 0) continue;
    uint32_t* skey = (uint32_t*)(base + p.o_skey[m]);
    uint32_t* sidx = (uint32_t*)(base + p.o_sidx[m]);
    // keys = the float offsets themselves, values = occurrence positions (the sort's default)
    MONO_TRY(sort_pairs_u32((const uint32_t*)offs[m], skey, nullptr, sidx, (size_t)h_lens[m], end_bit,
                            tmp, tmp_max, g_plan_stream));
    MONO_TRY(segment_plan_classify(skey, sidx, h_lens[m], h_dims[m], base + p.o_q[m], g_plan_stream));
  }
  MONO_HIP(hipEventRecord(p.ready, g_plan_stream));
  *out = &p;
  return 0;
}
static int64_t g_last_plan_token = 0;

static int sort_pairs(const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout,
                      size_t n, int end_bit, hipStream_t st, Workspace* ws = &g_ws_sort) {
  const size_t tmp = sort_pairs_tmp_bytes(n, end_bit);
  MONO_TRY(ws->ensure(tmp));
  return sort_pairs_u32(kin, kout, vin, vout, n, end_bit, ws->ptr, tmp, st);
}

static int exclusive_scan_i32(const int32_t* in, int32_t* out, size_t n, hipStream_t st,
                              Workspace* ws = &g_ws_sort) {
  size_t tmp = 0;
  MONO_HIP(rocprim::exclusive_scan(nullptr, tmp, in, out, 0, n, rocprim::plus<int32_t>(), st));
  MONO_TRY(ws->ensure(tmp + 256));
  MONO_HIP(rocprim::exclusive_scan(ws->ptr, tmp, in, out, 0, n, rocprim::plus<int32_t>(), st));
  return 0;
}

// ------------------------------------------------------------------------------ sortedness
template <typename K>
__global__ void __launch_bounds__(256) k_check_sorted(const K* keys, int64_t n, int64_t limit,
                                                      int* flags) {
  // flags[0] |= 1 if unsorted, flags[0] |= 2 if a key is outside [0, limit)
  int bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    K k = keys[i];
    if (i > 0 && keys[i - 1] > k) bad |= 1;
    if (k < 0 || (int64_t)k >= limit) bad |= 2;
  }
  if (bad) atomicOr(flags, bad);
}

template <typename K>
static int check_keys(const K* keys, int64_t n, int64_t limit, int* h_flags, hipStream_t st) {
  MONO_TRY(g_ws.ensure(256));
  int* flags = (int*)g_ws.ptr;
  MONO_HIP(hipMemsetAsync(flags, 0, sizeof(int), st));
  if (n > 0)
    hipLaunchKernelGGL(k_check_sorted<K>, dim3(grid_for(n, 256, 2048)), dim3(256), 0, st, keys, n,
                       limit, flags);
  MONO_HIP(hipMemcpyAsync(h_flags, flags, sizeof(int), hipMemcpyDeviceToHost, st));
  MONO_HIP(hipStreamSynchronize(st));
  return 0;
}

template <typename K>
__global__ void __launch_bounds__(256) k_keys_to_u32(const K* in, int64_t n, uint32_t* keys,
                                                     uint32_t* idx) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    keys[i] = (uint32_t)in[i];
    idx[i] = (uint32_t)i;
  }
}

// A view of row ids that is sorted: either the caller's array (already sorted) or a stable
// sort of it with the permutation.
template <typename K>
struct SortedRows {
  const K* rows = nullptr;       // valid when perm == nullptr
  const uint32_t* urows = nullptr;  // valid when perm != nullptr
  const uint32_t* perm = nullptr;
};

template <typename K>
static int make_sorted_rows(const K* row_ids, int64_t n, int64_t batch, SortedRows<K>* out,
                            const char* who, hipStream_t st) {
  int flags = 0;
  MONO_TRY(check_keys<K>(row_ids, n, batch, &flags, st));
  MONO_REQUIRE(!(flags & 2), "%s: a row id is outside [0, %lld)", who, (long long)batch);
  if (!(flags & 1)) {
    out->rows = row_ids;
    return 0;
  }
  MONO_REQUIRE(n < (int64_t)0xFFFFFFF0ll, "%s: too many rows", who);
  const size_t n_al = ((size_t)n + 63) & ~(size_t)63;
  MONO_TRY(g_ws.ensure(256 + 4 * n_al * sizeof(uint32_t)));
  uint32_t* kin = (uint32_t*)((char*)g_ws.ptr + 256);
  uint32_t* iin = kin + n_al;
  uint32_t* kout = iin + n_al;
  uint32_t* iout = kout + n_al;
  hipLaunchKernelGGL(k_keys_to_u32<K>, dim3(grid_for(n, 256, 2048)), dim3(256), 0, st, row_ids, n,
                     kin, iin);
  MONO_TRY(sort_pairs(kin, kout, iin, iout, (size_t)n, bits_for_u((uint64_t)batch), st));
  out->urows = kout;
  out->perm = iout;
  return 0;
}

template <typename T>
__device__ inline int64_t lower_bound_dev(const T* a, int64_t n, int64_t v) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if ((int64_t)a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

struct SplitArgs {
  int M;
  int dims[kMaxSplit];
  float* outs[kMaxSplit];
};

// E1/E2/E3: out[b, :] = sum over the rows of segment b, left to right. One thread per output
// element; consecutive threads read consecutive floats of the same value row (coalesced).
template <typename K, bool MEAN>
__global__ void __launch_bounds__(256) k_segment_reduce(SortedRows<K> sr, const float* __restrict__ v,
                                                        int64_t n, int64_t dim, int64_t batch,
                                                        SplitArgs sp) {
  const int64_t total = batch * dim;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / dim, k = e - b * dim;
    int64_t lo, hi;
    if (sr.perm) {
      lo = lower_bound_dev(sr.urows, n, b);
      hi = lower_bound_dev(sr.urows, n, b + 1);
    } else {
      lo = lower_bound_dev(sr.rows, n, b);
      hi = lower_bound_dev(sr.rows, n, b + 1);
    }
    float acc = 0.f;
    for (int64_t p = lo; p < hi; ++p) {
      int64_t i = sr.perm ? (int64_t)sr.perm[p] : p;
      acc += v[i * dim + k];
    }
    if (MEAN) {
      // reduce_op.cc:81: float multiply = 1.0 / static_cast<float>(count)
      float mul = (float)(1.0 / (double)(float)(uint64_t)(hi - lo));
      acc *= mul;
    }
    int m = 0;
    int64_t kk = k;
    while (m + 1 < sp.M && kk >= sp.dims[m]) {
      kk -= sp.dims[m];
      ++m;
    }
    sp.outs[m][b * sp.dims[m] + kk] = acc;
  }
}

// E1 gradients: out[i, :] = grads[row[i], :] (* 1/count(row[i]) for mean); E2 gradient gathers
// from M split tensors.
template <typename K, bool MEAN>
__global__ void __launch_bounds__(256) k_segment_gather(const K* __restrict__ row_ids,
                                                        SortedRows<K> sr, int64_t n, int64_t dim,
                                                        SplitArgs sp, float* __restrict__ out) {
  const int64_t total = n * dim;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / dim, k = e - i * dim;
    const int64_t b = (int64_t)row_ids[i];
    int m = 0;
    int64_t kk = k;
    while (m + 1 < sp.M && kk >= sp.dims[m]) {
      kk -= sp.dims[m];
      ++m;
    }
    float g = sp.outs[m][b * sp.dims[m] + kk];
    if (MEAN) {
      int64_t lo, hi;
      if (sr.perm) {
        lo = lower_bound_dev(sr.urows, n, b);
        hi = lower_bound_dev(sr.urows, n, b + 1);
      } else {
        lo = lower_bound_dev(sr.rows, n, b);
        hi = lower_bound_dev(sr.rows, n, b + 1);
      }
      g *= (float)(1.0 / (double)(float)(uint64_t)(hi - lo));
    }
    out[e] = g;
  }
}

// E4: out[b, pos, :] = v[i, :] for the first max_seq rows of each (sorted) segment.
__global__ void __launch_bounds__(256) k_first_n(const int64_t* __restrict__ row_ids,
                                                 const float* __restrict__ v, int64_t n, int64_t dim,
                                                 int64_t max_seq, float* __restrict__ out) {
  const int64_t total = n * dim;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / dim, k = e - i * dim;
    const int64_t b = row_ids[i];
    const int64_t pos = i - lower_bound_dev(row_ids, n, b);
    if (pos < max_seq) out[(b * max_seq + pos) * dim + k] = v[e];
  }
}

// F2 forward: out[j, k] = fused[off[j] + k].
__global__ void __launch_bounds__(256) k_gather_rows(const float* __restrict__ fused,
                                                     const int32_t* __restrict__ off, int64_t len,
                                                     int dim, float* __restrict__ out) {
  const int64_t total = len * dim;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = e / dim;
    const int k = (int)(e - j * dim);
    out[e] = fused[(int64_t)off[j] + k];
  }
}

// ... four floats per thread: every dim of the call is a multiple of 4 (so every offset is) and
// the buffers are 16-byte aligned. 32-bit index arithmetic (the host checks the range).
__global__ void __launch_bounds__(256) k_gather_rows4(const float4* __restrict__ fused,
                                                      const int32_t* __restrict__ off, uint32_t len,
                                                      uint32_t quads, float4* __restrict__ out) {
  const uint32_t total = len * quads;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const uint32_t j = e / quads, k = e - j * quads;
    out[e] = fused[((uint32_t)off[j] >> 2) + k];
  }
}

// ------------------------------------------------------------------------------ F1 reorder
struct ReorderArgs {
  int M, N, shard_rule;
  int64_t total;
};

struct SlotMeta {  // per merged slot, device array
  const int64_t* ids;
  int32_t* out_off;
  int64_t base;       // first global occurrence index of this slot
  int64_t len;
  int64_t hash_base;  // first cell of this slot's dedup region
  uint64_t hash_mask; // region size - 1 (power of two); cell [hash_base + size] = key -1
  int32_t dim;
  int64_t head_base;  // first index of this slot's head (its first 1/64) among all slots' heads
};

__device__ inline int find_slot_of(const SlotMeta* sm, int M, int64_t gidx) {
  int lo = 0, hi = M - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (sm[mid].base <= gidx) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__device__ inline int64_t floormod_dev(int64_t a, int64_t b) {
  int64_t r = a % b;
  return r < 0 ? r + b : r;
}

// Pass 1: per-slot dedup set in HBM, open addressing over 4-byte cells. A cell does not hold the
// key but the global index of an occurrence of it (its key is ids[index]), and that index only
// ever decreases, to the key's FIRST occurrence: claiming an empty cell (one 32-bit CAS) already
// records the occurrence, so a key seen once costs one atomic (with {key, first} cells it cost a
// 64-bit CAS plus an atomicMin, and device-scope atomics on random words are the bound here,
// ~20 G/s); the set is a quarter of the size, and so is the fill that resets it.
__device__ __forceinline__ void reorder_insert_one(const SlotMeta& sl, int64_t g, uint32_t* hfirst,
                                                   uint32_t* hcell) {
  const int64_t s_base = sl.base, s_hash_base = sl.hash_base;
  const uint64_t s_mask = sl.hash_mask;
  const int64_t* __restrict__ s_ids = sl.ids;
  const int64_t key = s_ids[g - s_base];
  uint64_t p = mix64((uint64_t)key) & s_mask;
  for (;;) {
    uint32_t* cp = hfirst + (s_hash_base + (int64_t)p);
    uint32_t cur = *cp;
    if (cur == 0xFFFFFFFFu) {
      cur = atomicCAS(cp, 0xFFFFFFFFu, (uint32_t)g);
      if (cur == 0xFFFFFFFFu) break;  // claimed: the cell says "first seen at g"
    }
    // the cell belongs to the key of occurrence `cur` (same slot: regions are per slot); a
    // later atomicMin can only replace cur by another occurrence of that same key
    if (s_ids[(int64_t)cur - s_base] == key) {
      // first only decreases: an occurrence that already sees a smaller index has nothing to
      // add, which spares the hot keys tens of thousands of serialised atomics on one word. A
      // stale value can only be LARGER than the current one, i.e. cause a redundant atomic.
      if (cur > (uint32_t)g) atomicMin(cp, (uint32_t)g);
      break;
    }
    p = (p + 1) & s_mask;
  }
  hcell[g] = (uint32_t)(s_hash_base + (int64_t)p);
}

__global__ void __launch_bounds__(256) k_reorder_insert(ReorderArgs a, const SlotMeta* sm,
                                                        uint32_t* hfirst, uint32_t* hcell,
                                                        int phase, int64_t heads_total) {
  // Two launches. Phase 0 inserts only the first 1/64 of every slot's occurrences, phase 1 the
  // rest. A hot key (thousands of occurrences) almost surely occurs in that head, so by the time
  // the bulk runs its cell already holds a small index and the guarded atomicMin is skipped; in
  // one launch ~10 k concurrent occurrences of one key would all see a large index and
  // serialise on a single word (~90 atomics / us). Phase 0 walks the heads only (packed by
  // SlotMeta::head_base) instead of skipping 63 of every 64 occurrences.
  if (phase == 0) {
    for (int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; h < heads_total;
         h += (int64_t)gridDim.x * blockDim.x) {
      int lo = 0, hi = a.M - 1;
      while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (sm[mid].head_base <= h) lo = mid; else hi = mid - 1;
      }
      reorder_insert_one(sm[lo], sm[lo].base + (h - sm[lo].head_base), hfirst, hcell);
    }
    return;
  }
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < a.total;
       g += (int64_t)gridDim.x * blockDim.x) {
    const int m = find_slot_of(sm, a.M, g);
    if ((g - sm[m].base) < ((sm[m].len + 63) >> 6)) continue;  // a head: phase 0 took it
    reorder_insert_one(sm[m], g, hfirst, hcell);
  }
}

// Pass 2: sort key of every occurrence = shard * M + slot for first occurrences, sentinel N*M
// for the rest; the stable sort then yields exactly the reference's output order.
__global__ void __launch_bounds__(256) k_reorder_keys(ReorderArgs a, const SlotMeta* sm,
                                                      const uint32_t* hfirst, const uint32_t* hcell,
                                                      uint32_t* skey, uint32_t* sidx) {
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < a.total;
       g += (int64_t)gridDim.x * blockDim.x) {
    const int m = find_slot_of(sm, a.M, g);
    uint32_t key = (uint32_t)(a.N * a.M);
    if (hfirst[hcell[g]] == (uint32_t)g) {
      const int64_t id = sm[m].ids[g - sm[m].base];
      // distribution_ops.py:255-272
      const int shard = a.shard_rule == 1 ? (int)(1 + floormod_dev(id, a.N - 1))
                                          : (int)floormod_dev(id, a.N);
      key = (uint32_t)(shard * a.M + m);
    }
    skey[g] = key;
    sidx[g] = (uint32_t)g;
  }
}

__global__ void __launch_bounds__(256) k_reorder_dims(ReorderArgs a, const SlotMeta* sm,
                                                      const uint32_t* skey, int32_t* dims) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.total;
       r += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t key = skey[r];
    dims[r] = key < (uint32_t)(a.N * a.M) ? sm[key % a.M].dim : 0;
  }
}

// sharded_slot_sizes[n * M + m] = number of sorted keys equal to n * M + m (binary search on
// the sorted keys: no counter contention whatever the skew).
__global__ void __launch_bounds__(256) k_reorder_hist(ReorderArgs a, const uint32_t* skey,
                                                      int32_t* hist) {
  const int cells = a.N * a.M;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += gridDim.x * blockDim.x)
    hist[c] = (int32_t)(lower_bound_dev(skey, a.total, (int64_t)c + 1) -
                        lower_bound_dev(skey, a.total, (int64_t)c));
}

__global__ void __launch_bounds__(256) k_reorder_emit(ReorderArgs a, const SlotMeta* sm,
                                                      const uint32_t* skey, const uint32_t* sidx,
                                                      const int32_t* emb_off, int uniform_dim,
                                                      const uint32_t* hcell, uint32_t* hfirst,
                                                      int64_t* out_ids) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.total;
       r += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t key = skey[r];
    if (key >= (uint32_t)(a.N * a.M)) continue;
    const uint32_t g = sidx[r];
    const SlotMeta& s = sm[key % a.M];
    out_ids[r] = s.ids[(int64_t)g - s.base];
    // the cell now carries the embedding offset: the exclusive scan of the dims, or, when every
    // slot has the same dim, simply rank * dim (unique ids sort to the front, the rest to a
    // sentinel key at the end)
    hfirst[hcell[g]] = emb_off ? (uint32_t)emb_off[r] : (uint32_t)r * (uint32_t)uniform_dim;
  }
}

__global__ void __launch_bounds__(256) k_reorder_offsets(ReorderArgs a, const SlotMeta* sm,
                                                         const uint32_t* hfirst,
                                                         const uint32_t* hcell) {
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < a.total;
       g += (int64_t)gridDim.x * blockDim.x) {
    const int m = find_slot_of(sm, a.M, g);
    sm[m].out_off[g - sm[m].base] = (int32_t)hfirst[hcell[g]];
  }
}

// ------------------------------------------------------------- sharding_sparse_fids (8f-2)
// data/kernels/parse_sparse_feature.cc:173-211 (FillFidList) + parse_sparse_feature.h:1472-1680
// (FeatureParallelUniqueParse) on ragged FID arrays that are already on the device.
struct ShardArgs {
  int M, ps_num, n_out, unique;
  int64_t total;
  const int32_t* out_index;  // [M * ps_num] GetFidOutputIndex(feature m, shard)
};

__device__ __forceinline__ int shard_of_fid(int64_t id, int ps_num) {  // unsigned % (:178, :195)
  return (int)((uint64_t)id % (uint64_t)ps_num);
}

__global__ void __launch_bounds__(256) k_shard_keys(ShardArgs a, const SlotMeta* sm,
                                                    const uint32_t* hfirst, const uint32_t* hcell,
                                                    uint32_t* skey) {
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < a.total;
       g += (int64_t)gridDim.x * blockDim.x) {
    const int m = find_slot_of(sm, a.M, g);
    uint32_t key = (uint32_t)a.n_out;
    if (!a.unique || hfirst[hcell[g]] == (uint32_t)g) {
      const int64_t id = sm[m].ids[g - sm[m].base];
      key = (uint32_t)a.out_index[m * a.ps_num + shard_of_fid(id, a.ps_num)];
    }
    skey[g] = key;
  }
}

__global__ void __launch_bounds__(256) k_shard_hist(ShardArgs a, const uint32_t* skey,
                                                    int32_t* hist) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < a.n_out; c += gridDim.x * blockDim.x)
    hist[c] = (int32_t)(lower_bound_dev(skey, a.total, (int64_t)c + 1) -
                        lower_bound_dev(skey, a.total, (int64_t)c));
}

// sorted position r -> fid_list[r] and the index of the fid inside its (feature, shard) list
__global__ void __launch_bounds__(256) k_shard_emit(ShardArgs a, const SlotMeta* sm,
                                                    const uint32_t* skey, const uint32_t* sidx,
                                                    const int32_t* start, const uint32_t* hcell,
                                                    uint32_t* hfirst, int64_t* fid_list,
                                                    uint64_t* fid_offset) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.total;
       r += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t key = skey[r];
    if (key >= (uint32_t)a.n_out) continue;
    const uint32_t g = sidx[r];
    const int m = find_slot_of(sm, a.M, g);
    fid_list[r] = sm[m].ids[(int64_t)g - sm[m].base];
    const uint32_t idx = (uint32_t)(r - start[key]);
    if (a.unique) hfirst[hcell[g]] = idx;  // the dedup cell now carries the list index
    else fid_offset[g] = ((uint64_t)key << 32) | idx;
  }
}

__global__ void __launch_bounds__(256) k_shard_offsets(ShardArgs a, const SlotMeta* sm,
                                                       const uint32_t* hfirst,
                                                       const uint32_t* hcell,
                                                       uint64_t* fid_offset) {
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < a.total;
       g += (int64_t)gridDim.x * blockDim.x) {
    const int m = find_slot_of(sm, a.M, g);
    const int64_t id = sm[m].ids[g - sm[m].base];
    const uint32_t key = (uint32_t)a.out_index[m * a.ps_num + shard_of_fid(id, a.ps_num)];
    fid_offset[g] = ((uint64_t)key << 32) | hfirst[hcell[g]];
  }
}

struct RunMeta {  // per feature: its run boundaries -> feature_offset entries
  const int64_t* row_splits;  // [runs + 1], relative to the feature's first fid
  int64_t fid_base;           // first global fid position of the feature
  int64_t run_base;           // first feature_offset entry of the feature
  int64_t runs;
};

__global__ void __launch_bounds__(256) k_shard_feature_offsets(const RunMeta* rm, int M,
                                                               int64_t total_runs,
                                                               int32_t* feature_offset) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total_runs;
       j += (int64_t)gridDim.x * blockDim.x) {
    int lo = 0, hi = M - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (rm[mid].run_base <= j) lo = mid; else hi = mid - 1;
    }
    // features without runs share a run_base with their successor: take the one that owns j
    while (lo > 0 && rm[lo].runs == 0) --lo;
    const RunMeta r = rm[lo];
    feature_offset[j] = (int32_t)(r.fid_base + r.row_splits[j - r.run_base]);
  }
}

static SplitArgs single_split(float* out, int64_t dim) {
  SplitArgs sp{};
  sp.M = 1;
  sp.dims[0] = (int)dim;
  sp.outs[0] = out;
  return sp;
}

template <typename K, bool MEAN>
static int run_reduce(const K* row_ids, const float* values, int64_t n, int64_t dim, int64_t batch,
                      const SplitArgs& sp, const char* who, hipStream_t st) {
  MONO_REQUIRE(dim > 0 && batch >= 0 && n >= 0, "%s: bad shape", who);
  if (batch == 0) return 0;
  SortedRows<K> sr;
  MONO_TRY(make_sorted_rows<K>(row_ids, n, batch, &sr, who, st));
  hipLaunchKernelGGL((k_segment_reduce<K, MEAN>), dim3(grid_for(batch * dim, 256)), dim3(256), 0, st,
                     sr, values, n, dim, batch, sp);
  MONO_HIP(hipGetLastError());
  return 0;
}

}  // namespace mono

using namespace mono;

extern "C" {

int mono_reduce_sum(const int64_t* row_ids, const float* values, int64_t n, int64_t dim,
                    int64_t batch, float* out, void* stream) {
  return run_reduce<int64_t, false>(row_ids, values, n, dim, batch, single_split(out, dim),
                                    "mono_reduce_sum", (hipStream_t)stream);
}

// ---------------------------------------------------------------- mono_reduce_sum_noalloc
// mono_reduce_sum for callers that cannot allocate (code under HIP graph capture, tight serving
// loops, harnesses that forbid hipMalloc): the same kernels, with the scratch the existing entry
// point takes from a growable workspace (flag word, (row, position) pairs, radix-sort temporaries)
// in a file-scope __device__ array instead. Every GPU the library runs on holds its own copy of
// that array until the process ends; its address is resolved once per device.
constexpr int64_t kNoAllocMaxRows = 1 << 19;        // row ids per call
constexpr size_t kNoAllocScratchBytes = 16u << 20;  // 16 MiB per GPU
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];
constexpr int kNoAllocMaxDevices = 64;
static unsigned char* g_scratch_ptr[kNoAllocMaxDevices] = {};

// The array's name is only a host-side handle: kernels, memsets and copies get the device address
// this returns, never `scratch` itself.
static int noalloc_scratch(unsigned char** out) {
  int dev = -1;
  MONO_HIP(hipGetDevice(&dev));
  MONO_REQUIRE(dev >= 0 && dev < kNoAllocMaxDevices, "mono_reduce_sum_noalloc: device %d out of range", dev);
  if (!g_scratch_ptr[dev]) {  // once per device, with that device current
    void* p = nullptr;
    MONO_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    MONO_REQUIRE(p, "mono_reduce_sum_noalloc: no address for the scratch array on device %d", dev);
    g_scratch_ptr[dev] = (unsigned char*)p;
  }
  *out = g_scratch_ptr[dev];
  return 0;
}

int mono_reduce_sum_noalloc(const int64_t* row_ids, const float* values, int64_t n, int64_t dim,
                            int64_t batch, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  MONO_REQUIRE(dim > 0 && batch >= 0 && n >= 0, "mono_reduce_sum_noalloc: bad shape");
  MONO_REQUIRE(n <= kNoAllocMaxRows,
               "mono_reduce_sum_noalloc: %lld row ids exceed the %lld this entry point's fixed "
               "scratch serves; use mono_reduce_sum",
               (long long)n, (long long)kNoAllocMaxRows);
  MONO_REQUIRE(batch < (int64_t)0xFFFFFFF0ll, "mono_reduce_sum_noalloc: batch %lld too large",
               (long long)batch);
  if (batch == 0) return 0;
  MONO_REQUIRE(out && (n == 0 || (row_ids && values)), "mono_reduce_sum_noalloc: null buffer");
  unsigned char* base = nullptr;
  MONO_TRY(noalloc_scratch(&base));
  // layout: [0, 256) flag word | four uint32 arrays of n_al | sort temporaries
  const size_t n_al = ((size_t)n + 63) & ~(size_t)63;
  const int end_bit = bits_for_u((uint64_t)batch);
  const size_t tmp_bytes = sort_pairs_tmp_bytes((size_t)n, end_bit);
  const size_t off_tmp = (256 + 4 * n_al * sizeof(uint32_t) + 255) & ~(size_t)255;
  MONO_REQUIRE(off_tmp + tmp_bytes <= kNoAllocScratchBytes,
               "mono_reduce_sum_noalloc: %zu bytes of scratch needed, %zu available",
               off_tmp + tmp_bytes, kNoAllocScratchBytes);
  int* flags = (int*)base;
  uint32_t* kin = (uint32_t*)(base + 256);
  uint32_t* iin = kin + n_al;
  uint32_t* kout = iin + n_al;
  uint32_t* iout = kout + n_al;
  // The array keeps the previous call's contents. What must start at zero: the flag word (cleared
  // here); the radix sort's control words are cleared by its own first kernel; everything else is
  // written before it is read.
  int h_flags = 0;
  MONO_HIP(hipMemsetAsync(flags, 0, sizeof(int), st));
  if (n > 0) {
    hipLaunchKernelGGL(k_check_sorted<int64_t>, dim3(grid_for(n, 256, 2048)), dim3(256), 0, st,
                       row_ids, n, batch, flags);
    MONO_HIP(hipGetLastError());
  }
  MONO_HIP(hipMemcpyAsync(&h_flags, flags, sizeof(int), hipMemcpyDeviceToHost, st));
  MONO_HIP(hipStreamSynchronize(st));
  MONO_REQUIRE(!(h_flags & 2), "mono_reduce_sum_noalloc: a row id is outside [0, %lld)", (long long)batch);
  SortedRows<int64_t> sr;
  const bool sorted_here = (h_flags & 1) != 0;
  if (!sorted_here) {
    sr.rows = row_ids;
  } else {
    hipLaunchKernelGGL(k_keys_to_u32<int64_t>, dim3(grid_for(n, 256, 2048)), dim3(256), 0, st, row_ids,
                       n, kin, iin);
    MONO_HIP(hipGetLastError());
    MONO_TRY(sort_pairs_u32(kin, kout, iin, iout, (size_t)n, end_bit, base + off_tmp, tmp_bytes, st));
    sr.urows = kout;
    sr.perm = iout;
  }
  hipLaunchKernelGGL((k_segment_reduce<int64_t, false>), dim3(grid_for(batch * dim, 256)), dim3(256),
                     0, st, sr, values, n, dim, batch, single_split(out, dim));
  MONO_HIP(hipGetLastError());
  if (sorted_here) {
    // the fused sort passes give up (and say so in a sticky word) rather than wait forever for a
    // look-back word: sums over a half-sorted order must not be handed out as a result
    uint32_t h_err = 0;
    if (uint32_t* w = sort_error_word()) {
      MONO_HIP(hipMemcpyAsync(&h_err, w, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      MONO_HIP(hipStreamSynchronize(st));
      MONO_TRY(sort_error_report(h_err));
    }
  }
  return 0;
}

int mono_reduce_mean(const int64_t* row_ids, const float* values, int64_t n, int64_t dim,
                     int64_t batch, float* out, void* stream) {
  return run_reduce<int64_t, true>(row_ids, values, n, dim, batch, single_split(out, dim),
                                   "mono_reduce_mean", (hipStream_t)stream);
}

int mono_reduce_sum_gradient(const int64_t* row_ids, const float* grads, int64_t n, int64_t dim,
                             int64_t batch, float* out, void* stream) {
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int flags = 0;
  MONO_TRY(check_keys<int64_t>(row_ids, n, batch, &flags, st));
  MONO_REQUIRE(!(flags & 2), "mono_reduce_sum_gradient: a row id is outside [0, %lld)", (long long)batch);
  SortedRows<int64_t> sr;
  hipLaunchKernelGGL((k_segment_gather<int64_t, false>), dim3(grid_for(n * dim, 256)), dim3(256), 0,
                     st, row_ids, sr, n, dim, single_split(const_cast<float*>(grads), dim), out);
  MONO_HIP(hipGetLastError());
  return 0;
}

int mono_reduce_mean_gradient(const int64_t* row_ids, const float* grads, int64_t n, int64_t dim,
                              int64_t batch, float* out, void* stream) {
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  SortedRows<int64_t> sr;
  MONO_TRY(make_sorted_rows<int64_t>(row_ids, n, batch, &sr, "mono_reduce_mean_gradient", st));
  hipLaunchKernelGGL((k_segment_gather<int64_t, true>), dim3(grid_for(n * dim, 256)), dim3(256), 0,
                     st, row_ids, sr, n, dim, single_split(const_cast<float*>(grads), dim), out);
  MONO_HIP(hipGetLastError());
  return 0;
}

int mono_fused_reduce_sum_and_split(const int64_t* row_ids, const float* values, int64_t n,
                                    int64_t dim, int64_t batch, const int32_t* h_split_dims, int M,
                                    float* const* outs, void* stream) {
  MONO_REQUIRE(M >= 1 && M <= kMaxSplit, "mono_fused_reduce_sum_and_split: %d splits (max %d)", M,
               kMaxSplit);
  SplitArgs sp{};
  sp.M = M;
  int64_t sum = 0;
  for (int m = 0; m < M; ++m) {
    sp.dims[m] = h_split_dims[m];
    sp.outs[m] = outs[m];
    sum += h_split_dims[m];
  }
  MONO_REQUIRE(sum == dim, "mono_fused_reduce_sum_and_split: sum(split_dims) %lld != dim %lld",
               (long long)sum, (long long)dim);
  return run_reduce<int64_t, false>(row_ids, values, n, dim, batch, sp,
                                    "mono_fused_reduce_sum_and_split", (hipStream_t)stream);
}

int mono_fused_reduce_sum_and_split_gradient(const int64_t* row_ids, const float* const* grads,
                                             int64_t n, int64_t batch, const int32_t* h_split_dims,
                                             int M, float* out, void* stream) {
  MONO_REQUIRE(M >= 1 && M <= kMaxSplit, "mono_fused_reduce_sum_and_split_gradient: %d splits", M);
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  SplitArgs sp{};
  sp.M = M;
  int64_t dim = 0;
  for (int m = 0; m < M; ++m) {
    sp.dims[m] = h_split_dims[m];
    sp.outs[m] = const_cast<float*>(grads[m]);
    dim += h_split_dims[m];
  }
  int flags = 0;
  MONO_TRY(check_keys<int64_t>(row_ids, n, batch, &flags, st));
  MONO_REQUIRE(!(flags & 2), "mono_fused_reduce_sum_and_split_gradient: row id out of range");
  SortedRows<int64_t> sr;
  hipLaunchKernelGGL((k_segment_gather<int64_t, false>), dim3(grid_for(n * dim, 256)), dim3(256), 0,
                     st, row_ids, sr, n, dim, sp, out);
  MONO_HIP(hipGetLastError());
  return 0;
}

int mono_fused_sorted_segment_sum(const int32_t* const* indices, const float* const* values,
                                  const int64_t* h_lens, const int64_t* h_dims,
                                  const int64_t* h_out_rows, int N, float* const* outs,
                                  void* stream) {
  for (int t = 0; t < N; ++t)
    MONO_TRY((run_reduce<int32_t, false>(indices[t], values[t], h_lens[t], h_dims[t], h_out_rows[t],
                                         single_split(outs[t], h_dims[t]),
                                         "mono_fused_sorted_segment_sum", (hipStream_t)stream)));
  return 0;
}

int mono_first_n(const int64_t* row_ids, const float* values, int64_t n, int64_t dim, int64_t batch,
                 int64_t max_seq, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  MONO_REQUIRE(dim > 0 && max_seq > 0, "mono_first_n: bad shape");
  MONO_HIP(hipMemsetAsync(out, 0, (size_t)batch * max_seq * dim * sizeof(float), st));
  if (n == 0) return 0;
  int flags = 0;
  MONO_TRY(check_keys<int64_t>(row_ids, n, batch, &flags, st));
  MONO_REQUIRE(flags == 0, "mono_first_n: row ids must be sorted and inside [0, %lld)", (long long)batch);
  hipLaunchKernelGGL(k_first_n, dim3(grid_for(n * dim, 256)), dim3(256), 0, st, row_ids, values, n,
                     dim, max_seq, out);
  MONO_HIP(hipGetLastError());
  return 0;
}

int mono_fused_gather_embeddings_by_input(const float* fused, int64_t fused_len,
                                          const int32_t* const* offsets, const int64_t* h_lens,
                                          const int32_t* h_dims, int M, float* const* outs,
                                          void* stream) {
  (void)fused_len;
  hipStream_t st = (hipStream_t)stream;
  bool quads = ((uintptr_t)fused & 15) == 0;
  for (int m = 0; m < M; ++m)
    quads = quads && h_dims[m] % 4 == 0 && ((uintptr_t)outs[m] & 15) == 0 &&
            h_lens[m] * h_dims[m] < (int64_t)0x7FFFFFF0;
  for (int m = 0; m < M; ++m) {
    if (h_lens[m] == 0) continue;
    if (quads)
      hipLaunchKernelGGL(k_gather_rows4, dim3(grid_for(h_lens[m] * (h_dims[m] / 4), 256)), dim3(256),
                         0, st, (const float4*)fused, offsets[m], (uint32_t)h_lens[m],
                         (uint32_t)(h_dims[m] / 4), (float4*)outs[m]);
    else
      hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(h_lens[m] * h_dims[m], 256)), dim3(256), 0, st,
                         fused, offsets[m], h_lens[m], h_dims[m], outs[m]);
  }
  MONO_HIP(hipGetLastError());
  return 0;
}

static int gather_gradient(int64_t plan_token, int64_t fused_len, const float* const* grads,
                           const int32_t* const* offsets, const int64_t* h_lens,
                           const int32_t* h_dims, int M, float* out, hipStream_t st) {
  double hp = host_prof_on() ? host_now() : 0.0;
  bool aligned4 = ((uintptr_t)out & 15) == 0;
  for (int m = 0; m < M; ++m)
    aligned4 = aligned4 && (h_dims[m] % 4 == 0) && ((uintptr_t)grads[m] & 15) == 0;
  // the plan F1 prepared for exactly these offset arrays, if the caller names it
  GradPlan* plan = nullptr;
  if (plan_token != 0)
    for (GradPlan& p : g_plans)
      if (p.token == (uint64_t)plan_token && p.M == M && fused_len >= p.covered) {
        bool same = true;
        for (int m = 0; m < M; ++m)
          same = same && p.offs[m] == offsets[m] && p.lens[m] == h_lens[m] && p.dims[m] == h_dims[m];
        if (same) plan = &p;
      }
  if (plan) {
    // every float below `covered` is written by the sum itself (each unique id has an occurrence)
    if (fused_len > plan->covered)
      MONO_HIP(hipMemsetAsync(out + plan->covered, 0, (size_t)(fused_len - plan->covered) * sizeof(float),
                              st));
    char* base = (char*)plan->ws.ptr;
    host_tick("grad: find plan + tail fill", &hp);
    for (int m = 0; m < M; ++m) {
      if (h_lens[m] == 0) continue;
      MONO_TRY(segment_set_sum_planned((const uint32_t*)(base + plan->o_skey[m]),
                                       (const uint32_t*)(base + plan->o_sidx[m]), h_lens[m], grads[m],
                                       h_dims[m], aligned4, out, base + plan->o_q[m], plan->ready, st));
    }
    if (!plan->consumed) MONO_HIP(hipEventCreateWithFlags(&plan->consumed, hipEventDisableTiming));
    MONO_HIP(hipEventRecord(plan->consumed, st));
    plan->consumed_set = true;
    host_tick("grad: planned sums (all slots)", &hp);
    return 0;
  }
  MONO_HIP(hipMemsetAsync(out, 0, (size_t)fused_len * sizeof(float), st));
  for (int m = 0; m < M; ++m) {
    const int64_t n = h_lens[m];
    if (n == 0) continue;
    const size_t n_al = ((size_t)n + 63) & ~(size_t)63;
    MONO_TRY(g_ws.ensure(256 + 4 * n_al * sizeof(uint32_t)));
    uint32_t* kin = (uint32_t*)((char*)g_ws.ptr + 256);
    uint32_t* iin = kin + n_al;
    uint32_t* kout = iin + n_al;
    uint32_t* iout = kout + n_al;
    hipLaunchKernelGGL(k_keys_to_u32<int32_t>, dim3(grid_for(n, 256, 2048)), dim3(256), 0, st,
                       offsets[m], n, kin, iin);
    MONO_TRY(sort_pairs(kin, kout, iin, iout, (size_t)n,
                                      bits_for_u((uint64_t)fused_len), st));
    // duplicates of one offset are summed in input order by the table's segment machinery
    MONO_TRY(segment_add_rows(kout, iout, n, grads[m], h_dims[m], aligned4, out, st));
  }
  return 0;
}

int mono_fused_gather_embeddings_by_input_gradient(int64_t fused_len, const float* const* grads,
                                                   const int32_t* const* offsets,
                                                   const int64_t* h_lens, const int32_t* h_dims,
                                                   int M, float* out, void* stream) {
  return gather_gradient(0, fused_len, grads, offsets, h_lens, h_dims, M, out, (hipStream_t)stream);
}

int mono_fused_gather_embeddings_by_input_gradient_planned(int64_t plan, int64_t fused_len,
                                                           const float* const* grads,
                                                           const int32_t* const* offsets,
                                                           const int64_t* h_lens,
                                                           const int32_t* h_dims, int M, float* out,
                                                           void* stream) {
  return gather_gradient(plan, fused_len, grads, offsets, h_lens, h_dims, M, out,
                         (hipStream_t)stream);
}

int64_t mono_fused_reorder_last_plan(void) { return g_last_plan_token; }

// F1 in two halves. begin launches everything on `stream` and returns; finish waits for the
// sizes (an event right behind the histogram, ahead of the offsets and the gradient plan) and
// reports them. One reorder may be pending at a time. The reorder has workspaces of its own, so a
// caller may issue begin on a side stream one step ahead (the reference computes the reorder in
// its input pipeline, distributed_ps_sync.py early_reorder_indicies_res_pack) while other ops of
// this library run on the main stream.
struct ReorderPending {
  bool active = false;
  int N = 0, M = 0;
  std::vector<int32_t> dims;
  GradPlan* plan = nullptr;
  int32_t* pin = nullptr;
};
static ReorderPending g_reorder;
static Workspace g_ws_reorder, g_ws_reorder_sort;
// Behind the last kernel of the previous reorder (its offsets pass and the plan of its gradient run
// on after the sizes event that finish() waits for): a begin on ANOTHER stream waits for it before
// it overwrites the shared scratch (dedup cells, slot table, sort scratch).
static hipEvent_t g_ev_tail = nullptr;
static hipStream_t g_tail_stream = nullptr;
static bool g_tail_set = false;

static int reorder_begin(const int64_t* const* inputs, const int64_t* h_lens, int M, int N,
                         const int32_t* h_dims, int shard_rule, int64_t* out_ids,
                         int32_t* const* out_offsets, hipStream_t st) {
  MONO_REQUIRE(M >= 1 && N >= 1, "mono_fused_reorder_by_indices: M %d N %d", M, N);
  MONO_REQUIRE(shard_rule == 0 || (shard_rule == 1 && N >= 2),
               "mono_fused_reorder_by_indices: shard_rule %d with N %d", shard_rule, N);
  MONO_REQUIRE(!g_reorder.active, "mono_fused_reorder_begin: the previous reorder was never finished");
  g_reorder.N = N;
  g_reorder.M = M;
  g_reorder.dims.assign(h_dims, h_dims + M);
  g_reorder.plan = nullptr;
  g_reorder.pin = nullptr;
  g_last_plan_token = 0;
  std::vector<SlotMeta> sm(M);
  int64_t total = 0, cells = 0, heads_total = 0;
  for (int m = 0; m < M; ++m) {
    uint64_t sz = 8;
    while (sz < (uint64_t)h_lens[m] * 2) sz <<= 1;
    sm[m].ids = inputs[m];
    sm[m].out_off = out_offsets[m];
    sm[m].base = total;
    sm[m].len = h_lens[m];
    sm[m].hash_base = cells;
    sm[m].hash_mask = sz - 1;
    sm[m].dim = h_dims[m];
    sm[m].head_base = heads_total;
    heads_total += (h_lens[m] + 63) >> 6;
    total += h_lens[m];
    cells += (int64_t)sz + 1;
  }
  g_reorder.active = true;
  if (total == 0) return 0;  // (finish reports zeros)
  g_reorder.active = false;
  MONO_REQUIRE(total < (int64_t)0xFFFFFFF0ll && cells < (int64_t)0xFFFFFFF0ll,
               "mono_fused_reorder_by_indices: too many ids");
  const size_t t_al = ((size_t)total + 63) & ~(size_t)63;
  const size_t c_al = ((size_t)cells + 63) & ~(size_t)63;
  const size_t hist_al = ((size_t)(N * M + 1) + 63) & ~(size_t)63;
  size_t bytes = 256 + sizeof(SlotMeta) * M + 64;
  const size_t off_meta = 256;
  const size_t off_hkeys = (off_meta + sizeof(SlotMeta) * M + 63) & ~(size_t)63;
  const size_t off_hfirst = off_hkeys;  // 4-byte cells
  const size_t off_hcell = off_hkeys + c_al * 4;
  const size_t off_skey = off_hcell + t_al * 4;
  const size_t off_sidx = off_skey + t_al * 4;
  const size_t off_skey2 = off_sidx + t_al * 4;
  const size_t off_sidx2 = off_skey2 + t_al * 4;
  const size_t off_dims = off_sidx2 + t_al * 4;
  const size_t off_scan = off_dims + t_al * 4;
  const size_t off_hist = off_scan + t_al * 4;
  bytes = off_hist + hist_al * 4;
  if (g_tail_set && g_tail_stream != st) MONO_HIP(hipStreamWaitEvent(st, g_ev_tail, 0));
  MONO_TRY(g_ws_reorder.ensure(bytes));
  char* base = (char*)g_ws_reorder.ptr;
  SlotMeta* d_sm = (SlotMeta*)(base + off_meta);
  uint32_t* hfirst = (uint32_t*)(base + off_hfirst);
  uint32_t* hcell = (uint32_t*)(base + off_hcell);
  uint32_t* skey = (uint32_t*)(base + off_skey);
  uint32_t* sidx = (uint32_t*)(base + off_sidx);
  uint32_t* skey2 = (uint32_t*)(base + off_skey2);
  uint32_t* sidx2 = (uint32_t*)(base + off_sidx2);
  int32_t* dims = (int32_t*)(base + off_dims);
  int32_t* scan = (int32_t*)(base + off_scan);
  int32_t* hist = (int32_t*)(base + off_hist);
  MONO_TRY(upload_async(d_sm, sm.data(), sizeof(SlotMeta) * M, st));
  MONO_HIP(hipMemsetAsync(hfirst, 0xFF, c_al * 4, st));
  MONO_HIP(hipMemsetAsync(hist, 0, hist_al * 4, st));
  ReorderArgs a{M, N, shard_rule, total};
  const dim3 grid(grid_for(total, 256)), blk(256);
  double hp = host_prof_on() ? host_now() : 0.0;
  hipLaunchKernelGGL(k_reorder_insert, dim3(grid_for(heads_total, 256)), blk, 0, st, a, d_sm, hfirst, hcell,
                     0, heads_total);
  hipLaunchKernelGGL(k_reorder_insert, grid, blk, 0, st, a, d_sm, hfirst, hcell, 1, (int64_t)0);
  hipLaunchKernelGGL(k_reorder_keys, grid, blk, 0, st, a, d_sm, hfirst, hcell, skey, sidx);
  MONO_HIP(hipGetLastError());
  MONO_TRY(sort_pairs(skey, skey2, sidx, sidx2, (size_t)total, bits_for_u((uint64_t)N * M), st,
                      &g_ws_reorder_sort));
  bool uniform = true;  // one dim for every slot: offsets need no scan
  for (int m = 1; m < M; ++m) uniform = uniform && h_dims[m] == h_dims[0];
  uniform = uniform && total * (int64_t)h_dims[0] < (int64_t)0x7FFFFFF0;
  hipLaunchKernelGGL(k_reorder_hist, dim3(grid_for((int64_t)N * M, 256)), blk, 0, st, a, skey2, hist);
  MONO_HIP(hipGetLastError());
  // The sizes are final here: they go to the host now (pinned staging, an event of their own), and
  // everything below — the embedding offsets and the plan of the gradient — runs on the device
  // while the host wakes up and prepares the exchange.
  int32_t* pin = nullptr;
  MONO_TRY(pinned_words((size_t)N * M + 1, &pin));
  if (!g_ev_sizes) MONO_HIP(hipEventCreateWithFlags(&g_ev_sizes, hipEventDisableTiming));
  MONO_HIP(hipMemcpyAsync(pin, hist, (size_t)N * M * 4, hipMemcpyDeviceToHost, st));
  pin[(size_t)N * M] = 0;
  if (uint32_t* w = sort_error_word())
    MONO_HIP(hipMemcpyAsync(pin + (size_t)N * M, w, 4, hipMemcpyDeviceToHost, st));
  MONO_HIP(hipEventRecord(g_ev_sizes, st));
  host_tick("reorder: launches up to the sizes", &hp);
  if (!uniform) {
    hipLaunchKernelGGL(k_reorder_dims, grid, blk, 0, st, a, d_sm, skey2, dims);
    MONO_HIP(hipGetLastError());
    MONO_TRY(exclusive_scan_i32(dims, scan, (size_t)total, st, &g_ws_reorder_sort));
  }
  hipLaunchKernelGGL(k_reorder_emit, grid, blk, 0, st, a, d_sm, skey2, sidx2,
                     uniform ? (const int32_t*)nullptr : scan, uniform ? h_dims[0] : 0, hcell,
                     hfirst, out_ids);
  hipLaunchKernelGGL(k_reorder_offsets, grid, blk, 0, st, a, d_sm, hfirst, hcell);
  MONO_HIP(hipGetLastError());
  host_tick("reorder: emit + offsets launches", &hp);
  {
    std::vector<int32_t*> offs(out_offsets, out_offsets + M);
    MONO_TRY(grad_plan_build(offs, h_lens, h_dims, M, total, st, &g_reorder.plan));
  }
  host_tick("reorder: plan build launches", &hp);
  if (!g_ev_tail) MONO_HIP(hipEventCreateWithFlags(&g_ev_tail, hipEventDisableTiming));
  MONO_HIP(hipEventRecord(g_ev_tail, st));
  g_tail_stream = st;
  g_tail_set = true;
  g_reorder.pin = pin;
  g_reorder.active = true;
  return 0;
}

static int reorder_finish(int64_t* h_n_unique, int32_t* h_shard_sizes, int32_t* h_sharded_slot_sizes) {
  MONO_REQUIRE(g_reorder.active, "mono_fused_reorder_finish: no reorder is pending");
  g_reorder.active = false;
  const int N = g_reorder.N, M = g_reorder.M;
  for (int i = 0; i < N; ++i) h_shard_sizes[i] = 0;
  for (int i = 0; i < N * M; ++i) h_sharded_slot_sizes[i] = 0;
  *h_n_unique = 0;
  const int32_t* pin = g_reorder.pin;
  if (!pin) return 0;  // no ids at all
  double hp = host_prof_on() ? host_now() : 0.0;
  MONO_HIP(hipEventSynchronize(g_ev_sizes));
  host_tick("reorder: sync sizes (incl. GPU wait)", &hp);
  MONO_TRY(sort_error_report((uint32_t)pin[(size_t)N * M]));  // the sizes come out of a sort: never hand out stale ones
  int64_t uniq = 0, covered = 0;
  for (int n = 0; n < N; ++n)
    for (int m = 0; m < M; ++m) {
      const int32_t c = pin[n * M + m];
      h_sharded_slot_sizes[n * M + m] = c;
      h_shard_sizes[n] += c;
      uniq += c;
      covered += (int64_t)c * g_reorder.dims[m];
    }
  *h_n_unique = uniq;
  if (GradPlan* plan = g_reorder.plan) {
    plan->covered = covered;
    plan->token = ++g_plan_seq;
    g_last_plan_token = (int64_t)plan->token;
  }
  return 0;
}

int mono_fused_reorder_begin(const int64_t* const* inputs, const int64_t* h_lens, int M, int N,
                             const int32_t* h_dims, int shard_rule, int64_t* out_ids,
                             int32_t* const* out_offsets, void* stream) {
  return reorder_begin(inputs, h_lens, M, N, h_dims, shard_rule, out_ids, out_offsets,
                       (hipStream_t)stream);
}

int mono_fused_reorder_finish(int64_t* h_n_unique, int32_t* h_shard_sizes,
                              int32_t* h_sharded_slot_sizes) {
  return reorder_finish(h_n_unique, h_shard_sizes, h_sharded_slot_sizes);
}

int mono_fused_reorder_by_indices(const int64_t* const* inputs, const int64_t* h_lens, int M, int N,
                                  const int32_t* h_dims, int shard_rule, int64_t* out_ids,
                                  int64_t* h_n_unique, int32_t* h_shard_sizes,
                                  int32_t* h_sharded_slot_sizes, int32_t* const* out_offsets,
                                  void* stream) {
  MONO_TRY(reorder_begin(inputs, h_lens, M, N, h_dims, shard_rule, out_ids, out_offsets,
                         (hipStream_t)stream));
  return reorder_finish(h_n_unique, h_shard_sizes, h_sharded_slot_sizes);
}

int mono_sharding_sparse_fids(const int64_t* const* fids, const int64_t* const* row_splits,
                              const int64_t* h_lens, const int64_t* h_runs, int M, int ps_num,
                              const int32_t* h_out_index, int n_out, int unique,
                              int64_t* out_fid_list, uint64_t* out_fid_offset,
                              int32_t* out_feature_offset, int32_t* h_out_counts, void* stream) {
  MONO_REQUIRE(M >= 1 && ps_num >= 1 && n_out >= 1, "mono_sharding_sparse_fids: M %d ps_num %d n_out %d",
               M, ps_num, n_out);
  MONO_REQUIRE(h_lens && h_runs && h_out_index && h_out_counts, "mono_sharding_sparse_fids: null argument");
  for (int i = 0; i < M * ps_num; ++i)
    MONO_REQUIRE(h_out_index[i] >= 0 && h_out_index[i] < n_out,
                 "mono_sharding_sparse_fids: out_index[%d] = %d outside [0, %d)", i, h_out_index[i], n_out);
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < n_out; ++i) h_out_counts[i] = 0;
  std::vector<SlotMeta> sm(M);
  std::vector<RunMeta> rm(M);
  int64_t total = 0, cells = 0, total_runs = 0, heads_total = 0;
  for (int m = 0; m < M; ++m) {
    MONO_REQUIRE(h_lens[m] >= 0 && h_runs[m] >= 0, "mono_sharding_sparse_fids: negative length");
    MONO_REQUIRE(h_lens[m] == 0 || fids[m], "mono_sharding_sparse_fids: null fids[%d]", m);
    MONO_REQUIRE(h_runs[m] == 0 || row_splits[m], "mono_sharding_sparse_fids: null row_splits[%d]", m);
    MONO_REQUIRE(h_runs[m] > 0 || h_lens[m] == 0, "mono_sharding_sparse_fids: fids without runs");
    uint64_t sz = 8;
    while (sz < (uint64_t)h_lens[m] * 2) sz <<= 1;
    sm[m] = SlotMeta{fids[m], nullptr, total, h_lens[m], cells, sz - 1, 1, heads_total};
    heads_total += (h_lens[m] + 63) >> 6;
    rm[m] = RunMeta{row_splits[m], total, total_runs, h_runs[m]};
    total += h_lens[m];
    total_runs += h_runs[m];
    if (unique) cells += (int64_t)sz + 1;
  }
  MONO_REQUIRE(total < (int64_t)0x7FFFFFF0ll && cells < (int64_t)0xFFFFFFF0ll &&
               total_runs < (int64_t)0x7FFFFFF0ll, "mono_sharding_sparse_fids: too many ids");
  MONO_REQUIRE(total_runs == 0 || out_feature_offset, "mono_sharding_sparse_fids: null feature_offset");
  MONO_REQUIRE(total == 0 || (out_fid_list && out_fid_offset), "mono_sharding_sparse_fids: null output");
  const size_t t_al = ((size_t)total + 63) & ~(size_t)63;
  const size_t c_al = ((size_t)cells + 63) & ~(size_t)63;
  const size_t o_al = ((size_t)(n_out + 1) + 63) & ~(size_t)63;
  const size_t off_sm = 256;
  const size_t off_rm = (off_sm + sizeof(SlotMeta) * M + 63) & ~(size_t)63;
  const size_t off_oi = (off_rm + sizeof(RunMeta) * M + 63) & ~(size_t)63;
  const size_t off_hkeys = (off_oi + (size_t)M * ps_num * 4 + 63) & ~(size_t)63;
  const size_t off_hfirst = off_hkeys;  // 4-byte cells
  const size_t off_hcell = off_hkeys + c_al * 4;
  const size_t off_skey = off_hcell + (unique ? t_al * 4 : 0);
  const size_t off_skey2 = off_skey + t_al * 4;
  const size_t off_sidx2 = off_skey2 + t_al * 4;
  const size_t off_hist = off_sidx2 + t_al * 4;
  const size_t off_start = off_hist + o_al * 4;
  MONO_TRY(g_ws.ensure(off_start + o_al * 4));
  char* base = (char*)g_ws.ptr;
  SlotMeta* d_sm = (SlotMeta*)(base + off_sm);
  RunMeta* d_rm = (RunMeta*)(base + off_rm);
  int32_t* d_oi = (int32_t*)(base + off_oi);
  uint32_t* hfirst = (uint32_t*)(base + off_hfirst);
  uint32_t* hcell = (uint32_t*)(base + off_hcell);
  uint32_t* skey = (uint32_t*)(base + off_skey);
  uint32_t* skey2 = (uint32_t*)(base + off_skey2);
  uint32_t* sidx2 = (uint32_t*)(base + off_sidx2);
  int32_t* hist = (int32_t*)(base + off_hist);
  int32_t* start = (int32_t*)(base + off_start);
  MONO_TRY(upload_async(d_sm, sm.data(), sizeof(SlotMeta) * M, st));
  MONO_TRY(upload_async(d_rm, rm.data(), sizeof(RunMeta) * M, st));
  MONO_TRY(upload_async(d_oi, h_out_index, (size_t)M * ps_num * 4, st));
  const dim3 blk(256);
  if (total_runs > 0)
    hipLaunchKernelGGL(k_shard_feature_offsets, dim3(grid_for(total_runs, 256)), blk, 0, st, d_rm, M,
                       total_runs, out_feature_offset);
  if (total == 0) {
    MONO_HIP(hipGetLastError());
    MONO_HIP(hipStreamSynchronize(st));  // sm / rm / h_out_index are read by the copies above
    return 0;
  }
  const dim3 grid(grid_for(total, 256));
  ShardArgs a{M, ps_num, n_out, unique ? 1 : 0, total, d_oi};
  if (unique) {
    MONO_HIP(hipMemsetAsync(hfirst, 0xFF, c_al * 4, st));
    ReorderArgs ra{M, ps_num, 0, total};
    hipLaunchKernelGGL(k_reorder_insert, dim3(grid_for(heads_total, 256)), blk, 0, st, ra, d_sm, hfirst,
                       hcell, 0, heads_total);
    hipLaunchKernelGGL(k_reorder_insert, grid, blk, 0, st, ra, d_sm, hfirst, hcell, 1, (int64_t)0);
  }
  hipLaunchKernelGGL(k_shard_keys, grid, blk, 0, st, a, d_sm, hfirst, hcell, skey);
  MONO_HIP(hipGetLastError());
  MONO_TRY(sort_pairs(skey, skey2, nullptr, sidx2, (size_t)total, bits_for_u((uint64_t)n_out), st));
  hipLaunchKernelGGL(k_shard_hist, dim3(grid_for(n_out, 256)), blk, 0, st, a, skey2, hist);
  MONO_HIP(hipGetLastError());
  MONO_TRY(exclusive_scan_i32(hist, start, (size_t)n_out, st));
  hipLaunchKernelGGL(k_shard_emit, grid, blk, 0, st, a, d_sm, skey2, sidx2, start, hcell, hfirst,
                     out_fid_list, out_fid_offset);
  if (unique)
    hipLaunchKernelGGL(k_shard_offsets, grid, blk, 0, st, a, d_sm, hfirst, hcell, out_fid_offset);
  MONO_HIP(hipGetLastError());
  MONO_HIP(hipMemcpyAsync(h_out_counts, hist, (size_t)n_out * 4, hipMemcpyDeviceToHost, st));
  uint32_t h_sort_err = 0;
  if (uint32_t* w = sort_error_word())
    MONO_HIP(hipMemcpyAsync(&h_sort_err, w, sizeof(h_sort_err), hipMemcpyDeviceToHost, st));
  MONO_HIP(hipStreamSynchronize(st));
  return sort_error_report(h_sort_err);
}

}  // extern "C"
