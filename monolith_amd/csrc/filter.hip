// filter.hip — host side + batch kernels of the HBM-resident sliding hash filter.
// Reference: runtime/neo_filter/{hash_filter.h,sliding_hash_filter.{h,cc},dummy_hash_filter.h},
// ops/neo_filter_tf_bridge.cc:25-40, ops/hash_filter_op.cc:109-141,
// ops/hash_filter_intercept_gradient_op.cc:39-62.
#include <algorithm>
#include <cstring>
#include <vector>

#include "filter_dev.h"
#include "mono_common.h"

namespace mono {

struct Filter {
  int kind = 0;
  FilterDev d{};
  int32_t* thresholds_dev = nullptr;
  Workspace ws;      // sort scratch
  Workspace ws_old;  // per-occurrence old counts of intercept_gradient
};

int filter_dev_view(void* filter, FilterDev* out, hipStream_t) {
  Filter* f = (Filter*)filter;
  *out = f->d;
  out->kind = f->kind;
  return 0;
}

// sliding_hash_filter.cc:83-87, applied at the batch boundary (one rotation per launch pair;
// filter_after_batch repeats the pair as often as the batch could have filled blocks).
__global__ void k_filter_rotate_decide(FilterDev f) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t head = f.state[0];
  uint64_t ne = 0;
  for (int sh = 0; sh < kFilterShards; ++sh) ne += *filter_counter(f, head, (uint32_t)sh);
  if (ne > f.capacity) ne = f.capacity;
  if (ne + 1 >= f.capacity) {  // HashFilter::full(): num_elements >= capacity - 1
    head = head + 1 == f.n_blocks ? 0 : head + 1;
    f.state[0] = head;
    f.state[1] += 1;
    f.state[3] = 1 + (head + 1) % f.n_blocks;  // block (head + max_forward - 1) is cleared
  } else {
    f.state[3] = 0;
  }
}

__global__ void __launch_bounds__(256) k_filter_rotate_clear(FilterDev f) {
  uint32_t flag = f.state[3];
  if (!flag) return;
  const uint32_t blk = flag - 1;
  uint16_t* base = f.cells + (uint64_t)blk * f.stride;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < f.stride;
       i += (uint64_t)gridDim.x * blockDim.x)
    base[i] = 0;
  if (blockIdx.x == 0 && threadIdx.x < kFilterShards) *filter_counter(f, blk, threadIdx.x) = 0;
}

// `new_ids`: an upper bound of the ids the batch may have added. A batch can fill the head block
// and spill into the next one (filter_add looks forward over two blocks); each decide/clear pair
// rotates once if the (new) head is full, so the pair runs once per block the batch could have
// filled, and the next batch never starts on a full head. (A single batch that overflows BOTH
// blocks it can reach still sees failed adds inside the batch — the reference rotates in the
// middle of its stream — which the gate admits and failure_count records: DESIGN.md deviation 2.)
int filter_after_batch(void* filter, hipStream_t st, int64_t new_ids) {
  Filter* f = (Filter*)filter;
  if (f->kind != 1) return 0;
  int64_t rounds = 1;
  if (new_ids > 0 && f->d.capacity > 0) rounds = 1 + new_ids / (int64_t)f->d.capacity;
  rounds = std::min<int64_t>(rounds, std::max<int64_t>(1, (int64_t)f->d.n_blocks - 1));
  for (int64_t r = 0; r < rounds; ++r) {
    hipLaunchKernelGGL(k_filter_rotate_decide, dim3(1), dim3(64), 0, st, f->d);
    hipLaunchKernelGGL(k_filter_rotate_clear, dim3(grid_for((int64_t)f->d.stride, 256, 2048)),
                       dim3(256), 0, st, f->d);
  }
  MONO_HIP(hipGetLastError());
  return 0;
}

__global__ void k_iota32(uint32_t* idx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    idx[i] = (uint32_t)i;
}

// One thread per sorted position; segment heads walk their occurrences in input order.
// old[i] = count the i-th occurrence's add would have returned (Filter::add semantics).
__global__ void __launch_bounds__(256) k_filter_batch_add(FilterDev f, const int64_t* skeys,
                                                          const uint32_t* sidx, int64_t n,
                                                          const int32_t* counts,
                                                          int respect_threshold, int32_t* old) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  int64_t key = skeys[j];
  if (j > 0 && skeys[j - 1] == key) return;
  bool gate_off = respect_threshold && filter_threshold(f, key) <= 0;
  uint32_t c = gate_off ? kFilterMaxCount : filter_get(f, (uint64_t)key);
  uint32_t total = 0;
  for (int64_t p = j; p < n && skeys[p] == key; ++p) {
    uint32_t i = sidx[p];
    uint32_t seen = c + total;
    old[i] = (int32_t)(seen > kFilterMaxCount ? kFilterMaxCount : seen);
    uint32_t add = counts ? (uint32_t)counts[i] : 1u;
    if (add > kFilterMaxCount) add = kFilterMaxCount;
    total += add;
    if (total > kFilterMaxCount) total = kFilterMaxCount;
  }
  if (!gate_off && total > 0) filter_add(f, (uint64_t)key, total);
}

__global__ void __launch_bounds__(256) k_filter_batch_get(FilterDev f, const int64_t* ids,
                                                          int64_t n, int32_t* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (int32_t)filter_get(f, (uint64_t)ids[i]);
}

__global__ void __launch_bounds__(256) k_intercept(FilterDev f, const int64_t* ids,
                                                   const int32_t* old, const float* grad,
                                                   int64_t n, int64_t width, float* out) {
  const int64_t total = n * width;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = e / width;
    int thr = filter_threshold(f, ids[i]);
    bool filtered = thr > 0 && old[i] < thr;
    out[e] = filtered ? 0.f : grad[e];
  }
}

__global__ void __launch_bounds__(256) k_fill_i32(int32_t* p, int64_t n, int32_t v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    p[i] = v;
}

// Group ids (equal ids adjacent, input order inside a group: sort.hip group_keys64) and run the
// batch add. old_out: device int32[n].
static int batch_add(Filter* f, const int64_t* ids, const int32_t* counts, int64_t n,
                     int respect_threshold, int32_t* old_out, hipStream_t st) {
  const size_t tmp_bytes = (group_keys64_tmp_bytes((size_t)n) + 255) & ~(size_t)255;
  const size_t n_al = ((size_t)n + 63) & ~(size_t)63;
  MONO_TRY(f->ws.ensure(n_al * 8 + n_al * 4 + tmp_bytes));
  int64_t* key_out = (int64_t*)f->ws.ptr;
  uint32_t* idx_out = (uint32_t*)(key_out + n_al);
  void* tmp = idx_out + n_al;
  MONO_TRY(group_keys64(ids, nullptr, key_out, idx_out, (size_t)n, tmp, tmp_bytes, st));
  hipLaunchKernelGGL(k_filter_batch_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, f->d,
                     key_out, idx_out, n, counts, respect_threshold, old_out);
  MONO_HIP(hipGetLastError());
  return filter_after_batch(f, st, n);
}

}  // namespace mono

using namespace mono;

extern "C" {

int mono_filter_create(int kind, int64_t capacity, int split_num, int default_threshold,
                       int n_slot_thresholds, const int32_t* h_slots, const int32_t* h_thresholds,
                       void** out_filter) {
  MONO_REQUIRE(out_filter, "mono_filter_create: null out_filter");
  MONO_REQUIRE(kind == MONO_FILTER_DUMMY || kind == MONO_FILTER_SLIDING,
               "mono_filter_create: unknown kind %d", kind);
  Filter* f = new Filter();
  f->kind = kind;
  std::vector<int32_t> thr(1 << 15, default_threshold);  // neo_filter_tf_bridge.cc:28-35
  for (int i = 0; i < n_slot_thresholds; ++i)
    if (h_slots[i] >= 0 && h_slots[i] < (1 << 15)) thr[h_slots[i]] = h_thresholds[i];
  if (hipMalloc((void**)&f->thresholds_dev, thr.size() * 4) != hipSuccess ||
      hipMemcpy(f->thresholds_dev, thr.data(), thr.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
    delete f;
    return fail("mono_filter_create: cannot allocate thresholds");
  }
  f->d.thresholds = f->thresholds_dev;
  if (kind == MONO_FILTER_SLIDING) {  // sliding_hash_filter.cc:28-40
    uint64_t cap = capacity < 300 ? 300 : (uint64_t)capacity;
    if (split_num < 5) split_num = 5;
    FilterDev& d = f->d;
    d.n_blocks = (uint32_t)split_num;
    d.max_backward = (uint32_t)(split_num - 2);
    d.capacity = cap / (uint64_t)(split_num - 1);  // sliding_hash_filter.h:33-35
    d.total_size = (uint64_t)((double)d.capacity * 1.2);  // hash_filter.h:90
    if (d.total_size == 0) {
      (void)hipFree(f->thresholds_dev);
      delete f;
      return fail("mono_filter_create: capacity %lld too small for %d splits", (long long)capacity,
                  split_num);
    }
    d.stride = d.total_size + 64;  // hash_filter.h:94
    size_t cells = (size_t)d.n_blocks * d.stride;
    cells = (cells + 1) & ~(size_t)1;
    if (hipMalloc((void**)&d.cells, cells * 2) != hipSuccess ||
        hipMalloc((void**)&d.num_elements, filter_counter_words(d.n_blocks) * 4) != hipSuccess ||
        hipMalloc((void**)&d.state, 4 * 4) != hipSuccess ||
        hipMemset(d.cells, 0, cells * 2) != hipSuccess ||
        hipMemset(d.num_elements, 0, filter_counter_words(d.n_blocks) * 4) != hipSuccess ||
        hipMemset(d.state, 0, 16) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      mono_filter_destroy(f);
      return fail("mono_filter_create: cannot allocate %zu filter cells", cells);
    }
  }
  *out_filter = f;
  return 0;
}

int mono_filter_destroy(void* filter) {
  if (!filter) return 0;
  Filter* f = (Filter*)filter;
  if (f->d.cells) (void)hipFree(f->d.cells);
  if (f->d.num_elements) (void)hipFree(f->d.num_elements);
  if (f->d.state) (void)hipFree(f->d.state);
  if (f->thresholds_dev) (void)hipFree(f->thresholds_dev);
  f->ws.release();
  f->ws_old.release();
  delete f;
  return 0;
}

int mono_filter_add(void* filter, const int64_t* ids, const int32_t* counts, int64_t n,
                    int32_t* old_counts, void* stream) {
  MONO_REQUIRE(filter, "mono_filter_add: null filter");
  if (n == 0) return 0;
  Filter* f = (Filter*)filter;
  hipStream_t st = (hipStream_t)stream;
  if (f->kind != 1) {
    hipLaunchKernelGGL(k_fill_i32, dim3(grid_for(n, 256)), dim3(256), 0, st, old_counts, n, 15);
    MONO_HIP(hipGetLastError());
    return 0;
  }
  return batch_add(f, ids, counts, n, 0, old_counts, st);
}

int mono_filter_get(void* filter, const int64_t* ids, int64_t n, int32_t* counts, void* stream) {
  MONO_REQUIRE(filter, "mono_filter_get: null filter");
  if (n == 0) return 0;
  Filter* f = (Filter*)filter;
  hipStream_t st = (hipStream_t)stream;
  if (f->kind != 1) {
    hipLaunchKernelGGL(k_fill_i32, dim3(grid_for(n, 256)), dim3(256), 0, st, counts, n, 15);
  } else {
    hipLaunchKernelGGL(k_filter_batch_get, dim3(grid_for(n, 256)), dim3(256), 0, st, f->d, ids, n,
                       counts);
  }
  MONO_HIP(hipGetLastError());
  return 0;
}

int mono_filter_intercept_gradient(void* filter, const int64_t* ids, const float* grad, int64_t n,
                                   int64_t width, float* out, void* stream) {
  MONO_REQUIRE(filter, "mono_filter_intercept_gradient: null filter");
  if (n == 0 || width == 0) return 0;
  Filter* f = (Filter*)filter;
  hipStream_t st = (hipStream_t)stream;
  if (f->kind != 1) {
    MONO_HIP(hipMemcpyAsync(out, grad, (size_t)n * width * 4, hipMemcpyDeviceToDevice, st));
    return 0;
  }
  MONO_TRY(f->ws_old.ensure((size_t)n * 4));
  int32_t* old = (int32_t*)f->ws_old.ptr;
  MONO_TRY(batch_add(f, ids, nullptr, n, 1, old, st));
  hipLaunchKernelGGL(k_intercept, dim3(grid_for(n * width, 256)), dim3(256), 0, st, f->d, ids, old,
                     grad, n, width, out);
  MONO_HIP(hipGetLastError());
  return 0;
}

int mono_filter_state(void* filter, int64_t* h_out, int64_t h_out_len, void* stream) {
  MONO_REQUIRE(filter && h_out, "mono_filter_state: null argument");
  Filter* f = (Filter*)filter;
  for (int64_t i = 0; i < h_out_len; ++i) h_out[i] = 0;
  if (f->kind != 1) return 0;
  hipStream_t st = (hipStream_t)stream;
  uint32_t state[4];
  std::vector<uint32_t> raw(filter_counter_words(f->d.n_blocks));
  MONO_HIP(hipMemcpyAsync(state, f->d.state, 16, hipMemcpyDeviceToHost, st));
  MONO_HIP(hipMemcpyAsync(raw.data(), f->d.num_elements, raw.size() * 4, hipMemcpyDeviceToHost, st));
  MONO_HIP(hipStreamSynchronize(st));
  std::vector<uint64_t> ne(f->d.n_blocks, 0);
  for (uint32_t b = 0; b < f->d.n_blocks; ++b)
    for (int sh = 0; sh < kFilterShards; ++sh)
      ne[b] += raw[((size_t)b * kFilterShards + sh) * kFilterShardStride];
  int64_t vals[5] = {state[0], state[1], state[2], f->d.n_blocks, (int64_t)f->d.total_size};
  for (int i = 0; i < 5 && i < h_out_len; ++i) h_out[i] = vals[i];
  for (uint32_t b = 0; b < f->d.n_blocks && 5 + b < h_out_len; ++b)
    h_out[5 + b] = (int64_t)std::min<uint64_t>(ne[b], f->d.capacity);
  return 0;
}

// One block ("split") of a sliding filter as the reference dumps it: the block's whole cell
// vector map_ (total_size + 64 cells, hash_filter.cc:44-56) to / from HOST memory.
int mono_filter_export_split(void* filter, int split, uint16_t* h_cells, int64_t n_cells,
                             int64_t* h_num_elements, void* stream) {
  MONO_REQUIRE(filter && h_cells, "mono_filter_export_split: null argument");
  Filter* f = (Filter*)filter;
  MONO_REQUIRE(f->kind == 1, "mono_filter_export_split: not a sliding filter");
  MONO_REQUIRE(split >= 0 && (uint32_t)split < f->d.n_blocks,
               "mono_filter_export_split: split %d of %u", split, f->d.n_blocks);
  MONO_REQUIRE((uint64_t)n_cells == f->d.stride, "mono_filter_export_split: %lld cells, block has %llu",
               (long long)n_cells, (unsigned long long)f->d.stride);
  hipStream_t st = (hipStream_t)stream;
  uint64_t ne = 0;
  std::vector<uint32_t> raw((size_t)kFilterShards * kFilterShardStride);
  MONO_HIP(hipMemcpyAsync(h_cells, f->d.cells + (size_t)split * f->d.stride, (size_t)n_cells * 2,
                          hipMemcpyDeviceToHost, st));
  MONO_HIP(hipMemcpyAsync(raw.data(), f->d.num_elements + (size_t)split * raw.size(), raw.size() * 4,
                          hipMemcpyDeviceToHost, st));
  MONO_HIP(hipStreamSynchronize(st));
  for (int sh = 0; sh < kFilterShards; ++sh) ne += raw[(size_t)sh * kFilterShardStride];
  if (h_num_elements) *h_num_elements = (int64_t)std::min<uint64_t>(ne, f->d.capacity);
  return 0;
}

// HashFilter::Restore (hash_filter.cc:60-78) for one block.
int mono_filter_import_split(void* filter, int split, const uint16_t* h_cells, int64_t n_cells,
                             int64_t num_elements, void* stream) {
  MONO_REQUIRE(filter && h_cells, "mono_filter_import_split: null argument");
  Filter* f = (Filter*)filter;
  MONO_REQUIRE(f->kind == 1, "mono_filter_import_split: not a sliding filter");
  MONO_REQUIRE(split >= 0 && (uint32_t)split < f->d.n_blocks,
               "mono_filter_import_split: split %d of %u", split, f->d.n_blocks);
  MONO_REQUIRE((uint64_t)n_cells == f->d.stride, "mono_filter_import_split: %lld cells, block has %llu",
               (long long)n_cells, (unsigned long long)f->d.stride);
  MONO_REQUIRE(num_elements >= 0, "mono_filter_import_split: negative num_elements");
  hipStream_t st = (hipStream_t)stream;
  const uint32_t ne = (uint32_t)std::min<uint64_t>((uint64_t)num_elements, f->d.capacity);
  MONO_HIP(hipMemcpyAsync(f->d.cells + (size_t)split * f->d.stride, h_cells, (size_t)n_cells * 2,
                          hipMemcpyHostToDevice, st));
  // (the restored count goes to shard 0, the other shards of the block restart at zero)
  std::vector<uint32_t> raw((size_t)kFilterShards * kFilterShardStride, 0u);
  raw[0] = ne;
  MONO_HIP(hipMemcpyAsync(f->d.num_elements + (size_t)split * raw.size(), raw.data(), raw.size() * 4,
                          hipMemcpyHostToDevice, st));
  MONO_HIP(hipStreamSynchronize(st));
  return 0;
}

// SlidingHashFilter::RestoreMetaDump (sliding_hash_filter.cc:187-201): head, head_increment,
// failure_count.
int mono_filter_set_state(void* filter, int64_t head, int64_t head_increment, int64_t failure_count,
                          void* stream) {
  MONO_REQUIRE(filter, "mono_filter_set_state: null filter");
  Filter* f = (Filter*)filter;
  MONO_REQUIRE(f->kind == 1, "mono_filter_set_state: not a sliding filter");
  MONO_REQUIRE(head >= 0 && (uint64_t)head < f->d.n_blocks, "mono_filter_set_state: head %lld of %u",
               (long long)head, f->d.n_blocks);
  hipStream_t st = (hipStream_t)stream;
  uint32_t state[4] = {(uint32_t)head, (uint32_t)head_increment, (uint32_t)failure_count, 0};
  MONO_HIP(hipMemcpyAsync(f->d.state, state, 16, hipMemcpyHostToDevice, st));
  MONO_HIP(hipStreamSynchronize(st));
  return 0;
}


}  // extern "C"
