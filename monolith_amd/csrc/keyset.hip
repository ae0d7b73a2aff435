// keyset.hip — the touched-key set on the device.
//
//   runtime/hopscotch/hopscotch_hash_set.{h,cc}      insert-only concurrent set of FIDs
//   ops/touched_key_set_{op,insert_op,steal_op}.cc   MonolithTouchedKeySet / Insert / Steal
//   ops/embedding_hash_table_tf_bridge.cc:303-307    every id an optimizer step was applied to
//
// Reference semantics, per key, in order (hopscotch_hash_set.cc:104-166): if the set holds MORE
// than `capacity` keys when an insert starts, all of them are dropped first (the call reports how
// many); then the key is added unless present. GetAndClear returns the keys in unspecified order.
//
// Batch form: a batch's keys are inserted in input order. With s0 keys before the batch and
// excl(g) = number of first occurrences of not-yet-present keys before position g, the set is
// dropped at the first g with s0 + excl(g) > capacity; the rest of the batch then starts from an
// empty set. One batch needs one round unless it overflows (rare: capacity defaults to 2 M keys).
//
// Storage: open addressing over `slots` = 2^k >= 2 * (capacity + 2) cells, 64-bit CAS on the key,
// EMPTY = INT64_MIN (a FID never takes that value in practice; it is still accepted, through a
// flag). mark[cell]: NONE = empty / not yet recorded, PRE = present before this batch, otherwise
// the smallest position of the key in the running batch.
#include <algorithm>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <vector>

#include "mono_common.h"

namespace mono {

constexpr int64_t kKsEmpty = INT64_MIN;
constexpr uint32_t kKsNone = 0xFFFFFFFFu;
constexpr uint32_t kKsPre = 0xFFFFFFFEu;

struct KeySet {
  int64_t capacity = 0;
  uint64_t slots = 0;
  int64_t* keys = nullptr;
  uint32_t* mark = nullptr;
  uint32_t* ctl = nullptr;  // [0] min-flag (holds EMPTY key), [1] overflow position, [2] steal cursor,
                            // [3] error: an insert found no cell (cannot happen, see round_max)
  int64_t size = 0;         // host copy: keys in the set (incl. the flagged EMPTY key)
  bool has_empty_key = false;
  Workspace ws;
};

__global__ void __launch_bounds__(256) k_ks_insert(int64_t* keys, uint32_t* mark, uint64_t mask,
                                                   const int64_t* __restrict__ ids, int64_t lo,
                                                   int64_t n, uint32_t* __restrict__ cell_of,
                                                   uint32_t* ctl) {
  for (int64_t g = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n;
       g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t key = ids[g];
    if (key == kKsEmpty) {  // cannot live in a cell: flag word, first position by atomicMin
      cell_of[g] = kKsNone;
      atomicMin(&ctl[0], (uint32_t)g);
      continue;
    }
    uint64_t p = mix64((uint64_t)key) & mask;
    // The host hands a round at most (cells - keys held - 1) positions, so an empty cell always
    // exists; the probe is bounded all the same: a full table raises ctl[3] instead of spinning.
    bool placed = false;
    for (uint64_t step = 0; step <= mask; ++step) {
      int64_t cur = keys[p];
      if (cur == kKsEmpty) {
        cur = (int64_t)atomicCAS(reinterpret_cast<unsigned long long*>(keys + p),
                                 (unsigned long long)kKsEmpty, (unsigned long long)key);
        if (cur == kKsEmpty) cur = key;
      }
      if (cur == key) { placed = true; break; }
      p = (p + 1) & mask;
    }
    if (!placed) {
      cell_of[g] = kKsNone;
      atomicExch(&ctl[3], 1u);
      continue;
    }
    cell_of[g] = (uint32_t)p;
    // keys present before this batch keep PRE; new ones collect their first position
    if (__hip_atomic_load(&mark[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kKsPre)
      atomicMin(&mark[p], (uint32_t)g);
  }
}

// flag[g] = 1 iff position g is the first occurrence of a key that was not in the set
__global__ void __launch_bounds__(256) k_ks_flags(const uint32_t* __restrict__ mark,
                                                  const uint32_t* __restrict__ cell_of,
                                                  const uint32_t* __restrict__ ctl, int64_t lo,
                                                  int64_t n, int empty_key_present,
                                                  int32_t* __restrict__ flag) {
  for (int64_t g = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n;
       g += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t c = cell_of[g];
    int f;
    if (c == kKsNone) f = !empty_key_present && ctl[0] == (uint32_t)g;
    else f = mark[c] == (uint32_t)g;
    flag[g - lo] = f;
  }
}

// the first position whose preceding new-key count pushes the set over its capacity
__global__ void __launch_bounds__(256) k_ks_overflow(const int32_t* __restrict__ excl, int64_t lo,
                                                     int64_t n, int64_t s0, int64_t capacity,
                                                     uint32_t* ctl) {
  for (int64_t g = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n;
       g += (int64_t)gridDim.x * blockDim.x) {
    const bool over = s0 + excl[g - lo] > capacity;
    const bool prev_over = g > lo && s0 + excl[g - lo - 1] > capacity;
    if (over && !prev_over) ctl[1] = (uint32_t)g;  // monotone: exactly one writer
  }
}

__global__ void __launch_bounds__(256) k_ks_commit(uint32_t* mark, const uint32_t* __restrict__ cell_of,
                                                   int64_t lo, int64_t n) {
  for (int64_t g = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n;
       g += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t c = cell_of[g];
    if (c != kKsNone && mark[c] == (uint32_t)g) mark[c] = kKsPre;
  }
}

__global__ void __launch_bounds__(256) k_ks_steal(const int64_t* __restrict__ keys, uint64_t slots,
                                                  int64_t* __restrict__ out, int64_t out_cap,
                                                  uint32_t* ctl) {
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < slots;
       p += (uint64_t)gridDim.x * blockDim.x) {
    const int64_t k = keys[p];
    const bool live = k != kKsEmpty;
    // one cursor bump per wave
    const unsigned long long m = __ballot(live);
    if (m == 0) continue;
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    const int leader = __ffsll((long long)m) - 1;
    if (lane == leader) base = atomicAdd(&ctl[2], (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    if (live) {
      const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if ((int64_t)pos < out_cap) out[pos] = k;
    }
  }
}

__global__ void __launch_bounds__(256) k_ks_fill(int64_t* keys, uint32_t* mark, uint64_t slots) {
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < slots;
       p += (uint64_t)gridDim.x * blockDim.x) {
    keys[p] = kKsEmpty;
    mark[p] = kKsNone;
  }
}

static int ks_clear(KeySet* s, hipStream_t st) {
  hipLaunchKernelGGL(k_ks_fill, dim3(grid_for((int64_t)s->slots, 256)), dim3(256), 0, st, s->keys,
                     s->mark, s->slots);
  MONO_HIP(hipGetLastError());
  s->size = 0;
  s->has_empty_key = false;
  return 0;
}

}  // namespace mono

using namespace mono;

extern "C" {

int mono_keyset_create(int64_t capacity, void** out_set) {
  MONO_REQUIRE(out_set, "mono_keyset_create: null out_set");
  MONO_REQUIRE(capacity >= 1 && capacity < (int64_t)1 << 30, "mono_keyset_create: capacity %lld",
               (long long)capacity);
  KeySet* s = new KeySet();
  s->capacity = capacity;
  uint64_t slots = 1024;
  while (slots < 2 * (uint64_t)(capacity + 2)) slots <<= 1;
  s->slots = slots;
  if (hipMalloc((void**)&s->keys, slots * 8) != hipSuccess ||
      hipMalloc((void**)&s->mark, slots * 4) != hipSuccess ||
      hipMalloc((void**)&s->ctl, 64) != hipSuccess) {
    if (s->keys) (void)hipFree(s->keys);
    if (s->mark) (void)hipFree(s->mark);
    delete s;
    return fail("mono_keyset_create: cannot allocate %llu cells", (unsigned long long)slots);
  }
  int rc = ks_clear(s, nullptr);
  if (rc == 0 && hipStreamSynchronize(nullptr) != hipSuccess) rc = fail("mono_keyset_create: sync failed");
  if (rc != 0) {
    (void)hipFree(s->keys); (void)hipFree(s->mark); (void)hipFree(s->ctl);
    delete s;
    return rc;
  }
  *out_set = s;
  return 0;
}

int mono_keyset_destroy(void* set) {
  if (!set) return 0;
  KeySet* s = (KeySet*)set;
  (void)hipFree(s->keys); (void)hipFree(s->mark); (void)hipFree(s->ctl);
  s->ws.release();
  delete s;
  return 0;
}

int mono_keyset_size(void* set, int64_t* out_size) {
  MONO_REQUIRE(set && out_size, "mono_keyset_size: null argument");
  *out_size = ((KeySet*)set)->size;
  return 0;
}

int mono_keyset_insert(void* set, const int64_t* ids, int64_t n, int64_t* h_dropped, void* stream) {
  MONO_REQUIRE(set && h_dropped, "mono_keyset_insert: null argument");
  MONO_REQUIRE(n >= 0 && n < (int64_t)0x7FFFFFF0ll, "mono_keyset_insert: n %lld", (long long)n);
  *h_dropped = 0;
  if (n == 0) return 0;
  MONO_REQUIRE(ids, "mono_keyset_insert: null ids");
  KeySet* s = (KeySet*)set;
  hipStream_t st = (hipStream_t)stream;
  const size_t n_al = ((size_t)n + 63) & ~(size_t)63;
  size_t scan_tmp = 0;
  MONO_HIP(rocprim::exclusive_scan(nullptr, scan_tmp, (int32_t*)nullptr, (int32_t*)nullptr, 0,
                                   (size_t)n, rocprim::plus<int32_t>(), st));
  const size_t off_cell = 0, off_flag = n_al * 4, off_excl = off_flag + n_al * 4,
               off_tmp = off_excl + n_al * 4;
  MONO_TRY(s->ws.ensure(off_tmp + scan_tmp + 256));
  char* base = (char*)s->ws.ptr;
  uint32_t* cell_of = (uint32_t*)(base + off_cell);
  int32_t* flag = (int32_t*)(base + off_flag);
  int32_t* excl = (int32_t*)(base + off_excl);
  void* tmp = base + off_tmp;
  int64_t lo = 0;
  const dim3 blk(256);
  // A round inserts every distinct key of its positions BEFORE the overflow position is known, so
  // it takes at most round_max positions: the set holds <= capacity + 1 keys when a round starts
  // (it is dropped as soon as it holds more) and has >= 2 * (capacity + 2) cells, hence at least
  // one cell stays empty and every probe ends. Every round either commits its positions or drops
  // the set and restarts at the overflow position (which then commits): <= 2 * n / round_max + 2
  // rounds.
  const int64_t round_max = (int64_t)s->slots - (s->capacity + 2) - 1;
  for (;;) {
    const int64_t hi = std::min(n, lo + round_max);
    const int64_t m = hi - lo;
    const dim3 grid(grid_for(m, 256));
    const uint32_t ctl0[4] = {kKsNone, kKsNone, 0, 0};
    MONO_HIP(hipMemcpyAsync(s->ctl, ctl0, sizeof(ctl0), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_ks_insert, grid, blk, 0, st, s->keys, s->mark, s->slots - 1, ids, lo, hi,
                       cell_of, s->ctl);
    hipLaunchKernelGGL(k_ks_flags, grid, blk, 0, st, s->mark, cell_of, s->ctl, lo, hi,
                       (int)s->has_empty_key, flag);
    MONO_HIP(hipGetLastError());
    MONO_HIP(rocprim::exclusive_scan(tmp, scan_tmp, flag, excl, 0, (size_t)m,
                                     rocprim::plus<int32_t>(), st));
    hipLaunchKernelGGL(k_ks_overflow, grid, blk, 0, st, excl, lo, hi, s->size, s->capacity, s->ctl);
    MONO_HIP(hipGetLastError());
    uint32_t h_ctl[4];
    int32_t last[2];
    MONO_HIP(hipMemcpyAsync(h_ctl, s->ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, st));
    MONO_HIP(hipMemcpyAsync(&last[0], excl + (m - 1), 4, hipMemcpyDeviceToHost, st));
    MONO_HIP(hipMemcpyAsync(&last[1], flag + (m - 1), 4, hipMemcpyDeviceToHost, st));
    MONO_HIP(hipStreamSynchronize(st));
    MONO_REQUIRE(h_ctl[3] == 0, "mono_keyset_insert: the set's %llu cells are full (internal error)",
                 (unsigned long long)s->slots);
    if (h_ctl[1] == kKsNone) {  // no overflow inside this round: commit it
      hipLaunchKernelGGL(k_ks_commit, grid, blk, 0, st, s->mark, cell_of, lo, hi);
      MONO_HIP(hipGetLastError());
      s->size += (int64_t)last[0] + last[1];
      if (h_ctl[0] != kKsNone) s->has_empty_key = true;
      if (hi == n) return 0;
      lo = hi;
      continue;
    }
    // the set is dropped right before position p (hopscotch_hash_set.cc:113-122)
    const int64_t p = (int64_t)h_ctl[1];
    int32_t before = 0;
    MONO_HIP(hipMemcpyAsync(&before, excl + (p - lo), 4, hipMemcpyDeviceToHost, st));
    MONO_HIP(hipStreamSynchronize(st));
    *h_dropped += s->size + before;
    MONO_TRY(ks_clear(s, st));
    lo = p;
  }
}

int mono_keyset_steal(void* set, int64_t* out_ids, int64_t out_cap, int64_t* h_count, void* stream) {
  MONO_REQUIRE(set && h_count, "mono_keyset_steal: null argument");
  KeySet* s = (KeySet*)set;
  hipStream_t st = (hipStream_t)stream;
  *h_count = s->size;
  if (s->size == 0) return 0;
  MONO_REQUIRE(out_ids && out_cap >= s->size, "mono_keyset_steal: room for %lld ids, the set holds %lld",
               (long long)out_cap, (long long)s->size);
  MONO_HIP(hipMemsetAsync(s->ctl + 2, 0, 4, st));
  const int64_t in_cells = s->size - (s->has_empty_key ? 1 : 0);
  hipLaunchKernelGGL(k_ks_steal, dim3(grid_for((int64_t)s->slots, 256)), dim3(256), 0, st, s->keys,
                     s->slots, out_ids, in_cells, s->ctl);
  MONO_HIP(hipGetLastError());
  if (s->has_empty_key) {
    const int64_t k = kKsEmpty;
    MONO_HIP(hipMemcpyAsync(out_ids + in_cells, &k, 8, hipMemcpyHostToDevice, st));
  }
  MONO_TRY(ks_clear(s, st));
  MONO_HIP(hipStreamSynchronize(st));
  return 0;
}

}  // extern "C"
