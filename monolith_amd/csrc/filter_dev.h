// filter_dev.h — device view + device routines of the sliding hash filter.
//
// Restates runtime/neo_filter/hash_filter.h:38-164 (HashFilter<uint16_t>: 12-bit signature |
// 4-bit saturating count, linear probe) and runtime/neo_filter/sliding_hash_filter.cc:57-113
// (ring of blocks, look forward <= 2 blocks for a cell, look backward for an old count) for a
// many-writers setting: cells are claimed / incremented with 32-bit atomicCAS on the word
// holding the 16-bit cell. One lane handles one DISTINCT fid per launch (callers group
// occurrences first), so two lanes only ever meet on a cell when their fids alias — which the
// reference treats as one counter too.
//
// The probe start is mix64(fid) % total_size: the reference uses absl::Hash, which is seeded
// per process and therefore unpinnable (hash_filter.h:191); the CPU oracle uses mix64 as well.
// Head rotation (sliding_hash_filter.cc:83-87) is applied between launches (k_filter_rotate).
#ifndef MONO_FILTER_DEV_H_
#define MONO_FILTER_DEV_H_

#include "mono_common.h"

namespace mono {

struct FilterDev {
  int kind;                   // 0 = none / dummy (never filters), 1 = sliding
  uint16_t* cells;            // [n_blocks * stride]
  uint32_t* num_elements;     // [n_blocks][kFilterShards]: a block's element count is the sum of
                              // its shards (one word would take every claim of a batch in turn:
                              // ~90 atomics per microsecond, 0.94 ms for config 5's 85 k new ids)
  uint32_t* state;            // [0] head, [1] head_increment, [2] failure_count, [3] rotate flag
  const int32_t* thresholds;  // [32768] per-slot occurrence threshold
  uint32_t n_blocks;
  uint32_t max_backward;      // split_num - 2
  uint64_t total_size;        // capacity * 1.2 per block
  uint64_t stride;            // total_size + 64 (hash_filter.h:94)
  uint64_t capacity;          // per block
};

// Host side (filter.hip)
int filter_dev_view(void* filter, FilterDev* out, hipStream_t st);
int filter_after_batch(void* filter, hipStream_t st, int64_t new_ids = 0);

constexpr int kFilterMaxStep = 16;  // sliding_hash_filter.h:98
constexpr int kFilterShards = 64;   // counter shards per block, each on a cache line of its own
constexpr int kFilterShardStride = 32;  // words between shards (128 B)
__host__ __device__ inline size_t filter_counter_words(uint32_t n_blocks) {
  return (size_t)n_blocks * kFilterShards * kFilterShardStride;
}
__device__ __forceinline__ uint32_t* filter_counter(const FilterDev& f, uint32_t blk, uint32_t shard) {
  return f.num_elements + ((size_t)blk * kFilterShards + shard) * kFilterShardStride;
}
constexpr uint32_t kFilterMaxCount = 15;

__device__ __forceinline__ uint32_t filter_sig(uint64_t fid) {  // hash_filter.h:149
  return (uint32_t)(((fid >> 17) | (fid << 15)) & 0xFFFull);
}

// ops/neo_filter_tf_bridge.cc:37-39
__device__ __forceinline__ int filter_threshold(const FilterDev& f, int64_t fid) {
  return f.thresholds[slot_id_v2((uint64_t)fid)];
}

__device__ __forceinline__ uint32_t cell_load(const FilterDev& f, uint64_t idx) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(f.cells) + (idx >> 1);
  uint32_t v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (v >> ((idx & 1) * 16)) & 0xFFFFu;
}

// CAS a 16-bit cell; returns the cell value observed (== expect on success).
__device__ __forceinline__ uint32_t cell_cas(const FilterDev& f, uint64_t idx, uint32_t expect,
                                             uint32_t desired) {
  uint32_t* w = reinterpret_cast<uint32_t*>(f.cells) + (idx >> 1);
  const int sh = (int)(idx & 1) * 16;
  uint32_t cur = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (;;) {
    uint32_t cell = (cur >> sh) & 0xFFFFu;
    if (cell != expect) return cell;
    uint32_t next = (cur & ~(0xFFFFu << sh)) | (desired << sh);
    uint32_t seen = atomicCAS(w, cur, next);
    if (seen == cur) return expect;
    cur = seen;
  }
}

// HashFilter::find (hash_filter.h:110-126): index of the first cell that is empty or carries
// the signature, or -1 after max_step probes. *value = the cell content seen.
__device__ inline int64_t block_find(const FilterDev& f, uint32_t blk, uint64_t fid,
                                     uint32_t* value) {
  const uint32_t sig = filter_sig(fid);
  uint64_t pos = mix64(fid) % f.total_size;
  const uint64_t base = (uint64_t)blk * f.stride;
  for (int step = 0; step < kFilterMaxStep; ++step) {
    uint32_t v = cell_load(f, base + pos);
    if (v == 0 || (v >> 4) == sig) {
      *value = v;
      return (int64_t)pos;
    }
    ++pos;
    if (pos == f.stride) pos = 0;
  }
  return -1;
}

// Backward look (sliding_hash_filter.cc:74-82): count in the newest older block that holds fid.
__device__ inline bool filter_old_count(const FilterDev& f, uint32_t head, uint32_t head_inc,
                                        uint64_t fid, uint32_t* old_count) {
  uint32_t looks = head_inc < f.max_backward ? head_inc : f.max_backward;
  uint32_t blk = head == 0 ? f.n_blocks - 1 : head - 1;
  for (uint32_t i = 0; i < looks; ++i) {
    uint32_t v;
    int64_t p = block_find(f, blk, fid, &v);
    if (p >= 0 && v != 0) {
      *old_count = v & kFilterMaxCount;
      return true;
    }
    blk = blk == 0 ? f.n_blocks - 1 : blk - 1;
  }
  return false;
}

// SlidingHashFilter::get (sliding_hash_filter.cc:91-113).
__device__ inline uint32_t filter_get(const FilterDev& f, uint64_t fid) {
  const uint32_t head = f.state[0], head_inc = f.state[1];
  uint32_t blk = head;
  for (int look = 0; look < 2; ++look) {
    uint32_t v;
    int64_t p = block_find(f, blk, fid, &v);
    if (p >= 0) {
      if (v != 0) return v & kFilterMaxCount;
      uint32_t old;
      return filter_old_count(f, head, head_inc, fid, &old) ? old : 0;
    }
    blk = blk + 1 == f.n_blocks ? 0 : blk + 1;
  }
  return kFilterMaxCount;
}

// SlidingHashFilter::add (sliding_hash_filter.cc:57-89) minus the rotation; returns the old count.
__device__ inline uint32_t filter_add(const FilterDev& f, uint64_t fid, uint32_t count) {
  if (count > kFilterMaxCount) count = kFilterMaxCount;
  const uint32_t head = f.state[0], head_inc = f.state[1];
  const uint32_t sig = filter_sig(fid);
  for (int attempt = 0; attempt < 64; ++attempt) {
    // look forward for a valid cell
    uint32_t blk = head, v = 0;
    int64_t p = -1;
    for (int look = 0; look < 2 && p < 0; ++look) {
      p = block_find(f, blk, fid, &v);
      if (p < 0) blk = blk + 1 == f.n_blocks ? 0 : blk + 1;
    }
    if (p < 0) {
      atomicAdd(&f.state[2], 1u);
      return kFilterMaxCount;
    }
    const uint64_t idx = (uint64_t)blk * f.stride + (uint64_t)p;
    if (v != 0) {  // HashFilterIterator::add on an occupied cell (hash_filter.h:55-60)
      for (;;) {
        uint32_t c = v & kFilterMaxCount;
        uint32_t nv = (c + count >= kFilterMaxCount) ? (v | kFilterMaxCount) : (v + count);
        uint32_t seen = cell_cas(f, idx, v, nv);
        if (seen == v) return c;
        v = seen;
        if (v == 0 || (v >> 4) != sig) break;  // cleared underneath us: start over
      }
      continue;
    }
    uint32_t old = 0;
    filter_old_count(f, head, head_inc, fid, &old);
    uint32_t add = old + count;
    if (add > kFilterMaxCount) add = kFilterMaxCount;
    uint32_t seen = cell_cas(f, idx, 0u, (sig << 4) + add);
    if (seen == 0u) {  // hash_filter.h:42-53
      atomicAdd(filter_counter(f, blk, (uint32_t)(mix64(fid) >> 58)), 1u);
      return old;
    }
    // another fid claimed the cell first: probe again
  }
  atomicAdd(&f.state[2], 1u);
  return kFilterMaxCount;
}

}  // namespace mono

#endif  // MONO_FILTER_DEV_H_
