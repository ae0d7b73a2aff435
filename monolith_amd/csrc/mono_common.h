// Shared host/device helpers of libmonolith_hip.so (gfx950 only).
#ifndef MONO_COMMON_H_
#define MONO_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/monolith_hip.h"

namespace mono {

// ---- error plumbing: no exception crosses the C ABI --------------------------------------
std::string& last_error_ref();
int fail(const char* fmt, ...);

#define MONO_HIP(expr)                                                                    \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess)                                                                \
      return ::mono::fail("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,              \
                          hipGetErrorString(e__));                                        \
  } while (0)

#define MONO_REQUIRE(cond, ...)                \
  do {                                         \
    if (!(cond)) return ::mono::fail(__VA_ARGS__); \
  } while (0)

#define MONO_TRY(expr)        \
  do {                        \
    int rc__ = (expr);        \
    if (rc__ != 0) return rc__; \
  } while (0)

// ---- hashing (the oracle uses the same functions) ------------------------------------------
__host__ __device__ inline uint64_t mix64(uint64_t x) {  // splitmix64 finalizer
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}

// data/training_instance/cc/reader_util.h:35-37
__host__ __device__ inline int slot_id_v2(uint64_t fid) { return (int)((fid >> 48) & 0x7FFF); }

// Counter-based stand-in for the reference's thread_local mt19937 uniform initializer
// (random_uniform_initializer.cc:30-37): distribution parity only.
__host__ __device__ inline float uniform_init(uint64_t seed, int64_t id, int elem, float lo,
                                              float hi) {
  uint64_t h = mix64(seed ^ mix64((uint64_t)id) ^ (0x9e3779b97f4a7c15ULL * (uint64_t)(elem + 1)));
  float u = (float)(h >> 40) * (1.0f / 16777216.0f);
  return lo + (hi - lo) * u;
}

// Grid sizing for memory-bound kernels: enough workgroups to fill 256 CUs several times over,
// grid-stride the rest (cdna_hip_programming.md Guideline 11).
inline int grid_for(int64_t work_items, int items_per_block, int max_blocks = 256 * 16) {
  int64_t b = (work_items + items_per_block - 1) / items_per_block;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

// Growable device scratch owned by a handle. grow() may hipFree + hipMalloc (synchronising
// the device); steady-state calls do not allocate.
struct Workspace {
  void* ptr = nullptr;
  size_t bytes = 0;
  int ensure(size_t need);
  void release();
};

// common.hip: copy a small host array (per-call metadata: descriptors, plans) to the device without
// blocking the host. hipMemcpyAsync from pageable memory waits until the copy has been carried out,
// behind whatever the stream's hardware queue holds (4 queues per process: that can be another
// stream's kernels — 150 us per reorder in the sharded step); here the bytes go through a ring of
// pinned staging slots first, so the call only enqueues. Larger than a slot: the plain copy.
int upload_async(void* dst_dev, const void* src_host, size_t bytes, hipStream_t st);

// sort.hip: stable LSD radix sort of (uint32 key, uint32 value) pairs on key bits [0, end_bit).
// vin == nullptr sorts the identity permutation (values = input positions). kin / vin are not
// modified; `tmp` needs sort_pairs_tmp_bytes(n, end_bit) bytes of device scratch.
size_t sort_pairs_tmp_bytes(size_t n, int end_bit);
int sort_pairs_u32(const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout,
                   size_t n, int end_bit, void* tmp, size_t tmp_bytes, hipStream_t st);

// sort.hip: stable GROUPING of 64-bit keys (equal keys adjacent, input order inside a group, groups
// in no particular order): key_out[j] / val_out[j] = key / value (vin, or the input position when
// vin == nullptr) of the j-th element of the grouped order. `tmp`: group_keys64_tmp_bytes(n) bytes.
size_t group_keys64_tmp_bytes(size_t n);
int group_keys64(const int64_t* keys, const uint32_t* vin, int64_t* key_out, uint32_t* val_out, size_t n,
                 void* tmp, size_t tmp_bytes, hipStream_t st);

// The fused passes' sticky error word on the current device (never allocates; nullptr only if
// the device cannot be queried) and the message for a non-zero value read back by the caller.
uint32_t* sort_error_word();
int sort_error_report(uint32_t h_word);

// table.hip: out[skeys[p] .. + D) += src[sidx[p]] over the sorted positions p, every segment summed
// left to right (segments longer than 4096 in ordered chunks). `aligned4`: every key is a
// multiple of 4 floats and `out` is 16-byte aligned (enables vector access).
int segment_add_rows(const uint32_t* skeys, const uint32_t* sidx, int64_t n, const float* src,
                     int D, bool aligned4, float* out, hipStream_t st);

// table.hip: out[skeys[p] .. + D) = src_ptrs[sidx[p]] of the segment's first occurrence + the later
// ones in input order, for the segments whose occurrences belong to a class of width D
// (cls[occurrence] -> cls_dim[class]); keys equal to skip_key are ignored.
int segment_set_sum_rows(const uint32_t* skeys, const uint32_t* sidx, int64_t n,
                         const float* const* src_ptrs, const uint32_t* cls, const int32_t* cls_dim,
                         int D, bool aligned4, uint32_t skip_key, float* out, hipStream_t st,
                         int key_unit = 1);
// The same sum with the source row of an occurrence packed into its 32-bit value instead of looked
// up through a pointer array: value = class | sample << sbits | seq << (sbits + bbits), source row =
// tab[class].base + sample * tab[class].stride + seq * tab[class].seq_stride (one dependent memory
// level and 12 bytes per occurrence less than segment_set_sum_rows). Keys count units of
// `key_unit` floats: out[key * key_unit .. + D).
struct SrcTab {
  const float* base;
  int32_t stride, seq_stride;
};
int segment_set_sum_packed(const uint32_t* skeys, const uint32_t* svals, int64_t n, const SrcTab* tab,
                           int sbits, int bbits, const int32_t* cls_dim, int D, int key_unit,
                           bool aligned4, uint32_t skip_key, float* out, hipStream_t st);
// host-side call timing, printed at exit when MONO_HOST_PROF is set (table.hip)
bool host_prof_on();
double host_now();
void host_tick(const char* name, double* t0);  // adds (now - *t0) to `name`, restarts *t0
size_t segment_plan_bytes(int64_t n, int D);
int segment_plan_classify(const uint32_t* skeys, const uint32_t* sidx, int64_t n, int D, char* q_base,
                          hipStream_t st);
int segment_set_sum_planned(const uint32_t* skeys, const uint32_t* sidx, int64_t n, const float* src,
                            int D, bool aligned4, float* out, char* q_base, hipEvent_t ready,
                            hipStream_t st);

// table.hip, for the fused lookup + pool of layout.hip: the slot of every id (0xFFFFFFFF = not in
// the table), and where the rows of a table live (weights first: columns [0, dim) of a row of
// row_floats floats). The view is valid until the next call that can grow the table.
int table_resolve_slots(void* table, const int64_t* ids, int64_t n, uint32_t* slots, hipStream_t st);
int table_row_view(void* table, const float** rows, int* dim, int* row_floats);
// the same slots, taken as a LOOKUP of the table (slot record + prepared update order for the
// update of the same ids that follows); *slots = nullptr when the record is switched off
int table_lookup_slots(void* table, const int64_t* ids, int64_t n, const uint32_t** slots,
                       hipStream_t st);

}  // namespace mono

#endif  // MONO_COMMON_H_
