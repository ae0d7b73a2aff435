// table.hip — HBM-resident open-addressing embedding table for MI355X (gfx950).
//
// Replaces runtime/hash_table/cuckoohash/* + entry_accessor + optimizer/* + the table half of
// ops/embedding_hash_table_tf_bridge.cc. Only set semantics of the cuckoo map are observable
// (SURVEY.md §8a-A2), so placement is ours:
//
//   keys  int64[cap]            power-of-two slots, 8-slot (64 B) buckets, linear probing over
//                               buckets; EMPTY = -1 (hipMemset 0xFF clears), TOMB = -2
//   rows  float[(cap+2) * R]    slot s -> R = dim + optimizer-state floats, contiguous so one
//                               update touches one 128 B line at dim 16 + Adagrad
//   ts    uint32[cap+2]         last update time (entry_defs.h:31-33)
//   special uint32[2]           presence of the two keys that collide with the sentinels; their
//                               rows live in slots cap and cap+1, so every int64 is a valid key
//
// Execution model: an 8-lane sub-group of a 64-lane wavefront owns one id. The 8 lanes load one
// bucket (one coalesced 64 B access), vote with a wave ballot, then move the row with the same
// 8 lanes (coalesced 8/16 B per lane). Inserts claim a slot with a 64-bit atomicCAS.
// Writers never race on a row: every write path first groups occurrences by slot (stable LSD
// radix sort of 32-bit slots; behind a hash filter the ids the table does not hold are grouped by
// id instead) and one sub-group walks a key's occurrences in input order — a key with more than
// 96 of them goes to one wave of k_long_chain, which adds its packed rows in the same order —
// which is also what makes duplicate handling bit-exact with the reference's sequential loops at
// any number of occurrences (SURVEY.md §8a closed forms; bridge :241-266).
#include <algorithm>
#include <ctime>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_fp16.h>

#include "filter_dev.h"
#include "mono_common.h"
#include "chain_dev.h"

namespace mono {

constexpr int64_t kEmpty = -1;
constexpr int64_t kTomb = -2;
constexpr int kBucket = 8;
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
constexpr int kMaxSeg = 8;
constexpr double kMaxLoad = 0.6;    // rehash when live + tombstones would exceed this
constexpr double kGrowTarget = 0.3;  // load right after a rehash

enum { MODE_ASSIGN = 0, MODE_ASSIGN_ADD = 1, MODE_OPT_SEQ = 2, MODE_OPT_DEDUP = 3, MODE_IMPORT = 4,
       // stateless segment op: row = first occurrence, + the others in input order ("first touch
       // copies", fused_embedding_to_layout.cc:385-400); no optimizer, no row read
       MODE_SET_SUM = 5 };
enum { OPTK_SGD = 0, OPTK_ADAGRAD = 1, OPTK_FTRL = 2, OPTK_MULTI = 3 };

struct SegDev {
  int dim, opt_kind, w_off, st_off, init_kind;
  float init_acc, wd, l1, l2, beta, init_a, init_b;
  float momentum, epsilon, rho, conf_lr;
  int nesterov;
  float beta1, beta2;
  int comp_kind;    // serving tables: MONO_COMP_*
  float cq0, cq1;   // FixedR8: step, half step; OneBit: amplitude, scale
  int sr_pre;       // fp16 stochastic rounding: first decorated segment before this one, or -1
  int sr_post;      // ... this segment is decorated
};

// per-element optimizer state slots of a row: norm | (FTRL) zero; n; accum | accum_update;
// m | v (| vhat)
__host__ __device__ inline int opt_nstate(int kind) {
  switch (kind) {
    case MONO_OPT_SGD: case MONO_OPT_MOVING_AVERAGE: return 0;
    case MONO_OPT_FTRL: case MONO_OPT_ADADELTA: case MONO_OPT_ADAM:
    case MONO_OPT_BATCH_SOFTMAX: return 2;  // one int64 (dim is 1)
    case MONO_OPT_AMSGRAD: return 3;
    default: return 1;
  }
}
// per-row scalars behind a segment's state slots: Adam / AMSGrad keep beta1^t, beta2^t
__host__ __device__ inline int opt_nscalar(int kind) {
  return (kind == MONO_OPT_ADAM || kind == MONO_OPT_AMSGRAD) ? 2 : 0;
}
__host__ __device__ inline int opt_state_floats(int kind, int dim) {
  return opt_nstate(kind) * dim + opt_nscalar(kind);
}

constexpr int kLiveStripe0 = 8, kLiveStripes = 64;
constexpr int kCounterWords = kLiveStripe0 + 8 * kLiveStripes;
struct TableDev {  // passed to kernels by value
  int64_t* keys;
  uint32_t* ts;
  float* rows;
  uint32_t* special;
  unsigned long long* counters;  // [0] live rows (+ the stripes below), [1] tombstones, [2] export
                                 // cursor; [kLiveStripe0 + 8 k], k < 64: rows inserted since, striped
                                 // over 64 lines (one word takes ~300 atomics per microsecond: a batch
                                 // that inserts 17 k rows spent 60 us on it)
  uint64_t cap;
  uint64_t bucket_mask;
  int dim, row_floats, n_seg;
  int default_expire_days;
  int serving;  // MONO_ENTRY_SERVING: writes store compressor round trips (comp_q)
  int any_sr;   // some segment rounds to fp16 stochastically (generic path, no long segments)
  uint64_t seed;
  const int32_t* slot_expire;  // int32[32768] or nullptr
  SegDev seg[kMaxSeg];
};

// one row was claimed by this sub-group (`claimed`, wave-divergent): the claims the active lanes
// of the wave made together go to one stripe with one atomic
__device__ __forceinline__ void count_claimed(const TableDev& t, bool claimed_sub0) {
  const unsigned long long cm = __ballot(claimed_sub0);
  if (cm && (int)(threadIdx.x & 63) == __ffsll((long long)cm) - 1) {
    const uint32_t stripe = (uint32_t)(blockIdx.x * 4u + (threadIdx.x >> 6)) & (kLiveStripes - 1);
    atomicAdd(&t.counters[kLiveStripe0 + 8 * stripe], (unsigned long long)__popcll(cm));
  }
}

struct Table {
  TableDev d{};
  std::vector<mono_segment_cfg> segs;
  int opt_tmpl = OPTK_SGD;  // kernel specialisation
  int vec = 1;              // floats per lane per access
  int64_t used_upper = 0;   // host upper bound of live + tombstones
  // Slots the last mono_table_lookup resolved, position by position: the write call that
  // follows on the same ids (forward lookup, backward update) reads them back instead of
  // probing again. Valid while no key has moved or been removed (rehash / evict / clear reset
  // cache_n); an insert never moves an existing key. A position whose id differs, or that
  // missed in the lookup, probes as usual.
  int64_t* cache_ids = nullptr;
  uint32_t* cache_slots = nullptr;
  int64_t cache_n = 0, cache_cap = 0;
  // Prepared update order ("lookup, then update the same ids": the training step). When the calls
  // on this table follow that pattern, mono_table_lookup resolves the slots first (k_probe), and
  // the grouping of the positions by slot that the update needs (radix sort + k_classify) runs
  // on a stream of its own BESIDE the row gather of the same lookup: the sort is latency-bound and
  // leaves HBM idle, the gather is the opposite. The write call then checks on the device that
  // the ids are still the ones the lookup saw and that all of them were found (k_verify, one
  // 8-byte read-back), and goes straight to the apply kernels; anything else takes the full path.
  // Valid under the slot record's conditions (no rehash / evict / clear since the lookup).
  struct Prep {
    Workspace ws;                 // sorted slots | positions | sort scratch | queues + heads
    int64_t n = 0;                // positions the prepared order covers; 0 = none
    uint32_t* sslots = nullptr;
    uint32_t* sidx = nullptr;
    char* q_base = nullptr;
    int mode_class = -1;          // what k_classify was run for (-1: not run)
    hipStream_t stream = nullptr;
    hipEvent_t ev_probe = nullptr, ev_ready = nullptr, ev_verify = nullptr, ev_gated = nullptr;
    // row gather of the prepared lookup: throttled (the update follows at once and waits for the
    // order) or at full depth (the order is ready long before the update asks for it)
    bool gather_full = false;
    int slack_run = 0, full_backoff = 0, full_backoff_len = 64;
    // [0] ids the lookup did not find  [1] positions k_verify found changed (running count).
    // [0] is zeroed by every k_probe and filled by k_verify from the per-workgroup miss counts that
    // k_probe left in d_blk_miss (plain stores: with a hash filter 5 % of a batch misses, and 85 k
    // atomics on one word made k_probe 198 us instead of 42) — exactly the last lookup's count,
    // whatever happened to the lookups before it.
    uint32_t* d_flags = nullptr;
    uint32_t* d_blk_miss = nullptr;  // [kProbeMaxBlocks]
    int probe_blocks = 0;
    uint32_t* h_flags = nullptr;  // pinned host copy
    // recorded on the caller's stream behind the update that used this order: a hint that
    // recycles the slot (its arrays, the slot record swapped out with it) may come from another
    // stream than the table's lookups and updates, and waits for it
    hipEvent_t ev_consumed = nullptr;
    bool consumed_set = false;
    uint32_t seen_changed = 0;    // d_flags[1] as of the last read-back (it only grows)
    bool chain_pending = false;   // a sort + classify chain was launched and the caller's stream
                                  // has not waited for it since
    bool hinted = false;          // built by mono_table_lookup_hint: valid only if d_flags[2] == 0
    bool grouped_tail = false;    // the order's tail of absent ids is grouped by id (k_probe surrogates)
    uint32_t sur_base = 0;        // ... whose sort keys start here (0: none); every k_classify of
                                  // this order must know, or it takes them for slots
  } prep;
  // mono_table_lookup_hint: the slots and the update order of the NEXT lookups' ids (a queue of up
  // to kHintDepth, oldest first), resolved ahead on the table's chain streams while the current
  // steps are still running (an update through the prepared order never changes the key set, so
  // the probe does not have to wait for it). Beside an update the latency-bound sort runs several
  // times slower than alone, which is why a hint one lookup ahead is not enough: two ahead gives
  // the chain a whole step. The lookup that follows swaps the oldest hint into `prep` (and the
  // hinted slot record into cache_*), checks on the device that its ids are the hinted ones
  // (k_hint_fix re-probes any that are not and flags it: the order is then not used, and the
  // queue is dropped; it also re-probes the ids the hint recorded as absent, which a gated update
  // may have inserted since, without giving up the order) and only gathers the rows. Every call that may change the key set drops the
  // queue.
  static constexpr int kHintDepth = 3;
  struct Hint {
    Prep p;
    int64_t* ids = nullptr;
    uint32_t* slots = nullptr;
    int64_t cap = 0;
  } hints[kHintDepth];
  int hint_head = 0, hint_count = 0;
  bool hints_kept = false;  // a gated update has kept queued hints: their absent ids get re-probed
  // the streams the update-order chains run on (all Preps share them, alternating: the process
  // has four hardware queues, and the caller's stream and the apply stream take two)
  hipStream_t chain_stream[2] = {nullptr, nullptr};
  int preps_made = 0;
  hipEvent_t ev_hint_src = nullptr;  // the caller's stream at the time of the hint (its ids are ready)
  hipEvent_t ev_keys = nullptr;      // behind the last call that may have changed the key set
  bool ev_keys_set = false;
  bool expect_update = false;     // the last update came right after one lookup of as many ids
  int lookups_since_update = 0;
  int prep_strikes = 0;           // prepared orders in a row that the update could not use
  int updates_since_strike = 0;
  int last_mode_class = 1;
  int64_t max_update_ts = 0;
  // periodic eviction (bridge :69-100): every evict_every_sec of wall clock, checked at the entry
  // of a write call; -1 = off. test_clock >= 0 stands in for the wall clock.
  int64_t evict_every_sec = -1, last_evict_wall = 0, test_clock = -1, periodic_evictions = 0;
  int32_t* slot_expire_dev = nullptr;
  Workspace ws;
  Workspace ws_gated;  // the gated remainder of a write call with a hash filter
  Workspace ws_pack_gated;  // ... and the packed rows of its hot ids
  int32_t* h_unk_cnt = nullptr;  // pinned: k_unk_count's per-chunk counts (write_path_gated)
  bool gated_seen = false;       // the last write call went through a hash filter
  // mono_table_path_counts: [0] write calls served by a prepared order, [1] write calls that
  // tried one and fell back, [2] lookups served by a hint, [3] hint queues dropped non-empty
  int64_t path_counts[4] = {0, 0, 0, 0};
  // optional per-phase timing of the write path (mono_table_profile)
  bool profile = false;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_k[2] = {nullptr, nullptr};  // around k_apply alone, on the stream it runs on
  bool ev_k_valid = false;
  // fork / join of the long-segment kernel beside the sub-group kernel
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_pack = nullptr;
  Workspace ws_pack;  // packed rows of the long segments (k_long_pack -> k_long_chain)
};

// ---------------------------------------------------------------------------------------------
// device helpers

__device__ __forceinline__ int lane_sub() { return threadIdx.x & 7; }
__device__ __forceinline__ int lane_gshift() { return (threadIdx.x & 63) & ~7; }
__device__ __forceinline__ uint32_t group_bits(bool pred, int gshift) {
  return (uint32_t)((__ballot(pred) >> gshift) & 0xFFull);
}

__device__ inline uint32_t group_find(const TableDev& t, int64_t key, int sub, int gshift) {
  if (key == kEmpty || key == kTomb) {
    int j = key == kTomb;
    return t.special[j] ? (uint32_t)(t.cap + j) : kNoSlot;
  }
  uint64_t b = mix64((uint64_t)key) & t.bucket_mask;
  for (uint64_t probes = 0; probes <= t.bucket_mask; ++probes) {
    int64_t k = t.keys[b * kBucket + sub];
    uint32_t m = group_bits(k == key, gshift);
    if (m) return (uint32_t)(b * kBucket + (__ffs(m) - 1));
    if (group_bits(k == kEmpty, gshift)) return kNoSlot;
    b = (b + 1) & t.bucket_mask;
  }
  return kNoSlot;
}

// group_find with the first bucket of the probe sequence already loaded (`k` = this lane's key
// of bucket `b`): lets a sub-group keep the first-bucket reads of several ids in flight at once.
__device__ inline uint32_t group_find_pre(const TableDev& t, int64_t key, uint64_t b, int64_t k,
                                          int sub, int gshift) {
  if (key == kEmpty || key == kTomb) {
    int j = key == kTomb;
    return t.special[j] ? (uint32_t)(t.cap + j) : kNoSlot;
  }
  for (uint64_t probes = 0; probes <= t.bucket_mask; ++probes) {
    uint32_t m = group_bits(k == key, gshift);
    if (m) return (uint32_t)(b * kBucket + (__ffs(m) - 1));
    if (group_bits(k == kEmpty, gshift)) return kNoSlot;
    b = (b + 1) & t.bucket_mask;
    k = t.keys[b * kBucket + sub];
  }
  return kNoSlot;
}

// Find `key` or claim the first empty slot of its probe sequence. Slots only go EMPTY -> key
// inside a launch, a possibly stale bucket image can only show EMPTY where a key already
// sits, and every believed-EMPTY slot is confirmed by the CAS, so concurrent sub-groups
// inserting the same key always converge on one slot.
__device__ inline uint32_t group_find_or_claim(const TableDev& t, int64_t key, int sub, int gshift,
                                               bool* claimed) {
  *claimed = false;
  if (key == kEmpty || key == kTomb) {
    int j = key == kTomb;
    bool won = false;
    if (sub == 0) won = atomicCAS(&t.special[j], 0u, 1u) == 0u;
    *claimed = group_bits(won, gshift) != 0;
    return (uint32_t)(t.cap + j);
  }
  uint64_t b = mix64((uint64_t)key) & t.bucket_mask;
  for (uint64_t probes = 0; probes <= t.bucket_mask; ++probes) {
    int64_t k = t.keys[b * kBucket + sub];
    for (;;) {
      uint32_t m = group_bits(k == key, gshift);
      if (m) return (uint32_t)(b * kBucket + (__ffs(m) - 1));
      uint32_t e = group_bits(k == kEmpty, gshift);
      if (!e) break;
      int pos = __ffs(e) - 1;
      bool won = false;
      if (sub == pos) {
        unsigned long long old =
            atomicCAS(reinterpret_cast<unsigned long long*>(&t.keys[b * kBucket + pos]),
                      (unsigned long long)kEmpty, (unsigned long long)key);
        won = (int64_t)old == kEmpty;
        k = won ? key : (int64_t)old;
      }
      if (group_bits(won, gshift)) {
        *claimed = true;
        return (uint32_t)(b * kBucket + pos);
      }
    }
    b = (b + 1) & t.bucket_mask;
  }
  return kNoSlot;  // table full: the host-side load bound makes this unreachable
}

// entry_accessor.cc:139-143: initializer for the weights, optimizer Init for the state.
__device__ inline float row_init_value(const TableDev& t, int64_t key, int e) {
  if (e < t.dim) {
    for (int s = 0; s < t.n_seg; ++s) {
      const SegDev& c = t.seg[s];
      if (e >= c.w_off && e < c.w_off + c.dim) {
        switch (c.init_kind) {
          case MONO_INIT_ZEROS: return 0.f;
          case MONO_INIT_ONES: return 1.f;
          case MONO_INIT_CONSTANT: return c.init_a;
          default: return uniform_init(t.seed, key, e, c.init_a, c.init_b);
        }
      }
    }
    return 0.f;
  }
  for (int s = 0; s < t.n_seg; ++s) {
    const SegDev& c = t.seg[s];
    const int len = opt_state_floats(c.opt_kind, c.dim);
    // Adagrad / FTRL: norm starts at initial_accumulator_value; Adam / AMSGrad: the row scalars
    // start at beta1, beta2 (adam_optimizer.cc:40-51); every other state starts at 0
    if (e >= c.st_off && e < c.st_off + len) {
      const int k = e - c.st_off - opt_nstate(c.opt_kind) * c.dim;
      if (k >= 0) return k == 0 ? c.beta1 : c.beta2;
      return (e - c.st_off < c.dim && c.opt_kind <= MONO_OPT_FTRL) ? c.init_acc : 0.f;
    }
  }
  return 0.f;
}

__device__ inline void group_init_row(const TableDev& t, int64_t key, uint32_t slot, int sub) {
  float* row = t.rows + (uint64_t)slot * t.row_floats;
  for (int e = sub; e < t.row_floats; e += 8) row[e] = row_init_value(t, key, e);
}

__device__ inline int expire_days(const TableDev& t, int64_t key) {
  return t.slot_expire ? t.slot_expire[slot_id_v2((uint64_t)key)] : t.default_expire_days;
}
// cuckoo_embedding_hash_table.cc:230
__device__ inline bool is_expired(const TableDev& t, int64_t key, uint32_t ts, int64_t now) {
  return now - (int64_t)ts >= (int64_t)expire_days(t, key) * 86400;
}

template <int V>
struct Vec;
template <>
struct Vec<1> { using T = float; };
template <>
struct Vec<2> { using T = float2; };
template <>
struct Vec<4> { using T = float4; };

template <int V>
__device__ __forceinline__ void vload(float (&r)[V], const float* p) {
  typename Vec<V>::T v = *reinterpret_cast<const typename Vec<V>::T*>(p);
  const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
  for (int i = 0; i < V; ++i) r[i] = f[i];
}
template <int V>
__device__ __forceinline__ void vstore(float* p, const float (&r)[V]) {
  typename Vec<V>::T v;
  float* f = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int i = 0; i < V; ++i) f[i] = r[i];
  *reinterpret_cast<typename Vec<V>::T*>(p) = v;
}

// ---------------------------------------------------------------------------------------------
// kernels

// B1: ops/hash_table_lookup_op.cc:36-73 -> cuckoo_embedding_hash_table.cc:151-160.
// A probe is two dependent random reads (key bucket, then row); with one id per sub-group at a
// time the chip holds only ~4 MB of reads in flight, which at ~1.5 us of loaded HBM latency caps
// the kernel near 2.7 TB/s (Little's law). Each sub-group therefore works on kIlp ids at once:
// kIlp bucket reads are issued back to back, then kIlp row reads.
constexpr int kIlp = 4;
#ifndef MONO_ILP_LOOKUP
#define MONO_ILP_LOOKUP 4
#endif
constexpr int kIlpL = MONO_ILP_LOOKUP;  // ids in flight per sub-group in k_lookup
template <int V>
__global__ void __launch_bounds__(256) k_lookup(TableDev t, const int64_t* __restrict__ ids,
                                                int64_t n, float* __restrict__ out,
                                                int64_t* __restrict__ cache_ids,
                                                uint32_t* __restrict__ cache_slots) {
  const int sub = lane_sub(), gshift = lane_gshift();
  const int64_t ngroups = (int64_t)gridDim.x * (blockDim.x / 8);
  const int D = t.dim;
  for (int64_t i0 = ((int64_t)blockIdx.x * (blockDim.x / 8) + threadIdx.x / 8) * kIlpL; i0 < n;
       i0 += ngroups * kIlpL) {
    int64_t key[kIlpL], k0[kIlpL];
    uint64_t b[kIlpL];
    uint32_t slot[kIlpL];
#pragma unroll
    for (int q = 0; q < kIlpL; ++q) key[q] = i0 + q < n ? ids[i0 + q] : kEmpty;
#pragma unroll
    for (int q = 0; q < kIlpL; ++q) {
      b[q] = mix64((uint64_t)key[q]) & t.bucket_mask;
      k0[q] = t.keys[b[q] * kBucket + sub];
    }
#pragma unroll
    for (int q = 0; q < kIlpL; ++q) slot[q] = group_find_pre(t, key[q], b[q], k0[q], sub, gshift);
    if (cache_ids && sub < kIlpL && i0 + sub < n) {  // lane q of the sub-group records id q
      int64_t ck = key[0];
      uint32_t cs = slot[0];
#pragma unroll
      for (int q = 1; q < kIlpL; ++q)
        if (sub == q) { ck = key[q]; cs = slot[q]; }
      cache_ids[i0 + sub] = ck;
      cache_slots[i0 + sub] = cs;
    }
    for (int e = sub * V; e < D; e += 8 * V) {
      float r[kIlpL][V];
#pragma unroll
      for (int q = 0; q < kIlpL; ++q) {
        if (slot[q] != kNoSlot) {
          vload<V>(r[q], t.rows + (uint64_t)slot[q] * t.row_floats + e);
        } else {
#pragma unroll
          for (int v = 0; v < V; ++v) r[q][v] = 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < kIlpL; ++q)
        if (i0 + q < n) vstore<V>(out + (i0 + q) * D + e, r[q]);
    }
  }
}

// mono_table_lookup in two kernels (prepared update order, see Table::Prep): k_probe resolves
// and records the slot of every position, k_gather_rows copies the rows. Together they move what
// k_lookup moves plus a re-read of the 4-byte slots.
constexpr int kIlpP = 8;  // one id per lane of the sub-group
constexpr int kProbeMaxBlocks = 4096;
__global__ void __launch_bounds__(256) k_probe(TableDev t, const int64_t* __restrict__ ids, int64_t n,
                                               int64_t* __restrict__ cache_ids,
                                               uint32_t* __restrict__ cache_slots,
                                               uint32_t* __restrict__ blk_miss,
                                               uint32_t* __restrict__ zero8,
                                               uint32_t* __restrict__ miss_total,
                                               uint32_t* __restrict__ sort_key,
                                               uint32_t* __restrict__ sur_cells, uint32_t sur_mask,
                                               uint32_t sur_base) {
  __shared__ uint32_t s_miss;
  if (threadIdx.x == 0) s_miss = 0;
  if (blockIdx.x == 0 && threadIdx.x < 16) zero8[threadIdx.x] = 0;  // counters of the prepared queues
  if (blockIdx.x == 0 && threadIdx.x == 8) miss_total[0] = 0;      // k_verify adds this lookup's up
  if (blockIdx.x == 0 && threadIdx.x == 9) miss_total[2] = 0;      // ids that were not the hinted ones
  __syncthreads();
  const int sub = lane_sub(), gshift = lane_gshift();
  const int64_t ngroups = (int64_t)gridDim.x * (blockDim.x / 8);
  uint32_t missed = 0;
  for (int64_t i0 = ((int64_t)blockIdx.x * (blockDim.x / 8) + threadIdx.x / 8) * kIlpP; i0 < n;
       i0 += ngroups * kIlpP) {
    // lane q of the sub-group loads id q (one coalesced 64-byte read) and hands it round
    const int64_t mine = i0 + sub < n ? ids[i0 + sub] : kEmpty;
    int64_t key[kIlpP], k0[kIlpP];
    uint64_t b[kIlpP];
#pragma unroll
    for (int q = 0; q < kIlpP; ++q) key[q] = __shfl(mine, q, 8);
#pragma unroll
    for (int q = 0; q < kIlpP; ++q) {
      b[q] = mix64((uint64_t)key[q]) & t.bucket_mask;
      k0[q] = t.keys[b[q] * kBucket + sub];
    }
    uint32_t my_slot = kNoSlot;
#pragma unroll
    for (int q = 0; q < kIlpP; ++q) {
      const uint32_t sl = group_find_pre(t, key[q], b[q], k0[q], sub, gshift);
      if (sub == q) my_slot = sl;
    }
    if (i0 + sub < n) {
      cache_ids[i0 + sub] = mine;
      cache_slots[i0 + sub] = my_slot;
      missed += my_slot == kNoSlot;
      // Behind a hash filter the ids the table does not hold are what the update gates, one group
      // per distinct id: they get a sort key of their own above every slot — sur_base + the
      // position that first claimed the id's cell in a scratch hash set (4-byte cells holding a
      // position, the id itself is ids[position]) — so that the update order's sort leaves them
      // grouped by id, in input order, behind the known ids, at no extra pass.
      if (sort_key) {
        uint32_t k = my_slot;
        if (my_slot == kNoSlot) {
          const uint32_t me = (uint32_t)(i0 + sub);
          uint32_t c = (uint32_t)(mix64((uint64_t)mine ^ 0x9E3779B97F4A7C15ull) >> 24) & sur_mask;
          uint32_t leader = me;
          for (;;) {
            uint32_t v = __hip_atomic_load(&sur_cells[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v == 0xFFFFFFFFu) {
              v = atomicCAS(&sur_cells[c], 0xFFFFFFFFu, me);
              if (v == 0xFFFFFFFFu) break;  // claimed: this position leads the id's group
            }
            if (ids[v] == mine) {
              leader = v;
              break;
            }
            c = (c + 1) & sur_mask;
          }
          k = sur_base + leader;
        }
        sort_key[i0 + sub] = k;
      }
    }
  }
  // ids not found: summed per workgroup (LDS), one plain store each; k_verify adds them up
  if (missed) atomicAdd(&s_miss, missed);
  __syncthreads();
  if (threadIdx.x == 0) blk_miss[blockIdx.x] = s_miss;
}

// out[i, :] = row of slot[i] (zeros for kNoSlot). LPR lanes move one row, 16 bytes per lane and
// step; kIlpG rows per lane group in flight.
template <int LPR, int kIlpG>
__global__ void __launch_bounds__(256) k_gather_rows4(TableDev t, const uint32_t* __restrict__ slots,
                                                      int64_t n, float* __restrict__ out, int nt) {
  const int l = threadIdx.x % LPR;
  const int64_t ngroups = (int64_t)gridDim.x * (256 / LPR);
  const int D = t.dim, R = t.row_floats;
  for (int64_t i0 = ((int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR) * kIlpG; i0 < n;
       i0 += ngroups * kIlpG) {
    uint32_t sl[kIlpG];
#pragma unroll
    for (int q = 0; q < kIlpG; ++q) sl[q] = i0 + q < n ? slots[i0 + q] : kNoSlot;
    for (int e = l * 4; e < D; e += LPR * 4) {
      float4 r[kIlpG];
#pragma unroll
      for (int q = 0; q < kIlpG; ++q)
        r[q] = sl[q] != kNoSlot ? *reinterpret_cast<const float4*>(t.rows + (uint64_t)sl[q] * R + e)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < kIlpG; ++q)
        if (i0 + q < n) {
          float4* dst = reinterpret_cast<float4*>(out + (i0 + q) * D + e);
          if (nt) {  // streamed output: not kept in L2 (MONO_GATHER_NT)
            __builtin_nontemporal_store(r[q].x, &dst->x);
            __builtin_nontemporal_store(r[q].y, &dst->y);
            __builtin_nontemporal_store(r[q].z, &dst->z);
            __builtin_nontemporal_store(r[q].w, &dst->w);
          } else {
            *dst = r[q];
          }
        }
    }
  }
}
__global__ void __launch_bounds__(256) k_gather_rows1(TableDev t, const uint32_t* __restrict__ slots,
                                                      int64_t n, float* __restrict__ out) {
  const int sub = lane_sub();
  const int64_t ngroups = (int64_t)gridDim.x * (blockDim.x / 8);
  const int D = t.dim, R = t.row_floats;
  for (int64_t i = (int64_t)blockIdx.x * (blockDim.x / 8) + threadIdx.x / 8; i < n; i += ngroups) {
    const uint32_t sl = slots[i];
    for (int e = sub; e < D; e += 8) out[i * D + e] = sl != kNoSlot ? t.rows[(uint64_t)sl * R + e] : 0.f;
  }
}

// flags[1] += 1 per wave that sees a position whose id is not the one the lookup recorded;
// flags[0] = ids that lookup did not find (sum of k_probe's per-workgroup counts).
__global__ void __launch_bounds__(256) k_verify(const int64_t* __restrict__ ids,
                                                const int64_t* __restrict__ cache_ids, int64_t n,
                                                uint32_t* __restrict__ flags,
                                                uint32_t* __restrict__ handout,
                                                const uint32_t* __restrict__ blk_miss, int n_blk) {
  if (blockIdx.x == 0 && threadIdx.x == 0) handout[0] = handout[1] = handout[2] = 0;  // the long queues' cursors
  if (threadIdx.x == 0) {  // this workgroup's slice of k_probe's per-workgroup miss counts
    const int per = (n_blk + (int)gridDim.x - 1) / (int)gridDim.x;
    uint32_t mc = 0;
    for (int i = blockIdx.x * per; i < min(n_blk, ((int)blockIdx.x + 1) * per); ++i) mc += blk_miss[i];
    if (mc) atomicAdd(&flags[0], mc);  // (no atomic at all when every id was found)
  }
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    bad = bad || ids[i] != cache_ids[i];
  const unsigned long long m = __ballot(bad);
  if (m && (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(&flags[1], 1u);
}

// A lookup that consumes a hint: positions whose id is not the hinted one are resolved now and
// the record corrected (so the rows gathered next are the right ones whatever was hinted);
// flags[2] says there were some, and the hinted update order is only used when there were none.
__global__ void __launch_bounds__(256) k_hint_fix(TableDev t, const int64_t* __restrict__ ids, int64_t n,
                                                  int64_t* __restrict__ rec_ids,
                                                  uint32_t* __restrict__ rec_slots,
                                                  uint32_t* __restrict__ flags) {
  const int sub = lane_sub(), gshift = lane_gshift();
  const int lane = threadIdx.x & 63;
  for (int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane; i0 < n;
       i0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = i0 + lane;
    const bool bad = i < n && ids[i] != rec_ids[i];
    // An id the hinted probe did not find may have been inserted since (a gated update keeps the
    // queued hints, see write_path): it is probed again and the record corrected, so that the
    // lookup returns its row. The hinted order stays usable: it carries that position in its
    // tail of absent ids, which the gated update walks by key and finds in the table
    // (gated_after_prepared). Only a wrong id makes the order useless.
    const bool again = i < n && !bad && rec_slots[i] == kNoSlot;
    // Both kinds are resolved like k_probe resolves a whole batch: the wave's flagged positions are
    // compacted onto its lanes (lane L takes the L-th flagged one), and every 8-lane group then
    // probes its eight ids with all eight bucket loads in flight. A wrong hint makes EVERY
    // position a flagged one: one round per 64 positions, as many as k_probe needs (one position
    // per group and round made that lookup 0.33 ms instead of 0.09).
    const unsigned long long mbad = __ballot(bad);
    const unsigned long long ma = mbad | __ballot(again);
    uint32_t changed = 0;  // (wave-uniform)
    if (ma) {
      const int cnt = __popcll(ma);
      // src = the lane holding the lane-th flagged position: select(ma, lane) by halving
      int src = 0;
      {
        unsigned long long m = ma;
        int k = lane;
#pragma unroll
        for (int w = 32; w >= 1; w >>= 1) {
          const unsigned long long low = m & ((1ull << w) - 1ull);
          const int c = __popcll(low);
          if (k >= c) {
            k -= c;
            m >>= w;
            src += w;
          } else {
            m = low;
          }
        }
      }
      const bool valid = lane < cnt;
      if (!valid) src = 0;
      const int64_t own = i < n ? ids[i] : kEmpty;
      int64_t mine = __shfl(own, src);
      if (!valid) mine = kEmpty;
      // (a group none of whose lanes took a position has nothing to probe; group-uniform)
      const bool group_has = (lane & ~7) < cnt;
      uint32_t my_slot = kNoSlot;
      if (group_has) {
        int64_t key[8], k0[8];
        uint64_t b[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) key[q] = __shfl(mine, q, 8);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          b[q] = mix64((uint64_t)key[q]) & t.bucket_mask;
          k0[q] = t.keys[b[q] * kBucket + sub];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const uint32_t sl = group_find_pre(t, key[q], b[q], k0[q], sub, gshift);
          if (sub == q) my_slot = sl;
        }
      }
      bool chg = false;
      if (valid) {
        const bool was_bad = (mbad >> src) & 1ull;
        if (was_bad) rec_ids[i0 + src] = mine;
        if (was_bad || my_slot != kNoSlot) rec_slots[i0 + src] = my_slot;
        chg = was_bad;
      }
      changed = __ballot(chg) != 0ull ? 1u : 0u;
    }
    // (a flag, not a count: single-word atomics retire ~90 per microsecond, and a wholly wrong
    // hint would issue one per wave; the load keeps the stores to a handful per L2)
    if (lane == 0 && changed && __builtin_nontemporal_load(&flags[2]) == 0u) flags[2] = 1u;
  }
}

__global__ void __launch_bounds__(256) k_contains(TableDev t, const int64_t* __restrict__ ids,
                                                  int64_t n, uint8_t* __restrict__ out) {
  const int sub = lane_sub(), gshift = lane_gshift();
  const int64_t ngroups = (int64_t)gridDim.x * (blockDim.x / 8);
  for (int64_t i = (int64_t)blockIdx.x * (blockDim.x / 8) + threadIdx.x / 8; i < n; i += ngroups) {
    uint32_t slot = group_find(t, ids[i], sub, gshift);
    if (sub == 0) out[i] = slot != kNoSlot;
  }
}

// LookupEntry (cuckoo_embedding_hash_table.cc:162-171): the whole entry of every id that is in the
// table — weights, optimizer context, timestamp — and whether it was found.
__global__ void __launch_bounds__(256) k_lookup_entry(TableDev t, const int64_t* __restrict__ ids,
                                                      int64_t n, float* __restrict__ rows,
                                                      uint32_t* __restrict__ ts,
                                                      uint8_t* __restrict__ found) {
  const int sub = lane_sub(), gshift = lane_gshift();
  const int R = t.row_floats;
  const int64_t ngroups = (int64_t)gridDim.x * (blockDim.x / 8);
  for (int64_t i = (int64_t)blockIdx.x * (blockDim.x / 8) + threadIdx.x / 8; i < n; i += ngroups) {
    const uint32_t slot = group_find(t, ids[i], sub, gshift);
    const float* row = t.rows + (uint64_t)slot * R;
    for (int e = sub; e < R; e += 8) rows[i * R + e] = slot != kNoSlot ? row[e] : 0.f;
    if (sub == 0) {
      found[i] = slot != kNoSlot;
      ts[i] = slot != kNoSlot ? t.ts[slot] : 0u;
    }
  }
}

// Write path, step 1 (no filter): resolve every occurrence to its slot, inserting + initialising
// missing keys (cuckoo_embedding_hash_table.cc:284-291). Emits (slot, occurrence index) pairs.
__global__ void __launch_bounds__(256) k_find_or_claim(TableDev t, const int64_t* __restrict__ ids,
                                                       int64_t n, uint32_t* __restrict__ slot_out,
                                                       uint32_t* __restrict__ zero4) {
  // the apply phase's four queue counters are cleared here: a separate memset costs ~10 us of
  // stream bubble between the sort and k_classify
  if (zero4 && blockIdx.x == 0 && threadIdx.x < 16) zero4[threadIdx.x] = 0;
  const int sub = lane_sub(), gshift = lane_gshift();
  const int64_t ngroups = (int64_t)gridDim.x * (blockDim.x / 8);
  for (int64_t i0 = ((int64_t)blockIdx.x * (blockDim.x / 8) + threadIdx.x / 8) * kIlp; i0 < n;
       i0 += ngroups * kIlp) {
    // first buckets of kIlp ids in flight together; the steady state (id present, first bucket)
    // is decided from them, anything else takes the claiming probe
    int64_t key[kIlp], k0[kIlp];
    uint64_t b[kIlp];
#pragma unroll
    for (int q = 0; q < kIlp; ++q) key[q] = i0 + q < n ? ids[i0 + q] : kEmpty;
#pragma unroll
    for (int q = 0; q < kIlp; ++q) {
      b[q] = mix64((uint64_t)key[q]) & t.bucket_mask;
      k0[q] = t.keys[b[q] * kBucket + sub];
    }
#pragma unroll
    for (int q = 0; q < kIlp; ++q) {
      if (i0 + q >= n) continue;
      const bool special = key[q] == kEmpty || key[q] == kTomb;
      uint32_t m = group_bits(!special && k0[q] == key[q], gshift);
      uint32_t slot;
      if (m) {
        slot = (uint32_t)(b[q] * kBucket + (__ffs(m) - 1));
      } else {
        bool claimed;
        slot = group_find_or_claim(t, key[q], sub, gshift, &claimed);
        if (claimed) {
          group_init_row(t, key[q], slot, sub);
          if (sub == 0) t.ts[slot] = 0;
        }
        count_claimed(t, claimed && sub == 0);
      }
      if (sub == 0) slot_out[i0 + q] = slot;
    }
  }
}

// Write path, step 1 when a lookup of the same positions came before: a wave checks 64
// positions with coalesced loads (id now, id and slot then); a position whose id is unchanged and
// was found takes its slot from the record, the others are probed, eight at a time, one per
// sub-group. In the steady state this kernel is a 34 MB streaming read.
__global__ void __launch_bounds__(256) k_claim_pending(TableDev t, const int64_t* __restrict__ ids,
                                                       int64_t n, uint32_t* slot_out,
                                                       uint32_t* __restrict__ zero4,
                                                       const int64_t* __restrict__ cache_ids,
                                                       const uint32_t* __restrict__ cache_slots,
                                                       int64_t cache_n) {
  if (zero4 && blockIdx.x == 0 && threadIdx.x < 16) zero4[threadIdx.x] = 0;
  const int lane = threadIdx.x & 63, sub = lane_sub(), gshift = lane_gshift(), grp = lane >> 3;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / 64);
  for (int64_t base = ((int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 64; base < n;
       base += nwaves * 64) {
    bool pend = false;
    if (base + lane < n) {
      const int64_t i = base + lane;
      uint32_t sv = kNoSlot;
      if (i < cache_n && cache_ids[i] == ids[i]) sv = cache_slots[i];
      if (sv != kNoSlot) slot_out[i] = sv;
      else pend = true;
    }
    unsigned long long m = __ballot(pend);
    while (m) {  // wave-uniform: up to eight pending positions per round
      // sub-group g takes the g-th set bit
      unsigned long long rest = m;
      int mine = -1;
      for (int g = 0; g < 8; ++g) {
        if (!rest) break;
        const int b = __ffsll((long long)rest) - 1;
        if (g == grp) mine = b;
        rest &= rest - 1;
      }
      m = rest;
      if (mine >= 0) {
        const int64_t i = base + mine;
        const int64_t key = ids[i];
        bool claimed;
        const uint32_t slot = group_find_or_claim(t, key, sub, gshift, &claimed);
        if (claimed) {
          group_init_row(t, key, slot, sub);
          if (sub == 0) t.ts[slot] = 0;
        }
        count_claimed(t, claimed && sub == 0);
        if (sub == 0) slot_out[i] = slot;
      }
    }
  }
}

// Write path with a hash filter, step 1: which positions name ids the table already holds? Those
// never touch the filter (bridge :274-275, :288-289) and take the plain path by slot; the others
// are marked (slot = `skip`, unknown = 1) for the gated path. Nothing is claimed here. Uses the
// lookup's slot record like k_claim_pending.
__global__ void __launch_bounds__(256) k_resolve_known(TableDev t, const int64_t* __restrict__ ids,
                                                       int64_t n, uint32_t* slot_out,
                                                       int32_t* unknown, uint32_t skip,
                                                       const int64_t* __restrict__ cache_ids,
                                                       const uint32_t* __restrict__ cache_slots,
                                                       int64_t cache_n) {
  const int lane = threadIdx.x & 63, sub = lane_sub(), gshift = lane_gshift(), grp = lane >> 3;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / 64);
  for (int64_t base = ((int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 64; base < n;
       base += nwaves * 64) {
    bool pend = false;
    if (base + lane < n) {
      const int64_t i = base + lane;
      uint32_t sv = kNoSlot;
      if (i < cache_n && cache_ids[i] == ids[i]) sv = cache_slots[i];
      if (sv != kNoSlot) {
        slot_out[i] = sv;
        unknown[i] = 0;
      } else {
        pend = true;
      }
    }
    unsigned long long m = __ballot(pend);
    while (m) {  // wave-uniform: up to eight pending positions per round
      unsigned long long rest = m;
      int mine = -1;
      for (int g = 0; g < 8; ++g) {
        if (!rest) break;
        const int b = __ffsll((long long)rest) - 1;
        if (g == grp) mine = b;
        rest &= rest - 1;
      }
      m = rest;
      if (mine >= 0) {
        const int64_t i = base + mine;
        const uint32_t slot = group_find(t, ids[i], sub, gshift);
        if (sub == 0) {
          slot_out[i] = slot == kNoSlot ? skip : slot;
          unknown[i] = slot == kNoSlot;
        }
      }
    }
  }
}

// ... step 2: the unknown positions, compacted in input order by two kernels over the same
// contiguous chunks (one workgroup each): k_unk_count leaves a count per chunk, k_unk_emit sums
// the counts before its chunk and writes its chunk's unknown positions in order (ballot ranks
// inside a wave, a four-entry LDS scan across waves). `flag` is k_resolve_known's mark per
// position.
constexpr int kUnkMaxChunks = 256;
struct UnkGeom {
  int chunks;
  int64_t chunk;  // positions per chunk, a multiple of 256
};
static UnkGeom unk_geom(int64_t n) {
  UnkGeom g;
  g.chunks = (int)std::min<int64_t>(kUnkMaxChunks, std::max<int64_t>(1, (n + 1023) / 1024));
  g.chunk = (((n + g.chunks - 1) / g.chunks) + 255) & ~(int64_t)255;
  return g;
}
__global__ void __launch_bounds__(256) k_unk_count(const uint32_t* __restrict__ flag, int64_t n,
                                                   int64_t chunk, int32_t* __restrict__ cnt) {
  __shared__ int s_w[4];
  const int64_t lo = (int64_t)blockIdx.x * chunk, hi = std::min<int64_t>(n, lo + chunk);
  int c = 0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) c += flag[i] != 0u ? 1 : 0;
  for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ void __launch_bounds__(256) k_unk_emit(const int64_t* __restrict__ ids,
                                                  const uint32_t* __restrict__ flag, int64_t n,
                                                  int64_t chunk, const int32_t* __restrict__ cnt,
                                                  int64_t* __restrict__ unk_ids,
                                                  uint32_t* __restrict__ unk_idx, int64_t cap) {
  __shared__ int s_w[4];
  __shared__ int s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int b = (int)threadIdx.x < (int)blockIdx.x ? cnt[threadIdx.x] : 0;
  for (int o = 32; o; o >>= 1) b += __shfl_xor(b, o);
  if (lane == 0) s_w[wave] = b;
  __syncthreads();
  if (threadIdx.x == 0) s_base = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  __syncthreads();
  int64_t base = s_base;
  const int64_t lo = (int64_t)blockIdx.x * chunk, hi = std::min<int64_t>(n, lo + chunk);
  for (int64_t t0 = lo; t0 < hi; t0 += 256) {
    const int64_t i = t0 + threadIdx.x;
    const bool f = i < hi && flag[i] != 0u;
    const unsigned long long m = __ballot(f);
    __syncthreads();  // (the previous tile's s_w has been read)
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      before += w < wave ? s_w[w] : 0;
      all += s_w[w];
    }
    if (f) {
      const int64_t o = base + before + __popcll(m & ((1ull << lane) - 1ull));
      if (o < cap) {  // (cap = the count the host sized the arrays for; always holds)
        unk_ids[o] = ids[i];
        unk_idx[o] = (uint32_t)i;
      }
    }
    base += all;
  }
}

// Behind a prepared order the same list needs no compaction: the order is a stable sort by slot
// with kNoSlot (all ones in the sorted bits, above every slot) for the ids the lookup did not find,
// so its last U entries ARE the unknown positions in input order.
__global__ void __launch_bounds__(256) k_unk_from_order(const int64_t* __restrict__ ids,
                                                        const uint32_t* __restrict__ order_tail,
                                                        int64_t U, int64_t n,
                                                        int64_t* __restrict__ unk_ids,
                                                        uint32_t* __restrict__ unk_idx) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < U;
       k += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t i = order_tail[k];
    unk_idx[k] = i;
    unk_ids[k] = (int64_t)i < n ? ids[i] : 0;
  }
}

__global__ void k_iota(uint32_t* idx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    idx[i] = (uint32_t)i;
}

struct LongSeg;
struct ApplyArgs {
  const void* skeys;     // sorted uint32 slots, or sorted int64 keys when from_keys
  const uint32_t* sidx;  // occurrence index of each sorted position
  int64_t n;
  const float* src;  // grads / values, row-major [*, dim] (MODE_IMPORT: [*, row_floats])
  const uint32_t* src_ts;  // MODE_IMPORT only
  int mode;
  int from_keys;
  uint32_t ts;
  float lr[kMaxSeg];
  FilterDev filt;  // from_keys only; filt.kind == 0 -> no gate
  int assign_add_semantics;  // filter gate of the per-id AssignAdd loop (bridge :186-209)
  long long global_step;     // Optimize()'s global_step (batch softmax reads it)
  struct LongSeg* queue;     // segments longer than kLongSeg, drained by k_apply_long
  uint32_t* qcount;          // [0] queue entries [1] compacted heads [2] split keys [3] partial rows
                             // [4] next queue entry to hand out (k_apply_long / k_long_chain)
                             // [5] next very long entry to hand out [6] chain workgroups that
                             // have started (k_chain_gate) [7] packed blocks
                             // [8 .. 13] very long entries per length class
  struct SplitSeg* cqueue;   // keys whose segment was cut into chunks, finished by k_combine
  float* partials;           // [partial row][dim] chunk sums of the split keys
  int chunk;                 // occurrences above which a key is split (chunk_size(): 1 << 30 = never)
  const uint4* heads;        // compacted (position, length, slot, first index) of the short
                             // segments, or nullptr
  const uint32_t* nheads;
  const uint32_t* sort_err;  // sort.hip's sticky error word: non-zero = the sorted arrays are not
                             // to be trusted, the batch is skipped (k_classify emits nothing)
  // stateless segment ops (layout gradient): source row of occurrence i = src_ptrs[i] (dim
  // floats) instead of src + i * dim; k_classify keeps only the segments whose occurrences belong
  // to a class of width want_dim (class of occurrence i = cls[i], width = cls_dim[class])
  const float* const* src_ptrs;
  int src_aligned4;
  const uint32_t* cls;
  const int32_t* cls_dim;
  int want_dim;      // 0 = every segment
  uint32_t skip_slot;  // sorted key that marks "no destination" (besides kNoSlot)
  uint32_t sur_base;   // != 0: sorted keys >= sur_base stand for ids the table did not hold (k_probe)
  // packed sources (segment_set_sum_packed): occurrence value = class | sample << pk_sbits |
  // seq << pk_qshift; its row = src_tab[class].base + sample * stride + seq * seq_stride, its class
  // (k_classify's width filter) = value & pk_smask
  const SrcTab* src_tab;
  uint32_t pk_smask, pk_bmask;
  int pk_sbits, pk_qshift;
  int64_t max_heads;  // the caller's bound on the number of distinct keys (0: n), sizes k_apply's grid
  // hot-key chains (k_long_pack / k_long_chain): the rows of every long segment are first copied
  // into `packed` in sorted order, 64-row blocks, four-row groups column-major (chain_dev.h);
  // blk_src[b] = (sorted position of block b's first row, rows in it). qcount[6] counts blocks.
  uint2* blk_src;
  uint32_t blk_cap;  // entries of blk_src
  // queue indices of the segments longer than kVeryLong by length class (longest class first,
  // vl_cap entries each; counts in qcount[8 + class]): each gets a CU to itself
  uint32_t* vlq;
  uint32_t vl_cap;
  float* packed;
  uint32_t packed_bytes;
  // apply kernels launched before the host has k_verify's verdict (try_prepared): they do nothing
  // unless guard[0] == 0 (no id missing), guard[1] == guard_seen (no id changed) and, for a hinted
  // order, guard[2] == 0 (the lookup brought the hinted ids)
  const uint32_t* guard;
  uint32_t guard_seen;
  int guard_hinted;
};
__device__ __forceinline__ bool guard_ok(const ApplyArgs& a) {
  if (!a.guard) return true;
  return a.guard[0] == 0u && a.guard[1] == a.guard_seen && (!a.guard_hinted || a.guard[2] == 0u);
}
// PTRS: can this instantiation be given per-occurrence source pointers at all? (Only the set-sum
// mode is; a run-time test in every other kernel's gradient loads cost the headline k_apply a
// quarter of its speed.)
template <bool PTRS>
__device__ __forceinline__ const float* src_row(const ApplyArgs& a, uint32_t idx, int D) {
  if (PTRS && a.src_tab) {
    const SrcTab s = a.src_tab[idx & a.pk_smask];
    return s.base + (int64_t)((idx >> a.pk_sbits) & a.pk_bmask) * s.stride +
           (int64_t)(idx >> a.pk_qshift) * s.seq_stride;
  }
  if (PTRS && a.src_ptrs) return a.src_ptrs[idx];
  return a.src + (uint64_t)idx * D;
}

// A key with many occurrences in one batch (a Zipf head): handed to a whole workgroup.
struct LongSeg {
  int64_t first, end, key;
  uint32_t slot, fresh;
  uint32_t part;  // partial-row index when this entry is one chunk of a split key, else kNoPart
  uint32_t blk;   // first 64-row block of this entry's packed rows (k_long_pack -> k_long_chain)
};
struct SplitSeg {
  int64_t key;
  uint32_t slot, fresh, pbase, nparts;
};
constexpr uint32_t kNoPart = 0xFFFFFFFFu;
#ifndef MONO_LONG_SEG
#define MONO_LONG_SEG 96
#endif
constexpr int kLongSeg = MONO_LONG_SEG;       // occurrences above which a segment goes to k_apply_long
// Only under MONO_CHUNK=<rows> (chunk_size(); off by default): a key with more occurrences than
// that in one batch is summed in near-equal chunks (each in input order), k_combine then adds the
// chunk sums in chunk order — the one form whose summation order differs from the sequential
// reference. MODE_OPT_SEQ is never split.
constexpr int kLongMaxDim = 512;   // wider rows stay on the sub-group path
constexpr int kPackRows = 64;      // rows per packed block = one loop iteration of chain_sum
constexpr int kVeryLong = 2048;    // occurrences above which a chain gets a CU to itself
constexpr int kVlClasses = 6;      // ... handed out longest class first
__host__ __device__ inline int vl_class(int64_t len) {
  return len > 65536 ? 0 : len > 32768 ? 1 : len > 16384 ? 2 : len > 8192 ? 3 : len > 4096 ? 4 : 5;
}
constexpr int kTileFloats = 2048;  // LDS staging tile (8 KB), double buffered

template <typename K>
__device__ inline int64_t segment_end(const K* keys, int64_t j, int64_t n, K key) {
  // galloping upper bound: first position > j whose key differs
  int64_t step = 1, lo = j, hi = j + 1;
  while (hi < n && keys[hi] == key) {
    lo = hi;
    step <<= 1;
    hi = j + step;
  }
  if (hi > n) hi = n;
  // invariant: keys[lo] == key, (hi == n or keys[hi] != key)
  while (hi - lo > 1) {
    int64_t mid = lo + ((hi - lo) >> 1);
    if (keys[mid] == key) lo = mid; else hi = mid;
  }
  return hi;
}

// One optimizer step on one element. Non-contracted arithmetic
// (-ffp-contract=off) with correctly rounded sqrt/div, i.e. the reference's baseline form.
// Serving entries (entry_accessor.cc:43-98) keep what the segment's compressor returns for a
// value: Decode(Encode(v)) of float_compressor.cc:63-137, stored as fp32 (the rows are not
// narrowed; a table too large for that is a training table restored shard by shard).
__device__ __forceinline__ float comp_q(const SegDev& c, float v) {
  if (c.comp_kind == MONO_COMP_FP16) {  // half_float::half(v): to nearest, ties to even
    return __half2float(__float2half_rn(v));
  } else if (c.comp_kind == MONO_COMP_FIXED_R8) {  // fake_quantizer.h:35-60
    if (isnan(v)) return 0.f * c.cq0;
    const float f = v >= 0 ? v + c.cq1 : v - c.cq1;
    const float q = __fdiv_rn(f, c.cq0);
    // int nstep = f / step_: out of int range the reference's x86 conversion yields INT_MIN
    int nstep = (fabsf(q) < 2147483648.f) ? (int)q : (int)0x80000000;
    nstep = nstep > 127 ? 127 : (nstep < -128 ? -128 : nstep);
    return (float)nstep * c.cq0;
  } else if (c.comp_kind == MONO_COMP_ONE_BIT) {  // hash_net_quantizer.h:42-44, :120-131
    const float y = c.cq0 * tanhf(c.cq1 * v);
    return (y > 0 ? 1.f : -1.f) * c.cq0;
  }
  return v;
}

// stochastic_rounding.h:27-37: vf goes to the binary16 neighbour above with probability
// (vf - down) / (up - down). The neighbours come from the to-nearest conversion: that result is
// one of them, the other is the next half on the far side (= float2half<round_toward_infinity /
// round_toward_neg_infinity> of the reference's half.hpp; the oracle's restatement is checked
// against that header, this one against the oracle).
__device__ __forceinline__ float half_neighbour(float v, bool up) {
  if (isnan(v)) return v;
  const __half hn = __float2half_rn(v);
  const float near = __half2float(hn);
  if (near == v) return near;
  if (up ? near > v : near < v) return near;
  const unsigned short h = __half_as_ushort(hn);
  const bool neg = (h & 0x8000u) != 0;
  unsigned short o;
  if (up) o = neg ? ((h & 0x7FFFu) ? (unsigned short)(h - 1) : (unsigned short)0x0001u) : (unsigned short)(h + 1);
  else o = neg ? (unsigned short)(h + 1) : ((h & 0x7FFFu) ? (unsigned short)(h - 1) : (unsigned short)0x8001u);
  return __half2float(__ushort_as_half(o));
}
__device__ __forceinline__ float stochastic_round(float vf, float p) {
  const float up = half_neighbour(vf, true), down = half_neighbour(vf, false);
  return (p <= __fdiv_rn(vf - down, up - down)) ? up : down;
}
// p: counter-based (table seed, global_step, batch position of the occurrence, element, stage);
// the reference's thread-local sequence cannot be pinned (see oracle.cc, sr_uniform)
__device__ __forceinline__ float sr_uniform(uint64_t seed, long long gstep, uint32_t pos, int e,
                                            int stage) {
  uint64_t x = mix64(seed ^ 0x5bd1e995u) ^ mix64((uint64_t)gstep + 0x9e3779b97f4a7c15ULL);
  x = mix64(x ^ ((uint64_t)pos << 32 | (uint32_t)e)) ^ mix64((uint64_t)stage + 77);
  const uint32_t u = (uint32_t)(mix64(x) >> 32);
  return __fdiv_rn((float)u, 4294967296.f);
}

// third state slot + per-row scalars (generic path only; the specialised kernels pass a dummy)
struct OptX {
  float s2 = 0.f, b1p = 0.f, b2p = 0.f;
  long long gstep = 0;  // the call's global_step (batch softmax)
};

__device__ __forceinline__ void opt_step(int kind, const SegDev& c, float lr, float g, float& w,
                                         float& s0, float& s1, OptX& x) {
  if (kind == MONO_OPT_SGD) {  // sgd_optimizer.cc:36-44
    w -= lr * g;
  } else if (kind == MONO_OPT_ADAGRAD) {  // avx_utils.h:29-38
    float gg = g + c.wd * w;
    s0 += gg * gg;
    float eff = __fdiv_rn(lr, sqrtf(s0));
    w -= eff * gg;
  } else if (kind == MONO_OPT_FTRL) {  // ftrl_optimizer.cc:49-72; s0 = norm, s1 = zero
    float norm_new = s0 + g * g;
    float sigma = __fdiv_rn(sqrtf(norm_new) - sqrtf(s0), lr);
    s1 += (g - sigma * w);
    s0 = norm_new;
    float sign = (float)(__float_as_uint(s1) >> 31);  // std::signbit as 0/1
    w = (fabsf(s1) > c.l1)
            ? __fdiv_rn(lr * (sign * c.l1 - s1), (sqrtf(s0) + c.beta + c.l2 * lr))
            : 0.0f;
  } else if (kind == MONO_OPT_MOMENTUM) {  // momentum_optimizer.cc:49-72; s0 = n
    const float dx = lr * (g + c.wd * w);
    const float new_n = c.momentum * s0 - dx;
    if (c.nesterov) w += -c.momentum * s0 + (1 + c.momentum) * new_n;
    else w += new_n;
    s0 = new_n;
  } else if (kind == MONO_OPT_RMSPROP || kind == MONO_OPT_RMSPROP_V2) {
    // rmsprop_optimizer.cc:46-63 / :119-137: the update is carried in double; V1 divides the
    // CONFIG learning rate and weights dx^2 by (1 - momentum), V2 uses the call's rate
    const double dx = (double)g + (double)c.wd * (double)w;
    const double mom = (double)c.momentum;
    const float new_n = kind == MONO_OPT_RMSPROP ? (float)(mom * (double)s0 + (1 - mom) * dx * dx)
                                                 : (float)(mom * (double)s0 + dx * dx);
    const double rate = kind == MONO_OPT_RMSPROP ? (double)c.conf_lr : (double)lr;
    const double eta = rate / (double)(sqrtf(new_n) + 1);
    w = (float)((double)w - eta * dx);
    s0 = new_n;
  } else if (kind == MONO_OPT_ADAM || kind == MONO_OPT_AMSGRAD) {
    // adam_optimizer.cc:53-80 / amsgrad_optimizer.cc:55-86; s0 = m, s1 = v, x.s2 = vhat. The
    // unqualified sqrt() of the reference resolves to ::sqrt(double), so the rate and the
    // quotient are carried in double and rounded once.
    const float rate = (float)((double)lr * sqrt((double)(1 - x.b2p)) / (double)(1 - x.b1p));
    const float cur = g + c.wd * w;
    const float new_m = s0 + (cur - s0) * (1 - c.beta1);
    const float new_v = s1 + (cur * cur - s1) * (1 - c.beta2);
    float den_v = new_v;
    if (kind == MONO_OPT_AMSGRAD) {
      x.s2 = fmaxf(x.s2, new_v);
      den_v = x.s2;
    }
    const float num = c.nesterov ? (cur * (1 - c.beta1) + c.beta1 * new_m) * rate : new_m * rate;
    w = (float)((double)w - (double)num / (sqrt((double)den_v) + (double)c.epsilon));
    s0 = new_m;
    s1 = new_v;
    x.b1p *= c.beta1;  // every lane keeps its own copy: one multiplication per Optimize call
    x.b2p *= c.beta2;
  } else if (kind == MONO_OPT_BATCH_SOFTMAX) {  // batch_softmax_optimizer.cc:42-54; g unused
    const long long A = (long long)(((unsigned long long)__float_as_uint(s1) << 32) |
                                    (unsigned long long)__float_as_uint(s0));
    w = (1 - lr) * w + lr * (float)(x.gstep - A);
    s0 = __uint_as_float((uint32_t)((unsigned long long)x.gstep & 0xFFFFFFFFull));
    s1 = __uint_as_float((uint32_t)((unsigned long long)x.gstep >> 32));
  } else if (kind == MONO_OPT_MOVING_AVERAGE) {  // moving_average_optimizer.cc:43-46
    w = c.momentum * w + (1 - c.momentum) * g;
  } else {  // adadelta_optimizer.cc:50-66; s0 = accum, s1 = accum_update
    const float cur = g + c.wd * w;
    const float new_accum = s0 * c.rho + cur * cur * (1 - c.rho);
    const float update = __fdiv_rn(sqrtf(s1 + c.epsilon), sqrtf(new_accum + c.epsilon)) * cur;
    w = w - update * lr;
    s1 = s1 * c.rho + update * update * (1 - c.rho);
    s0 = new_accum;
  }
}

// generic path: third state + row scalars of the segment that owns element e
__device__ __forceinline__ void optx_load(const TableDev& t, const SegDev& sg, const float* row,
                                          int e, bool fresh, int64_t key, OptX& x) {
  const int ns = opt_nstate(sg.opt_kind);
  if (ns >= 3) {
    const int off = sg.st_off + 2 * sg.dim + (e - sg.w_off);
    x.s2 = fresh ? row_init_value(t, key, off) : row[off];
  }
  if (opt_nscalar(sg.opt_kind)) {
    const int off = sg.st_off + ns * sg.dim;
    x.b1p = fresh ? row_init_value(t, key, off) : row[off];
    x.b2p = fresh ? row_init_value(t, key, off + 1) : row[off + 1];
  }
}
__device__ __forceinline__ void optx_store(const SegDev& sg, float* row, int e, const OptX& x) {
  const int ns = opt_nstate(sg.opt_kind);
  if (ns >= 3) row[sg.st_off + 2 * sg.dim + (e - sg.w_off)] = x.s2;
  // The row scalars are written by the lane of the segment's LAST weight: rows wider than one
  // pass (8 lanes x V, or 64 columns) are walked in ascending element order, and every pass
  // loads before it stores, so no lane can still need the old value.
  if (opt_nscalar(sg.opt_kind) && e == sg.w_off + sg.dim - 1) {
    row[sg.st_off + ns * sg.dim] = x.b1p;
    row[sg.st_off + ns * sg.dim + 1] = x.b2p;
  }
}

__device__ __forceinline__ bool long_split(const ApplyArgs& a, int64_t len) {
  return a.mode != MODE_OPT_SEQ && len > a.chunk;
}
__device__ __forceinline__ uint32_t long_parts(const ApplyArgs& a, int64_t len) {
  return long_split(a, len) ? (uint32_t)((len + a.chunk - 1) / a.chunk) : 1u;
}
// packed 64-row blocks of a long segment (every part of a split key is padded on its own)
__device__ __forceinline__ uint32_t long_blocks(const ApplyArgs& a, int64_t len) {
  const uint32_t nparts = long_parts(a, len);
  const int64_t csz = (len + nparts - 1) / nparts;
  const int64_t last = len - (int64_t)(nparts - 1) * csz;
  return (uint32_t)((nparts - 1) * ((csz + kPackRows - 1) / kPackRows) + (last + kPackRows - 1) / kPackRows);
}
// queue entries q .. q + nparts; a split key also takes partial rows pbase .. + nparts and the
// split-key entry c; packed blocks blk .. + long_blocks (all four reserved by the caller)
// `wg_list` (LDS, kClsBlkList entries, `wg_n` its counter): the caller's workgroup writes the
// block descriptors of the listed parts together afterwards (write_block_list) — one lane doing
// it for a 30 000-row segment is a 470-iteration loop that the whole workgroup waits for.
constexpr int kClsBlkList = 64;
__device__ inline void enqueue_long_at(const ApplyArgs& a, int64_t first, int64_t end, int64_t key,
                                       uint32_t slot, uint32_t fresh, uint32_t q, uint32_t pbase,
                                       uint32_t c, uint32_t blk, uint4* wg_list = nullptr,
                                       uint32_t* wg_n = nullptr) {
  const int64_t len = end - first;
  const bool split = long_split(a, len);
  const uint32_t nparts = long_parts(a, len);
  const int64_t csz = (len + nparts - 1) / nparts;
  if (split) {
    SplitSeg sp;
    sp.key = key; sp.slot = slot; sp.fresh = fresh; sp.pbase = pbase; sp.nparts = nparts;
    a.cqueue[c] = sp;
  }
  for (uint32_t p = 0; p < nparts; ++p) {
    LongSeg ls;
    ls.first = first + (int64_t)p * csz;
    ls.end = min(end, ls.first + csz);
    ls.key = key; ls.slot = slot; ls.fresh = fresh;
    ls.part = split ? pbase + p : kNoPart;
    ls.blk = blk;
    a.queue[q + p] = ls;
    const int64_t plen = ls.end - ls.first;
    if (a.blk_src) {
      const uint32_t li = wg_list && plen > 4 * kPackRows ? atomicAdd(wg_n, 1u) : (uint32_t)kClsBlkList;
      if (li < (uint32_t)kClsBlkList) {
        wg_list[li] = make_uint4((uint32_t)ls.first, (uint32_t)plen, blk, 0u);
        blk += (uint32_t)((plen + kPackRows - 1) / kPackRows);
      } else {
        for (int64_t r = 0; r < plen; r += kPackRows, ++blk)
          if (blk < a.blk_cap)  // (always: the host sizes blk_src for the worst case)
            a.blk_src[blk] = make_uint2((uint32_t)(ls.first + r), (uint32_t)min((int64_t)kPackRows, plen - r));
      }
      if (plen > kVeryLong) {  // (rare: a direct global atomic per entry)
        const int cl = vl_class(plen);
        const uint32_t i = atomicAdd(&a.qcount[8 + cl], 1u);
        if (i < a.vl_cap) a.vlq[(uint32_t)cl * a.vl_cap + i] = q + p;
      }
    }
  }
}
// (all threads of the workgroup, after a barrier behind the enqueue_long_at calls)
__device__ inline void write_block_list(const ApplyArgs& a, const uint4* wg_list, uint32_t n_list) {
  n_list = min(n_list, (uint32_t)kClsBlkList);
  for (uint32_t e = 0; e < n_list; ++e) {
    const uint4 d = wg_list[e];  // (first sorted position, rows, first block)
    const uint32_t nb = (d.y + kPackRows - 1) / kPackRows;
    for (uint32_t r = threadIdx.x; r < nb; r += blockDim.x)
      if (d.z + r < a.blk_cap)
        a.blk_src[d.z + r] = make_uint2(d.x + r * kPackRows, min((uint32_t)kPackRows, d.y - r * kPackRows));
  }
}
__device__ inline void enqueue_long(const ApplyArgs& a, int64_t first, int64_t end, int64_t key,
                                    uint32_t slot, uint32_t fresh) {
  const int64_t len = end - first;
  const bool split = long_split(a, len);
  const uint32_t nparts = long_parts(a, len);
  const uint32_t q = atomicAdd(&a.qcount[0], nparts);
  uint32_t pbase = kNoPart, c = 0;
  if (split) {
    pbase = atomicAdd(&a.qcount[3], nparts);
    c = atomicAdd(&a.qcount[2], 1u);
  }
  const uint32_t blk = a.blk_src ? atomicAdd(&a.qcount[7], long_blocks(a, len)) : 0u;
  enqueue_long_at(a, first, end, key, slot, fresh, q, pbase, c, blk);
}

// Write path, step 2: one 8-lane sub-group per distinct key walks that key's occurrences in
// input order (the sort is stable) with the row chunk held in registers, then stores row + ts.
// B3/B4: bridge :156-313, cuckoo_embedding_hash_table.cc:174-216, entry_accessor.cc:149-171.
// 8 waves / SIMD (64 VGPRs, a 36-byte spill): the kernel is bound by memory latency x resident
// sub-groups, and the extra wave is worth more than the spill costs (7.83 -> 8.01 G).
// MODE_T >= 0 fixes the mode at compile time (the dedup instantiation of the headline step then
// carries none of the other walks: fewer live registers at 8 waves / SIMD); -1 reads mode.
template <int V, int OPTK, int MODE_T = -1>
__global__ void __launch_bounds__(256, 8) k_apply(TableDev t, ApplyArgs a) {
  const int mode = MODE_T >= 0 ? MODE_T : a.mode;
  constexpr bool kPtrs = MODE_T < 0 || MODE_T == MODE_SET_SUM;
  // the fixed-mode instantiation is launched on the no-filter path only: slots are resolved and
  // k_classify has compacted the segment heads
  const bool from_keys = MODE_T >= 0 ? false : (a.from_keys != 0);
  const bool has_heads = MODE_T >= 0 ? true : (a.heads != nullptr);
  const int sub = lane_sub(), gshift = lane_gshift();
  int64_t j = (int64_t)blockIdx.x * (blockDim.x / 8) + threadIdx.x / 8;
  if (j >= a.n) return;
  if (!guard_ok(a)) return;
  const int D = t.dim, R = t.row_floats;
  uint4 hd = make_uint4(0u, 0u, 0u, 0u);
  if (has_heads) {  // compacted segment heads (k_classify): every sub-group has work
    if (j >= (int64_t)*a.nheads) return;
    hd = a.heads[j];
    j = hd.x;
  }

  uint32_t slot;
  int64_t end;
  int64_t first = j;  // first occurrence (sorted position) that is applied
  bool fresh = false; // row claimed here and not yet initialised in memory
  int64_t key = 0;
  uint32_t idx0 = 0;  // occurrence index of sorted position j
  if (!from_keys) {
    const uint32_t* sk = static_cast<const uint32_t*>(a.skeys);
    // four independent loads up front: the common single-occurrence segment then needs only
    // two dependent memory levels (sorted arrays -> row + gradient)
    if (has_heads) {  // slot and first index ride in the head record: one memory level less
      slot = hd.z;
      idx0 = hd.w;
      end = j + hd.y;
    } else {
      slot = sk[j];
      idx0 = a.sidx[j];
      const uint32_t prev = j > 0 ? sk[j - 1] : ~slot;
      const uint32_t next = j + 1 < a.n ? sk[j + 1] : ~slot;
      if (prev == slot) return;
      if (slot == kNoSlot || (a.sur_base != 0u && slot >= a.sur_base)) return;
      end = next != slot ? j + 1 : segment_end<uint32_t>(sk, j, a.n, slot);
    }
  } else {
    const int64_t* sk = static_cast<const int64_t*>(a.skeys);
    key = sk[j];
    idx0 = a.sidx[j];
    if (j > 0 && sk[j - 1] == key) return;
    end = segment_end<int64_t>(sk, j, a.n, key);
    slot = group_find(t, key, sub, gshift);
    if (slot == kNoSlot) {
      // Unknown id: the hash-filter gate (bridge :167-176, :195-198, :274-277, :288-291).
      int64_t k = end - j;
      int64_t skip = 0;
      if (a.filt.kind != 0) {
        int thr = filter_threshold(a.filt, key);
        if (thr > 0) {
          uint32_t c = 0;
          if (mode == MODE_OPT_DEDUP) {
            // one add(fid, k); admitted iff old count >= threshold
            if (sub == 0) c = filter_add(a.filt, (uint64_t)key, (uint32_t)min((int64_t)15, k));
            c = __shfl(c, 0, 8);
            skip = c >= (uint32_t)thr ? 0 : k;
          } else {
            // occurrence q sees old count min(15, c + q): first admitted q* = max(0, thr - c).
            // AssignAdd stops counting once the id is inserted; Assign / Optimize count all k.
            int64_t q;
            if (sub == 0) {
              c = filter_get(a.filt, (uint64_t)key);
              q = (int64_t)thr - (int64_t)c;
              if (q < 0) q = 0;
              if (thr > 15) q = k;  // counts saturate at 15: never admitted
              int64_t adds = a.assign_add_semantics ? (q < k ? q + 1 : k) : k;
              if (adds > 0) filter_add(a.filt, (uint64_t)key, (uint32_t)min((int64_t)15, adds));
              c = (uint32_t)(q < k ? q : k);
            }
            c = __shfl(c, 0, 8);
            skip = c;
          }
        }
      }
      if (skip >= k) return;
      first = j + skip;
      bool claimed;
      slot = group_find_or_claim(t, key, sub, gshift, &claimed);
      fresh = claimed;
      count_claimed(t, claimed && sub == 0);
    }
  }

  if (end - first > kLongSeg && D <= kLongMaxDim && !t.serving && !t.any_sr &&
      (mode == MODE_ASSIGN_ADD || mode == MODE_OPT_SEQ || mode == MODE_OPT_DEDUP ||
       mode == MODE_SET_SUM)) {
    if (sub == 0) enqueue_long(a, first, end, key, slot, fresh ? 1u : 0u);
    return;
  }

  float* row = t.rows + (uint64_t)slot * R;
  if (mode == MODE_IMPORT) {  // full-row restore: last occurrence wins
    uint32_t src_i = a.sidx[end - 1];
    if (t.serving) {  // Restore = Encode(dump.num) (entry_accessor.cc:89-94); R == D here
      for (int e = sub; e < R; e += 8) {
        int sgi = 0;
        for (int z = 0; z < t.n_seg; ++z)
          if (e >= t.seg[z].w_off) sgi = z;
        row[e] = comp_q(t.seg[sgi], a.src[(uint64_t)src_i * R + e]);
      }
    } else {
      for (int e = sub; e < R; e += 8) row[e] = a.src[(uint64_t)src_i * R + e];
    }
    if (sub == 0) t.ts[slot] = a.src_ts[src_i];
    return;
  }

  for (int c0 = 0; c0 < D; c0 += 8 * V) {
    const int e0 = c0 + sub * V;
    if (e0 >= D) continue;
    float w[V], s0[V], s1[V];
    OptX ox[V];
#pragma unroll
    for (int q = 0; q < V; ++q) ox[q].gstep = a.global_step;
    int kind[V];
    int segi[V];
    int st0_off[V], st1_off[V];
    // --- locate state + load
#pragma unroll
    for (int q = 0; q < V; ++q) {
      if (OPTK == OPTK_MULTI) {
        int e = e0 + q, s = 0;
        for (int z = 0; z < t.n_seg; ++z)
          if (e >= t.seg[z].w_off && e < t.seg[z].w_off + t.seg[z].dim) s = z;
        segi[q] = s;
        kind[q] = t.seg[s].opt_kind;
        st0_off[q] = t.seg[s].st_off + (e - t.seg[s].w_off);
        st1_off[q] = st0_off[q] + t.seg[s].dim;
      } else {
        segi[q] = 0;
        kind[q] = OPTK;
        st0_off[q] = D + e0 + q;
        st1_off[q] = 2 * D + e0 + q;
      }
    }
    if (OPTK == OPTK_MULTI && t.any_sr) {  // the rounding decorator calls Optimize without the step
#pragma unroll
      for (int q = 0; q < V; ++q)
        if (t.seg[segi[q]].sr_post) ox[q].gstep = 0;
    }
    const bool need_state = mode == MODE_OPT_SEQ || mode == MODE_OPT_DEDUP;
    if (mode == MODE_SET_SUM) {
#pragma unroll
      for (int q = 0; q < V; ++q) w[q] = s0[q] = s1[q] = 0.f;
    } else if (fresh) {
#pragma unroll
      for (int q = 0; q < V; ++q) {
        w[q] = row_init_value(t, key, e0 + q);
        s0[q] = opt_nstate(kind[q]) >= 1 ? row_init_value(t, key, st0_off[q]) : 0.f;
        s1[q] = opt_nstate(kind[q]) >= 2 ? row_init_value(t, key, st1_off[q]) : 0.f;
      }
    } else {
      vload<V>(w, row + e0);
      if (need_state) {
        if (OPTK == OPTK_ADAGRAD || OPTK == OPTK_FTRL) vload<V>(s0, row + D + e0);
        if (OPTK == OPTK_FTRL) vload<V>(s1, row + 2 * D + e0);
        if (OPTK == OPTK_MULTI) {
#pragma unroll
          for (int q = 0; q < V; ++q) {
            s0[q] = opt_nstate(kind[q]) >= 1 ? row[st0_off[q]] : 0.f;
            s1[q] = opt_nstate(kind[q]) >= 2 ? row[st1_off[q]] : 0.f;
          }
        }
      }
    }
    if (OPTK == OPTK_MULTI && (need_state || fresh)) {
#pragma unroll
      for (int q = 0; q < V; ++q) optx_load(t, t.seg[segi[q]], row, e0 + q, fresh, key, ox[q]);
    }
    // --- walk the occurrences in input order
    if (mode == MODE_ASSIGN) {
      vload<V>(w, src_row<kPtrs>(a, a.sidx[end - 1], D) + e0);
      if (OPTK == OPTK_MULTI && t.serving) {
#pragma unroll
        for (int q = 0; q < V; ++q) w[q] = comp_q(t.seg[segi[q]], w[q]);
      }
    } else if (mode == MODE_ASSIGN_ADD) {
      int64_t p = first;
      if (!(OPTK == OPTK_MULTI && t.serving)) {
        // 4 independent row loads in flight, adds stay in input order (a walk with one load at a
        // time makes a 96-occurrence segment a ~150 us chain of memory round trips)
        for (; p + 4 <= end; p += 4) {
          float g0[V], g1[V], g2[V], g3[V];
          vload<V>(g0, src_row<kPtrs>(a, a.sidx[p], D) + e0);
          vload<V>(g1, src_row<kPtrs>(a, a.sidx[p + 1], D) + e0);
          vload<V>(g2, src_row<kPtrs>(a, a.sidx[p + 2], D) + e0);
          vload<V>(g3, src_row<kPtrs>(a, a.sidx[p + 3], D) + e0);
#pragma unroll
          for (int q = 0; q < V; ++q) w[q] = (((w[q] + g0[q]) + g1[q]) + g2[q]) + g3[q];
        }
      }
      for (; p < end; ++p) {
        float g[V];
        vload<V>(g, src_row<kPtrs>(a, a.sidx[p], D) + e0);
#pragma unroll
        for (int q = 0; q < V; ++q) w[q] += g[q];
        if (OPTK == OPTK_MULTI && t.serving) {  // every AssignAdd re-encodes (entry_accessor.cc:70-77)
#pragma unroll
          for (int q = 0; q < V; ++q) w[q] = comp_q(t.seg[segi[q]], w[q]);
        }
      }
    } else if (mode == MODE_OPT_SEQ) {
      auto step_one = [&](const float(&g)[V], uint32_t pos) {  // one Optimize call of this id
#pragma unroll
        for (int q = 0; q < V; ++q) {
          const SegDev& sg = t.seg[segi[q]];
          if (OPTK == OPTK_MULTI && t.any_sr && sg.sr_pre >= 0)
            w[q] = stochastic_round(w[q], sr_uniform(t.seed, a.global_step, pos, e0 + q, sg.sr_pre));
          opt_step(kind[q], sg, a.lr[segi[q]], g[q], w[q], s0[q], s1[q], ox[q]);
          if (OPTK == OPTK_MULTI && t.any_sr && sg.sr_post)
            w[q] = stochastic_round(w[q], sr_uniform(t.seed, a.global_step, pos, e0 + q, segi[q]));
        }
      };
      int64_t p = first;
      for (; p + 4 <= end; p += 4) {  // the steps depend on each other, the row loads do not
        const uint32_t i0 = a.sidx[p], i1 = a.sidx[p + 1], i2 = a.sidx[p + 2], i3 = a.sidx[p + 3];
        float g0[V], g1[V], g2[V], g3[V];
        vload<V>(g0, src_row<kPtrs>(a, i0, D) + e0);
        vload<V>(g1, src_row<kPtrs>(a, i1, D) + e0);
        vload<V>(g2, src_row<kPtrs>(a, i2, D) + e0);
        vload<V>(g3, src_row<kPtrs>(a, i3, D) + e0);
        step_one(g0, i0);
        step_one(g1, i1);
        step_one(g2, i2);
        step_one(g3, i3);
      }
      for (; p < end; ++p) {
        float g[V];
        const uint32_t i0 = a.sidx[p];
        vload<V>(g, src_row<kPtrs>(a, i0, D) + e0);
        step_one(g, i0);
      }
    } else {  // MODE_OPT_DEDUP: in-order sum (bridge :251-266), then one step
      float acc[V];
      vload<V>(acc, src_row<kPtrs>(a, first == j ? idx0 : a.sidx[first], D) + e0);
      int64_t p = first + 1;
      // the stateless sums (layout gradient) meet tens of thousands of 10..96-row segments per
      // batch: eight rows in flight, their occurrence values fetched by one coalesced load of the
      // sub-group (lane u holds value u) instead of one dependent load each. (V = 4 would spill
      // at 8 waves / SIMD; the shuffle needs all 8 lanes inside this column loop.)
      if (MODE_T == MODE_SET_SUM && V <= 2 && D % (8 * V) == 0) {
        for (; p + 8 <= end; p += 8) {
          const uint32_t mine = a.sidx[p + sub];
          float g[8][V];
#pragma unroll
          for (int u = 0; u < 8; ++u) vload<V>(g[u], src_row<kPtrs>(a, __shfl(mine, u, 8), D) + e0);
#pragma unroll
          for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int q = 0; q < V; ++q) acc[q] = acc[q] + g[u][q];
        }
      }
      for (; p + 4 <= end; p += 4) {  // 4 independent loads in flight, adds stay in order
        float g0[V], g1[V], g2[V], g3[V];
        vload<V>(g0, src_row<kPtrs>(a, a.sidx[p], D) + e0);
        vload<V>(g1, src_row<kPtrs>(a, a.sidx[p + 1], D) + e0);
        vload<V>(g2, src_row<kPtrs>(a, a.sidx[p + 2], D) + e0);
        vload<V>(g3, src_row<kPtrs>(a, a.sidx[p + 3], D) + e0);
#pragma unroll
        for (int q = 0; q < V; ++q) acc[q] = (((acc[q] + g0[q]) + g1[q]) + g2[q]) + g3[q];
      }
      for (; p < end; ++p) {
        float g[V];
        vload<V>(g, src_row<kPtrs>(a, a.sidx[p], D) + e0);
#pragma unroll
        for (int q = 0; q < V; ++q) acc[q] = acc[q] + g[q];
      }
      const uint32_t pos0 = first == j ? idx0 : a.sidx[first];
      if (mode == MODE_SET_SUM) {
#pragma unroll
        for (int q = 0; q < V; ++q) w[q] = acc[q];
      } else
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const SegDev& sg = t.seg[segi[q]];
        if (OPTK == OPTK_MULTI && t.any_sr && sg.sr_pre >= 0)
          w[q] = stochastic_round(w[q], sr_uniform(t.seed, a.global_step, pos0, e0 + q, sg.sr_pre));
        opt_step(kind[q], sg, a.lr[segi[q]], acc[q], w[q], s0[q], s1[q], ox[q]);
        if (OPTK == OPTK_MULTI && t.any_sr && sg.sr_post)
          w[q] = stochastic_round(w[q], sr_uniform(t.seed, a.global_step, pos0, e0 + q, segi[q]));
      }
    }
    // --- store
    vstore<V>(row + e0, w);
    if (need_state || fresh) {
      if (OPTK == OPTK_ADAGRAD || OPTK == OPTK_FTRL) vstore<V>(row + D + e0, s0);
      if (OPTK == OPTK_FTRL) vstore<V>(row + 2 * D + e0, s1);
      if (OPTK == OPTK_MULTI) {
#pragma unroll
        for (int q = 0; q < V; ++q) {
          if (opt_nstate(kind[q]) >= 1) row[st0_off[q]] = s0[q];
          if (opt_nstate(kind[q]) >= 2) row[st1_off[q]] = s1[q];
          optx_store(t.seg[segi[q]], row, e0 + q, ox[q]);
        }
      }
    }
  }
  if (sub == 0 && t.ts) t.ts[slot] = a.ts;
}

// Segment classification (no-filter path): kClsItems sorted positions per thread. Segment heads
// are compacted into `heads` (so the sub-group kernel has no idle groups) or, when longer than
// kLongSeg in a walking mode, pushed to the long-segment queue. Head slots are handed out
// through an LDS counter and ONE global atomic per workgroup: a device-scope atomic on a single
// word retires at only ~90 per microsecond, so a per-wave atomic would cost more than the sort.
constexpr int kClsThreads = 1024;
constexpr int kClsItems = 4;
__global__ void __launch_bounds__(kClsThreads) k_classify(ApplyArgs a, int allow_long,
                                                          uint4* __restrict__ heads) {
  __shared__ uint32_t s_cnt, s_base, s_lq, s_lp, s_lc, s_lb, s_lqb, s_lpb, s_lcb, s_lbb, s_nl;
  __shared__ uint4 s_ll[kClsBlkList];
  const uint32_t* __restrict__ sk = static_cast<const uint32_t*>(a.skeys);
  const int64_t n = a.n;
  const int tid = threadIdx.x, lane = tid & 63;
  // a sort that failed (look-back time-out) leaves stale ping-pong data behind: emit no heads and
  // no long segments, so that nothing is applied; the host reports the word at its next read-back
  if (a.sort_err && *a.sort_err != 0u) return;
  if (tid == 0) s_cnt = s_lq = s_lp = s_lc = s_lb = s_nl = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * (kClsThreads * kClsItems);
  uint32_t loc[kClsItems], len[kClsItems], slotv[kClsItems], idxv[kClsItems];
  // long segments: queue entries / partial rows / split-key entries reserved in LDS first, one
  // global atomic per counter and workgroup (a Zipf-heavy batch queues thousands of them)
  uint32_t lq[kClsItems], lp[kClsItems], lc[kClsItems], lb[kClsItems];
  bool sh[kClsItems];
#pragma unroll
  for (int k = 0; k < kClsItems; ++k) {
    const int64_t j = base + (int64_t)k * kClsThreads + tid;
    sh[k] = false;
    len[k] = 1;
    loc[k] = 0;
    lq[k] = kNoPart;
    lp[k] = lc[k] = lb[k] = 0;
    slotv[k] = idxv[k] = 0;
    if (j < n) {
      const uint32_t slot = sk[j];
      slotv[k] = slot;
      idxv[k] = a.sidx[j];
      const uint32_t prev = j > 0 ? sk[j - 1] : ~slot;
      const uint32_t next = j + 1 < n ? sk[j + 1] : ~slot;
      bool keep = prev != slot && slot != kNoSlot && slot != a.skip_slot &&
                  (a.sur_base == 0u || slot < a.sur_base);
      if (keep && a.want_dim)
        keep = a.cls_dim[a.src_tab ? (idxv[k] & a.pk_smask) : a.cls[idxv[k]]] == a.want_dim;
      if (keep) {
        if (next == slot) len[k] = (uint32_t)(segment_end<uint32_t>(sk, j, n, slot) - j);
        if (allow_long && len[k] > (uint32_t)kLongSeg) {
          const uint32_t nparts = long_parts(a, len[k]);
          lq[k] = atomicAdd(&s_lq, nparts);
          if (a.blk_src) lb[k] = atomicAdd(&s_lb, long_blocks(a, len[k]));
          if (long_split(a, len[k])) {
            lp[k] = atomicAdd(&s_lp, nparts);
            lc[k] = atomicAdd(&s_lc, 1u);
          }
        } else {
          sh[k] = true;
        }
      }
    }
    const unsigned long long m = __ballot(sh[k]);
    if (m) {
      const int leader = __ffsll((long long)m) - 1;
      uint32_t wb = 0;
      if (lane == leader) wb = atomicAdd(&s_cnt, (uint32_t)__popcll(m));
      wb = __shfl(wb, leader);
      loc[k] = wb + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
  }
  __syncthreads();
  if (tid == 0) s_base = s_cnt ? atomicAdd(&a.qcount[1], s_cnt) : 0u;
  if (tid == 64) s_lqb = s_lq ? atomicAdd(&a.qcount[0], s_lq) : 0u;
  if (tid == 128) s_lpb = s_lp ? atomicAdd(&a.qcount[3], s_lp) : 0u;
  if (tid == 192) s_lcb = s_lc ? atomicAdd(&a.qcount[2], s_lc) : 0u;
  if (tid == 256) s_lbb = s_lb ? atomicAdd(&a.qcount[7], s_lb) : 0u;
  __syncthreads();
  const uint32_t gb = s_base;
#pragma unroll
  for (int k = 0; k < kClsItems; ++k) {
    const int64_t j = base + (int64_t)k * kClsThreads + tid;
    if (sh[k]) heads[gb + loc[k]] = make_uint4((uint32_t)j, len[k], slotv[k], idxv[k]);
    if (lq[k] != kNoPart)
      enqueue_long_at(a, j, j + len[k], 0, slotv[k], 0u, s_lqb + lq[k], s_lpb + lp[k], s_lcb + lc[k],
                      s_lbb + lb[k], s_ll, &s_nl);
  }
  if (a.blk_src) {
    __syncthreads();
    write_block_list(a, s_ll, s_nl);
  }
}

// Long segments: one 256-thread workgroup per key (or per chunk of a split key), split into
// roles. Waves 1-3 (192 loader threads) gather the occurrence rows, 16 bytes per lane and load,
// TWO tiles ahead of the consumer (two register stages, two LDS tiles; occurrence indices run
// further ahead through an LDS ring, so a row load never waits on an index load): ~800 rows of
// 64 bytes are in flight per workgroup, which is what hides an HBM round trip at the rate the
// consumer walks. Wave 0 only consumes: lane e owns column e and adds / steps the rows of the
// column-major LDS tile in input order, four rows per 16-byte LDS load, sixteen rows in flight —
// bit-exact with the sequential reference, one dependent floating-point add per row.
constexpr int kLoaders = 192;
constexpr int kLTileFloatsDefault = 6144;      // 24 KB per LDS tile, 8 float4 per loader thread
constexpr int kLMaxRows = 2 * kLoaders;        // rows per tile (two prefetched indices per thread)
constexpr int kLIdxRing = 8;

// OCC: workgroups per CU the register budget is set for, TILE: floats per LDS tile. Four workgroups
// with half-size tiles (36 KB of LDS each) for the layout gradient's sums (it runs millions of rows
// through these chains — one in-order chain per workgroup — and nothing else competes for the CUs'
// registers); one beside the table's k_apply and for the exchange's gradient sum, where
// the second workgroup costs the kernel that carries the traffic more than it gains (config 2:
// 355 -> 396 us per step, config 4: 1.07 -> 1.12 ms).
template <int OPTK, int MODE, int OCC = 1, int TILE = kLTileFloatsDefault>
__global__ void __launch_bounds__(256, OCC) k_apply_long(TableDev t, ApplyArgs a) {
  constexpr int kLTileFloats = TILE;                     // floats per LDS tile
  constexpr int kLStage4 = kLTileFloats / 4 / kLoaders;  // float4 per loader thread per tile
  // Register stages = tiles in flight ahead of the consumer. Two are enough: with four (tiles
  // tl + 1 .. tl + 4 in flight) a lone 30k-row chain beside k_apply takes the same 249 us — the
  // chain is bound by the dependent add itself (~8 ns per row), not by the loaders.
  constexpr int NS = 2;
  static_assert(NS + 3 <= kLIdxRing, "index ring too short");
  static_assert(kLStage4 * 4 * kLoaders == kLTileFloats, "tile size must be a multiple of 4 * kLoaders");
  __shared__ __attribute__((aligned(16))) float tile[2][kLTileFloats];
  __shared__ uint32_t idxs[kLIdxRing][kLMaxRows];
  __shared__ uint32_t s_qi;
  const int D = t.dim, R = t.row_floats;
  const int tid = threadIdx.x;
  const bool loader = tid >= 64;
  const int lt = tid - 64;  // loader thread index
  const bool vec4 = (D & 3) == 0 && ((a.src_ptrs || a.src_tab) ? a.src_aligned4 != 0
                                                : (reinterpret_cast<uintptr_t>(a.src) & 15) == 0);
  constexpr bool kSums = MODE == MODE_OPT_DEDUP || MODE == MODE_SET_SUM;
  const uint32_t nq = guard_ok(a) ? *a.qcount : 0u;
  // The consumer wave is a lone dependent chain; when this kernel shares CUs with k_apply its
  // issue slots must not be arbitrated away, so it runs at raised wave priority.
  if (!loader) __builtin_amdgcn_s_setprio(3);
  // Entries differ in length by 40x: they are handed out one at a time.
  for (;;) {
    __syncthreads();
    if (tid == 0) s_qi = atomicAdd(&a.qcount[4], 1u);
    __syncthreads();
    const uint32_t qi = __builtin_amdgcn_readfirstlane(s_qi);  // (uniform: ls lives in SGPRs)
    if (qi >= nq) break;
    const LongSeg ls = a.queue[qi];
    float* row = t.rows + (uint64_t)ls.slot * R;
    for (int eb = 0; eb < D; eb += 64) {  // 64 columns per pass (one pass for dim <= 64)
      const int Dw = min(64, D - eb);     // columns of this pass
      // rows per tile; the per-thread walk over the tile's (row, column) cells, without a
      // division per cell: cell k * kLoaders + lt, cells = float4 quads (vec4) or floats
      // The LDS tile is column-major, [column][RS]: the consumer's lane e then reads FOUR
      // consecutive rows of its column with one 16-byte LDS load. RS = 4 * odd spreads the lanes'
      // loads (and the loaders' transposed stores) over the banks.
      const int RS = ((min(kLMaxRows, kLTileFloats / Dw) - 4) & ~7) + 4;
      const int rpt = RS;
      const int64_t ntiles = (ls.end - ls.first + rpt - 1) / rpt;
      const int cpr = vec4 ? Dw / 4 : Dw;  // cells per row
      const int r0 = loader ? lt / cpr : 0, c0 = loader ? lt - r0 * cpr : 0;
      const int dr = kLoaders / cpr, dc = kLoaders - dr * cpr;
      const int e = eb + tid;
      const bool owner = !loader && tid < Dw;
      int kind = OPTK, segi = 0, st0 = D + e, st1 = 2 * D + e;
      if (OPTK == OPTK_MULTI && owner) {
        for (int z = 0; z < t.n_seg; ++z)
          if (e >= t.seg[z].w_off && e < t.seg[z].w_off + t.seg[z].dim) segi = z;
        kind = t.seg[segi].opt_kind;
        st0 = t.seg[segi].st_off + (e - t.seg[segi].w_off);
        st1 = st0 + t.seg[segi].dim;
      }
      float w = 0.f, s0 = 0.f, s1 = 0.f, acc = 0.f;
      OptX ox;
      ox.gstep = a.global_step;
      const bool need_state = MODE == MODE_OPT_SEQ || MODE == MODE_OPT_DEDUP;
      const bool part = ls.part != kNoPart;  // one chunk of a split key: emit the chunk sum only
      if (owner && !part && MODE != MODE_SET_SUM) {
        if (ls.fresh) {
          w = row_init_value(t, ls.key, e);
          s0 = opt_nstate(kind) >= 1 ? row_init_value(t, ls.key, st0) : 0.f;
          s1 = opt_nstate(kind) >= 2 ? row_init_value(t, ls.key, st1) : 0.f;
        } else {
          w = row[e];
          if (need_state) {
            s0 = opt_nstate(kind) >= 1 ? row[st0] : 0.f;
            s1 = opt_nstate(kind) >= 2 ? row[st1] : 0.f;
          }
        }
        if (OPTK == OPTK_MULTI && (need_state || ls.fresh))
          optx_load(t, t.seg[segi], row, e, ls.fresh != 0, ls.key, ox);
      }
      const SegDev& sg = t.seg[segi];
      const float lr = a.lr[segi];
      auto load_idx = [&](int64_t tl, int half) -> uint32_t {
        const int rr = lt + half * kLoaders;
        const int64_t p = ls.first + tl * rpt + rr;
        return (loader && rr < rpt && p < ls.end) ? a.sidx[p] : 0u;
      };
      // loaders: gather tile tl into a register stage (16 bytes per cell in the vec4 form, else
      // the same registers hold 4 x as many 4-byte cells)
      auto issue = [&](float4(&S)[kLStage4], int64_t tl) {
        if (!loader || tl >= ntiles) return;
        const int rows = (int)min((int64_t)rpt, ls.end - (ls.first + tl * rpt));
        const uint32_t* ix = idxs[tl & (kLIdxRing - 1)];
        int r = r0, c = c0;
        if (vec4) {
#pragma unroll
          for (int k = 0; k < kLStage4; ++k) {
            S[k] = (r < rows) ? *reinterpret_cast<const float4*>(src_row<MODE == MODE_SET_SUM>(a, ix[r], D) + eb + 4 * c)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
            r += dr;
            c += dc;
            if (c >= cpr) { c -= cpr; ++r; }
          }
        } else {
          float* Sf = reinterpret_cast<float*>(&S[0]);
#pragma unroll
          for (int k = 0; k < 4 * kLStage4; ++k) {
            Sf[k] = (r < rows) ? src_row<MODE == MODE_SET_SUM>(a, ix[r], D)[eb + c] : 0.f;
            r += dr;
            c += dc;
            if (c >= cpr) { c -= cpr; ++r; }
          }
        }
      };
      auto put = [&](const float4(&S)[kLStage4], int buf) {  // column-major [column][RS] store
        if (!loader) return;
        int r = r0, c = c0;
        if (vec4) {
#pragma unroll
          for (int k = 0; k < kLStage4; ++k) {
            if (r < rpt) {
              float* q = &tile[buf][4 * c * RS + r];
              q[0] = S[k].x;
              q[RS] = S[k].y;
              q[2 * RS] = S[k].z;
              q[3 * RS] = S[k].w;
            }
            r += dr;
            c += dc;
            if (c >= cpr) { c -= cpr; ++r; }
          }
        } else {
          const float* Sf = reinterpret_cast<const float*>(&S[0]);
#pragma unroll
          for (int k = 0; k < 4 * kLStage4; ++k) {
            if (r < rpt) tile[buf][c * RS + r] = Sf[k];
            r += dr;
            c += dc;
            if (c >= cpr) { c -= cpr; ++r; }
          }
        }
      };
      auto step = [&](float v) {
        if (MODE == MODE_ASSIGN_ADD) w += v;
        else if (MODE == MODE_OPT_SEQ) opt_step(kind, sg, lr, v, w, s0, s1, ox);
        else acc = acc + v;
      };
      auto compute = [&](int64_t tl, int buf) {  // wave 0: walk the tile in input order
        if (!owner) return;
        const int rows = (int)min((int64_t)rpt, ls.end - (ls.first + tl * rpt));
        const float4* tp = reinterpret_cast<const float4*>(&tile[buf][tid * RS]);
        const int G = rows >> 2, rem = rows & 3;  // whole groups of four rows, rows of the last one
        int g = 0;                                // (uniform: the loop control below is scalar)
        if (kSums && tl == 0) {                   // first row seeds the sum
          const float4 x = tp[0];
          acc = x.x;
          if (rows > 1) step(x.y);
          if (rows > 2) step(x.z);
          if (rows > 3) step(x.w);
          g = 1;
        }
        // A lone wave issues one instruction every four cycles, so the chain's cost is its
        // instruction count: one LDS load per four rows, and the NEXT sixteen rows loaded before
        // this sixteen's dependent adds (two register sets, alternating; the loads are
        // unconditional — past the tile's rows they read stale LDS that is never used).
#define MONO_STEP4(q) step(q.x); step(q.y); step(q.z); step(q.w)
        if (g + 4 <= G) {
          float4 a0 = tp[g], a1 = tp[g + 1], a2 = tp[g + 2], a3 = tp[g + 3];
          for (;;) {
            // (scheduling barriers: the compiler otherwise sinks these loads below the adds,
            // next to the following set's, and waits for both before the first add)
            // (and the explicit wait — lgkmcnt(0), nothing else — lets the set loaded sixteen adds
            // ago land BEFORE the next loads are issued: placed after them it would wait for both)
            __builtin_amdgcn_s_waitcnt(0xC07F);
            const float4 b0 = tp[g + 4], b1 = tp[g + 5], b2 = tp[g + 6], b3 = tp[g + 7];
            __builtin_amdgcn_sched_barrier(0);
            MONO_STEP4(a0); MONO_STEP4(a1); MONO_STEP4(a2); MONO_STEP4(a3);
            __builtin_amdgcn_sched_barrier(0);
            g += 4;
            if (g + 4 > G) break;
            __builtin_amdgcn_s_waitcnt(0xC07F);
            a0 = tp[g + 4]; a1 = tp[g + 5]; a2 = tp[g + 6]; a3 = tp[g + 7];
            __builtin_amdgcn_sched_barrier(0);
            MONO_STEP4(b0); MONO_STEP4(b1); MONO_STEP4(b2); MONO_STEP4(b3);
            __builtin_amdgcn_sched_barrier(0);
            g += 4;
            if (g + 4 > G) break;
          }
        }
        for (; g < G; ++g) {
          const float4 x = tp[g];
          MONO_STEP4(x);
        }
#undef MONO_STEP4
        if (rem != 0 && g == G) {  // (g > G: the seeding group above was the only, partial one)
          const float4 x = tp[G];
          step(x.x);
          if (rem > 1) step(x.y);
          if (rem > 2) step(x.z);
        }
      };
      // one round: tile tl sits in LDS[tl & 1]; stage S carries tile tl + 1, the other stage
      // tile tl + 2. Park tl + 1, refill S with tl + 3, advance the index ring, consume tl.
      uint32_t nx0 = 0, nx1 = 0;
      auto round = [&](int64_t tl, float4(&S)[kLStage4]) {
        const int cur = (int)(tl & 1);
        if (tl + 1 < ntiles) put(S, cur ^ 1);
        if (loader) {
          idxs[(tl + NS + 2) & (kLIdxRing - 1)][lt] = nx0;
          idxs[(tl + NS + 2) & (kLIdxRing - 1)][lt + kLoaders] = nx1;
        }
        nx0 = load_idx(tl + NS + 3, 0);
        nx1 = load_idx(tl + NS + 3, 1);
        issue(S, tl + NS + 1);
        compute(tl, cur);
        __syncthreads();
      };

      __syncthreads();  // LDS of the previous segment / column pass is free
      if (loader) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int z = 0; z < NS + 2; ++z) idxs[z][lt + h * kLoaders] = load_idx(z, h);
      }
      __syncthreads();
      nx0 = load_idx(NS + 2, 0);
      nx1 = load_idx(NS + 2, 1);
      if constexpr (NS == 2) {
        float4 Sa[kLStage4], Sb[kLStage4];
        issue(Sa, 0);
        put(Sa, 0);
        issue(Sb, 1);  // tile 1 -> parked by round 0
        issue(Sa, 2);  // tile 2 -> parked by round 1
        __syncthreads();
        for (int64_t tl = 0; tl < ntiles; tl += 2) {
          round(tl, Sb);
          if (tl + 1 < ntiles) round(tl + 1, Sa);
        }
      } else {  // four stages: tile k travels in stage k % 4, tiles tl + 1 .. tl + 4 are in flight
        float4 Sa[kLStage4], Sb[kLStage4], Sc[kLStage4], Sd[kLStage4];
        issue(Sa, 0);
        put(Sa, 0);
        issue(Sb, 1);
        issue(Sc, 2);
        issue(Sd, 3);
        issue(Sa, 4);
        __syncthreads();
        for (int64_t tl = 0; tl < ntiles; tl += 4) {
          round(tl, Sb);
          if (tl + 1 < ntiles) round(tl + 1, Sc);
          if (tl + 2 < ntiles) round(tl + 2, Sd);
          if (tl + 3 < ntiles) round(tl + 3, Sa);
        }
      }
      if (owner && part) {
        a.partials[(uint64_t)ls.part * D + e] = MODE == MODE_ASSIGN_ADD ? w : acc;
      } else if (owner) {
        if (MODE == MODE_OPT_DEDUP) opt_step(kind, sg, lr, acc, w, s0, s1, ox);
        if (MODE == MODE_SET_SUM) w = acc;
        row[e] = w;
        if (MODE != MODE_SET_SUM && (need_state || ls.fresh)) {
          if (opt_nstate(kind) >= 1) row[st0] = s0;
          if (opt_nstate(kind) >= 2) row[st1] = s1;
          if (OPTK == OPTK_MULTI) optx_store(sg, row, e, ox);
        }
      }
    }
    if (tid == 0 && ls.part == kNoPart && t.ts) t.ts[ls.slot] = a.ts;
  }
}

// Split keys: add the chunk sums in chunk order, then finish like an unsplit segment. One wave
// per key, lane e owns column e.
template <int OPTK>
__global__ void __launch_bounds__(64) k_combine(TableDev t, ApplyArgs a) {
  const int D = t.dim, R = t.row_floats;
  const uint32_t nc = guard_ok(a) ? a.qcount[2] : 0u;
  for (uint32_t ci = blockIdx.x; ci < nc; ci += gridDim.x) {
    const SplitSeg sp = a.cqueue[ci];
    float* row = t.rows + (uint64_t)sp.slot * R;
    for (int e = threadIdx.x; e < D; e += 64) {
      int kind = OPTK, segi = 0, st0 = D + e, st1 = 2 * D + e;
      if (OPTK == OPTK_MULTI) {
        for (int z = 0; z < t.n_seg; ++z)
          if (e >= t.seg[z].w_off && e < t.seg[z].w_off + t.seg[z].dim) segi = z;
        kind = t.seg[segi].opt_kind;
        st0 = t.seg[segi].st_off + (e - t.seg[segi].w_off);
        st1 = st0 + t.seg[segi].dim;
      }
      const bool need_state = a.mode == MODE_OPT_SEQ || a.mode == MODE_OPT_DEDUP;
      float w, s0 = 0.f, s1 = 0.f;
      OptX ox;
      ox.gstep = a.global_step;
      if (OPTK == OPTK_MULTI && (need_state || sp.fresh))
        optx_load(t, t.seg[segi], row, e, sp.fresh != 0, sp.key, ox);
      if (a.mode == MODE_SET_SUM) {
        w = 0.f;
      } else if (sp.fresh) {
        w = row_init_value(t, sp.key, e);
        s0 = opt_nstate(kind) >= 1 ? row_init_value(t, sp.key, st0) : 0.f;
        s1 = opt_nstate(kind) >= 2 ? row_init_value(t, sp.key, st1) : 0.f;
      } else {
        w = row[e];
        if (need_state) {
          s0 = opt_nstate(kind) >= 1 ? row[st0] : 0.f;
          s1 = opt_nstate(kind) >= 2 ? row[st1] : 0.f;
        }
      }
      const float* pp = a.partials + (uint64_t)sp.pbase * D + e;
      if (a.mode == MODE_ASSIGN_ADD) {
        for (uint32_t p = 0; p < sp.nparts; ++p) w += pp[(uint64_t)p * D];
      } else {
        float acc = pp[0];
        for (uint32_t p = 1; p < sp.nparts; ++p) acc = acc + pp[(uint64_t)p * D];
        if (a.mode == MODE_SET_SUM) w = acc;
        else opt_step(kind, t.seg[segi], a.lr[segi], acc, w, s0, s1, ox);
      }
      row[e] = w;
      if (a.mode != MODE_SET_SUM && (need_state || sp.fresh)) {
        if (opt_nstate(kind) >= 1) row[st0] = s0;
        if (opt_nstate(kind) >= 2) row[st1] = s1;
        if (OPTK == OPTK_MULTI) optx_store(t.seg[segi], row, e, ox);
      }
    }
    if (threadIdx.x == 0 && t.ts) t.ts[sp.slot] = a.ts;
  }
}


// ---- hot-key chains (chain_dev.h) ---------------------------------------------------------------
// Pack pass: the rows of every long segment are copied, in sorted (= input) order, into 64-row
// blocks whose four-row groups are column-major — cell (g, e) = rows 4g .. 4g+3 of column e — so
// that the chain wave reads four rows of its column with one 16-byte LDS load and fetches 1 KiB of
// consecutive memory per DMA instruction. Rows past a block's count are -0.0f (x + -0.0f == x for
// every x, signed zeros included). Massively parallel and bandwidth-bound: the one dependent
// chain per key only starts behind it.
template <bool PTRS>
__global__ void __launch_bounds__(256) k_long_pack(TableDev t, ApplyArgs a) {
  const int D = t.dim;
  // (the block records are read before the block count is known — up to `lim` they are in bounds —
  // so that the count and the record are one memory level, not two)
  const uint32_t lim = min(a.blk_cap, a.packed_bytes / (uint32_t)(kPackRows * D * sizeof(float)));
  const uint32_t* nbp = a.qcount + 7;
  const bool vec4 = (D & 3) == 0 && ((a.src_ptrs || a.src_tab) ? a.src_aligned4 != 0
                                                : (reinterpret_cast<uintptr_t>(a.src) & 15) == 0);
  const float nz = -0.0f;
  if (vec4) {
    // task = (group g, column quad qd): four 16-byte row loads, a 4 x 4 transpose, four cells
    const int qpr = D / 4, tpb = 16 * qpr;  // quads per row, tasks per block
    const int bpw = tpb >= 256 ? 1 : 256 / tpb;
    for (int t0 = threadIdx.x; t0 < bpw * tpb; t0 += 256) {
      const int tb = t0 / tpb, tt = t0 - tb * tpb;
      const int g = tt / qpr, qd = tt - g * qpr;
      for (uint32_t b = blockIdx.x * bpw + tb; b < lim; b += gridDim.x * bpw) {
        const uint2 bs = a.blk_src[b];
        if (b >= *nbp) break;
        float4 r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t rw = 4 * g + i;
          r[i] = make_float4(nz, nz, nz, nz);
          if (rw < bs.y)
            r[i] = *reinterpret_cast<const float4*>(src_row<PTRS>(a, a.sidx[(uint64_t)bs.x + rw], D) + 4 * qd);
        }
        float4* dst = reinterpret_cast<float4*>(a.packed + (uint64_t)b * kPackRows * D) + g * D + 4 * qd;
        dst[0] = make_float4(r[0].x, r[1].x, r[2].x, r[3].x);
        dst[1] = make_float4(r[0].y, r[1].y, r[2].y, r[3].y);
        dst[2] = make_float4(r[0].z, r[1].z, r[2].z, r[3].z);
        dst[3] = make_float4(r[0].w, r[1].w, r[2].w, r[3].w);
      }
    }
  } else {
    for (uint32_t b = blockIdx.x; b < lim; b += gridDim.x) {
      const uint2 bs = a.blk_src[b];
      if (b >= *nbp) break;
      float4* dst = reinterpret_cast<float4*>(a.packed + (uint64_t)b * kPackRows * D);
      for (int c = threadIdx.x; c < 16 * D; c += 256) {
        const int g = c / D, e = c - g * D;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t rw = 4 * g + i;
          v[i] = rw < bs.y ? src_row<PTRS>(a, a.sidx[(uint64_t)bs.x + rw], D)[e] : nz;
        }
        dst[c] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}

// Ring slots of a chain wave by row width: RV for a very long chain that has its CU to itself,
// RM for one of AW waves that share a CU (128 KB of LDS either way).
template <int D> struct ChainCfg;
template <> struct ChainCfg<8> { static constexpr int RV = 60, RM = 8, AW = 16; };
template <> struct ChainCfg<16> { static constexpr int RV = 60, RM = 8, AW = 16; };
template <> struct ChainCfg<32> { static constexpr int RV = 56, RM = 8, AW = 16; };
template <> struct ChainCfg<64> { static constexpr int RV = 48, RM = 16, AW = 8; };
constexpr int kChainLds = 128 * 1024;

// one queue entry, start to finish, by one wave: row / state of the key into registers, the
// chain over its packed rows, then what k_apply_long does at the end of a segment
template <int OPTK, int MODE, int D, int R>
__device__ __forceinline__ void chain_entry(const TableDev& t, const ApplyArgs& a, const LongSeg& ls,
                                            chain_v4i rs, uint32_t ring) {
  const int e = (int)__lane_id();
  const bool owner = e < D;
  const bool part = ls.part != kNoPart;  // one chunk of a split key: emit the chunk sum only
  float* row = t.rows + (uint64_t)ls.slot * t.row_floats;
  constexpr int kind = OPTK;
  const int st0 = D + e, st1 = 2 * D + e;
  float w = 0.f, s0 = 0.f, s1 = 0.f;
  OptX ox;
  ox.gstep = a.global_step;
  if (owner && !part && MODE != MODE_SET_SUM) {
    if (ls.fresh) {
      w = row_init_value(t, ls.key, e);
      s0 = opt_nstate(kind) >= 1 ? row_init_value(t, ls.key, st0) : 0.f;
      s1 = opt_nstate(kind) >= 2 ? row_init_value(t, ls.key, st1) : 0.f;
    } else {
      w = row[e];
      if (MODE == MODE_OPT_DEDUP) {
        s0 = opt_nstate(kind) >= 1 ? row[st0] : 0.f;
        s1 = opt_nstate(kind) >= 2 ? row[st1] : 0.f;
      }
    }
  }
  // assign_add adds the rows to the weight one by one; the sums start from the first row
  // (-0.0f + x == x)
  const float seed = (MODE == MODE_ASSIGN_ADD && !part) ? w : -0.0f;
  const uint32_t nblk = (uint32_t)((ls.end - ls.first + kPackRows - 1) / kPackRows);
  const float acc = chain_sum<D, R>(rs, ls.blk * (uint32_t)(kPackRows * D * sizeof(float)), nblk * (D / 4),
                                    ring, seed);
  if (owner && part) {
    a.partials[(uint64_t)ls.part * D + e] = acc;
  } else if (owner) {
    if (MODE == MODE_OPT_DEDUP) opt_step(kind, t.seg[0], a.lr[0], acc, w, s0, s1, ox);
    else w = acc;
    row[e] = w;
    if (MODE != MODE_SET_SUM && (MODE == MODE_OPT_DEDUP || ls.fresh)) {
      if (opt_nstate(kind) >= 1) row[st0] = s0;
      if (opt_nstate(kind) >= 2) row[st1] = s1;
    }
  }
  if (e == 0 && !part && t.ts) t.ts[ls.slot] = a.ts;
}
template <int OPTK, int MODE, bool VL>
__device__ __forceinline__ void chain_entry_d(const TableDev& t, const ApplyArgs& a, const LongSeg& ls,
                                              chain_v4i rs, uint32_t lds0, int w) {
#define MONO_CHAIN_CASE(DD)                                                                       \
  case DD:                                                                                        \
    if (VL) chain_entry<OPTK, MODE, DD, ChainCfg<DD>::RV>(t, a, ls, rs, lds0);                    \
    else chain_entry<OPTK, MODE, DD, ChainCfg<DD>::RM>(t, a, ls, rs, lds0 + (uint32_t)w * ChainCfg<DD>::RM * 1024u); \
    break;
  switch (t.dim) {
    MONO_CHAIN_CASE(8)
    MONO_CHAIN_CASE(16)
    MONO_CHAIN_CASE(32)
    MONO_CHAIN_CASE(64)
    default: break;
  }
#undef MONO_CHAIN_CASE
}

// Holds the stream it is launched on until `want` workgroups of the chain kernel have started
// (or the time is up): the sub-group kernel behind it would otherwise, now and then, fill every
// CU before the chain workgroups — which need whole CUs — are placed, and the longest chain then
// runs as the tail of the phase (226 us instead of 115, profiles/r03_c_step_timeline.txt).
__global__ void __launch_bounds__(64) k_chain_gate(uint32_t* started, uint32_t want, unsigned timeout_us) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    while (atomicAdd(started, 0u) < want &&
           __builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)timeout_us * 100ull)
      __builtin_amdgcn_s_sleep(8);
  }
}

// The chains' kernel: workgroups of 16 waves built for 128 VGPRs per lane, so that a workgroup
// fills the register file of its CU and no other kernel's wave runs beside it. That is the
// point: beside k_apply's 32 waves a chain wave gets a thirty-third of its CU's memory pipeline
// (26 ns per row measured, scripts/exp_chain.hip) — alone on the CU it runs at the rate of its
// instruction stream, 3.4 - 3.6 ns per row with the chip busy or not. A CU under load feeds about
// one full-rate chain (19 GB/s), so a very long chain (> kVeryLong rows) gets a CU to itself:
// wave 0 of every workgroup drains that queue first, longest class first, while the other waves
// wait; then all waves take the remaining entries, one entry per wave.
template <int OPTK, int MODE>
__global__ void __launch_bounds__(1024) k_long_chain(TableDev t, ApplyArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char chain_lds[];
  __shared__ volatile int s_vl_done;
  if (threadIdx.x == 0) atomicAdd(&a.qcount[6], 1u);  // this workgroup has its CU (k_chain_gate)
  const uint32_t nq = a.qcount[0];
  if (nq == 0 || !guard_ok(a)) return;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const chain_v4i rs = chain_rsrc(a.packed, a.packed_bytes);
  const uint32_t lds0 = (uint32_t)(uintptr_t)chain_lds;
  if (threadIdx.x == 0) s_vl_done = 0;
  __syncthreads();
  if (w == 0) {
    uint32_t cnt[kVlClasses], total = 0;
#pragma unroll
    for (int c = 0; c < kVlClasses; ++c) {
      cnt[c] = min(a.qcount[8 + c], a.vl_cap);
      total += cnt[c];
    }
    for (;;) {
      uint32_t tk = 0;
      if (lane == 0) tk = atomicAdd(&a.qcount[5], 1u);
      tk = __builtin_amdgcn_readfirstlane(tk);
      if (tk >= total) break;
      int c = 0;
#pragma unroll
      for (int z = 0; z < kVlClasses - 1; ++z)
        if (c == z && tk >= cnt[z]) { tk -= cnt[z]; c = z + 1; }
      const LongSeg ls = a.queue[a.vlq[(uint32_t)c * a.vl_cap + tk]];
      chain_entry_d<OPTK, MODE, true>(t, a, ls, rs, lds0, 0);
    }
    if (lane == 0) s_vl_done = 1;
  } else {
    while (s_vl_done == 0) __builtin_amdgcn_s_sleep(32);  // (wave 0 always gets there)
  }
  int active = 16;
  if (t.dim == 64) active = ChainCfg<64>::AW;
  if (w < active) {
    for (;;) {
      uint32_t qi = 0;
      if (lane == 0) qi = atomicAdd(&a.qcount[4], 1u);
      qi = __builtin_amdgcn_readfirstlane(qi);
      if (qi >= nq) break;
      const LongSeg ls = a.queue[qi];
      if (ls.end - ls.first > kVeryLong) continue;  // (served above)
      chain_entry_d<OPTK, MODE, false>(t, a, ls, rs, lds0, w);
    }
  }
  __syncthreads();  // the workgroup keeps its CU until its last chain is done
}

// D1: cuckoo_embedding_hash_table.cc:220-233. One thread per slot, 12 B (key + ts) per slot.
// The two counters are bumped once per WORKGROUP (counts kept per thread over the grid-stride
// loop, summed through LDS): an eviction that removes tens of millions of rows otherwise queues
// millions of atomics on two words (~90 per microsecond: 100 ms for config 5's first eviction).
__global__ void __launch_bounds__(256) k_evict(TableDev t, int64_t now) {
  __shared__ unsigned int s_dead, s_dead_special;
  if (threadIdx.x == 0) s_dead = s_dead_special = 0;
  __syncthreads();
  const int64_t total = (int64_t)t.cap + 2;
  unsigned int n_dead = 0, n_special = 0;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total;
       s += (int64_t)gridDim.x * blockDim.x) {
    const bool special = s >= (int64_t)t.cap;
    int64_t key;
    bool live;
    if (!special) {
      key = t.keys[s];
      live = key != kEmpty && key != kTomb;
    } else {
      key = s == (int64_t)t.cap ? kEmpty : kTomb;
      live = t.special[s - t.cap] != 0;
    }
    const bool dead = live && is_expired(t, key, t.ts[s], now);
    if (dead) {
      if (!special) t.keys[s] = kTomb; else t.special[s - t.cap] = 0;
      ++n_dead;
      n_special += special;
    }
  }
  if (n_dead) atomicAdd(&s_dead, n_dead);
  if (n_special) atomicAdd(&s_dead_special, n_special);
  __syncthreads();
  if (threadIdx.x == 0 && s_dead) {
    atomicAdd(&t.counters[0], (unsigned long long)(-(long long)s_dead));
    atomicAdd(&t.counters[1], (unsigned long long)(s_dead - s_dead_special));
  }
}

__global__ void __launch_bounds__(256) k_export(TableDev t, int64_t max_rows, int apply_expiry,
                                                int64_t now, int64_t* ids, float* rows,
                                                uint32_t* ts) {
  const int64_t total = (int64_t)t.cap + 2;
  const int R = t.row_floats;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total;
       s += (int64_t)gridDim.x * blockDim.x) {
    int64_t key;
    bool live;
    if (s < (int64_t)t.cap) {
      key = t.keys[s];
      live = key != kEmpty && key != kTomb;
    } else {
      key = s == (int64_t)t.cap ? kEmpty : kTomb;
      live = t.special[s - t.cap] != 0;
    }
    if (live && apply_expiry && is_expired(t, key, t.ts[s], now)) live = false;
    // wave-aggregated cursor: one atomic per wave
    const unsigned long long active = __ballot(true);
    const unsigned long long m = __ballot(live);
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)active) - 1;
    unsigned long long base = 0;
    if (lane == leader && m) base = atomicAdd(&t.counters[2], (unsigned long long)__popcll(m));
    base = __shfl(base, leader);
    if (live) {
      unsigned long long pos = base + __popcll(m & ((1ull << lane) - 1ull));
      if ((int64_t)pos < max_rows) {
        ids[pos] = key;
        ts[pos] = t.ts[s];
        for (int e = 0; e < R; ++e) rows[pos * R + e] = t.rows[(uint64_t)s * R + e];
      }
    }
  }
}

// Re-insert every live row of `o` into the (empty) table `nw`: one sub-group per old bucket.
__global__ void __launch_bounds__(256) k_rehash(TableDev o, TableDev nw) {
  const int sub = lane_sub(), gshift = lane_gshift();
  const int64_t ngroups = (int64_t)gridDim.x * (blockDim.x / 8);
  const int64_t nb = (int64_t)o.bucket_mask + 1;
  const int R = o.row_floats;
  for (int64_t b = (int64_t)blockIdx.x * (blockDim.x / 8) + threadIdx.x / 8; b < nb; b += ngroups) {
    int64_t k = o.keys[b * kBucket + sub];
    uint32_t m = group_bits(k != kEmpty && k != kTomb, gshift);
    while (m) {
      int p = __ffs(m) - 1;
      m &= m - 1;
      int64_t key = __shfl(k, p, 8);
      bool claimed;
      uint32_t ns = group_find_or_claim(nw, key, sub, gshift, &claimed);
      const float* src = o.rows + (uint64_t)(b * kBucket + p) * R;
      float* dst = nw.rows + (uint64_t)ns * R;
      for (int e = sub; e < R; e += 8) dst[e] = src[e];
      if (sub == 0) nw.ts[ns] = o.ts[b * kBucket + p];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side

static uint64_t pow2_at_least(uint64_t x) {
  uint64_t p = 1024;
  while (p < x) p <<= 1;
  return p;
}

// Slots are 32-bit (kNoSlot = 0xFFFFFFFF, slots cap and cap + 1 hold the sentinel-valued keys) and
// the write path sorts them on at most 32 bits: a table never grows past 2^31 slots per GPU
// (1.29e9 rows at the 0.6 load bound); beyond that the call fails instead of truncating slots.
constexpr uint64_t kMaxCap = 1ull << 31;

static int alloc_arrays(TableDev* d, uint64_t cap) {
  MONO_REQUIRE(cap <= kMaxCap,
               "table of %llu slots exceeds the %llu-slot limit of 32-bit slot indices "
               "(at most %.3g rows per GPU: shard the table over more GPUs)",
               (unsigned long long)cap, (unsigned long long)kMaxCap, kMaxLoad * (double)kMaxCap);
  d->cap = cap;
  d->bucket_mask = cap / kBucket - 1;
  MONO_HIP(hipMalloc((void**)&d->keys, cap * sizeof(int64_t)));
  MONO_HIP(hipMalloc((void**)&d->ts, (cap + 2) * sizeof(uint32_t)));
  MONO_HIP(hipMalloc((void**)&d->rows, (cap + 2) * (uint64_t)d->row_floats * sizeof(float)));
  MONO_HIP(hipMalloc((void**)&d->special, 2 * sizeof(uint32_t)));
  MONO_HIP(hipMalloc((void**)&d->counters, kCounterWords * sizeof(unsigned long long)));
  return 0;
}

static void free_arrays(TableDev* d) {
  if (d->keys) (void)hipFree(d->keys);
  if (d->ts) (void)hipFree(d->ts);
  if (d->rows) (void)hipFree(d->rows);
  if (d->special) (void)hipFree(d->special);
  if (d->counters) (void)hipFree(d->counters);
  d->keys = nullptr; d->ts = nullptr; d->rows = nullptr; d->special = nullptr; d->counters = nullptr;
}

static int reset_arrays(TableDev* d, hipStream_t st) {
  MONO_HIP(hipMemsetAsync(d->keys, 0xFF, d->cap * sizeof(int64_t), st));
  MONO_HIP(hipMemsetAsync(d->special, 0, 2 * sizeof(uint32_t), st));
  MONO_HIP(hipMemsetAsync(d->counters, 0, kCounterWords * sizeof(unsigned long long), st));
  return 0;
}

static int read_counters(Table* t, hipStream_t st, int64_t* live, int64_t* tombs) {
  static thread_local unsigned long long h[kCounterWords];
  uint32_t h_err = 0;
  MONO_HIP(hipMemcpyAsync(h, t->d.counters, sizeof(h), hipMemcpyDeviceToHost, st));
  if (uint32_t* w = sort_error_word())
    MONO_HIP(hipMemcpyAsync(&h_err, w, sizeof(h_err), hipMemcpyDeviceToHost, st));
  MONO_HIP(hipStreamSynchronize(st));
  unsigned long long lv = h[0];  // (k_evict subtracts here: the sum is what counts, modulo 2^64)
  for (int k = 0; k < kLiveStripes; ++k) lv += h[kLiveStripe0 + 8 * k];
  *live = (int64_t)lv;
  *tombs = (int64_t)h[1];
  return sort_error_report(h_err);  // every host read-back of the write path reports a failed sort
}

static int hint_drop(Table* t, hipStream_t st);
static bool keep_hints_enabled() {  // MONO_KEEP_HINTS=0: a gated update drops the queued hints
  static const bool v = [] {
    const char* e = getenv("MONO_KEEP_HINTS");
    return !e || atoi(e) != 0;
  }();
  return v;
}
static int keys_changed(Table* t, hipStream_t st);

// Rehash into a table of `new_cap` slots (drops tombstones). Peak memory = old + new arrays.
static int rehash(Table* t, uint64_t new_cap, hipStream_t st) {
  t->cache_n = 0;  // every key moves
  t->prep.n = 0;
  MONO_TRY(hint_drop(t, st));
  TableDev nw = t->d;
  nw.keys = nullptr; nw.ts = nullptr; nw.rows = nullptr; nw.special = nullptr; nw.counters = nullptr;
  MONO_TRY(alloc_arrays(&nw, new_cap));
  MONO_TRY(reset_arrays(&nw, st));
  const int64_t nb = (int64_t)t->d.bucket_mask + 1;
  hipLaunchKernelGGL(k_rehash, dim3(grid_for(nb, 32)), dim3(256), 0, st, t->d, nw);
  MONO_HIP(hipGetLastError());
  const size_t R = t->d.row_floats;
  MONO_HIP(hipMemcpyAsync(nw.rows + nw.cap * R, t->d.rows + t->d.cap * R, 2 * R * sizeof(float),
                          hipMemcpyDeviceToDevice, st));
  MONO_HIP(hipMemcpyAsync(nw.ts + nw.cap, t->d.ts + t->d.cap, 2 * sizeof(uint32_t),
                          hipMemcpyDeviceToDevice, st));
  MONO_HIP(hipMemcpyAsync(nw.special, t->d.special, 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
  MONO_HIP(hipMemcpyAsync(nw.counters, t->d.counters, sizeof(unsigned long long),
                          hipMemcpyDeviceToDevice, st));
  MONO_HIP(hipMemcpyAsync(nw.counters + kLiveStripe0, t->d.counters + kLiveStripe0,
                          8 * kLiveStripes * sizeof(unsigned long long), hipMemcpyDeviceToDevice, st));
  MONO_HIP(hipStreamSynchronize(st));
  free_arrays(&t->d);
  t->d = nw;
  return keys_changed(t, st);
}

// Called before any launch that can insert up to `incoming` new keys.
static bool slot_cache_enabled() {  // MONO_SLOT_CACHE=0 switches the lookup -> update hand-off off
  static const bool on = [] {
    const char* e = getenv("MONO_SLOT_CACHE");
    return !e || atoi(e) != 0;
  }();
  return on;
}

static int ensure_room(Table* t, int64_t incoming, hipStream_t st) {
  const int64_t limit = (int64_t)(kMaxLoad * (double)t->d.cap);
  if (t->used_upper + incoming <= limit) {
    t->used_upper += incoming;
    return 0;
  }
  int64_t live, tombs;
  MONO_TRY(read_counters(t, st, &live, &tombs));
  if (live + tombs + incoming > limit) {
    uint64_t want = pow2_at_least((uint64_t)((double)(live + incoming) / kGrowTarget) + 1);
    // near the slot-index limit settle for any capacity that keeps the load bound
    if (want > kMaxCap && (double)(live + incoming) <= kMaxLoad * (double)kMaxCap) want = kMaxCap;
    MONO_TRY(rehash(t, want, st));
    tombs = 0;
  }
  t->used_upper = live + tombs + incoming;
  return 0;
}

template <int V>
static void launch_lookup(const Table* t, const int64_t* ids, int64_t n, float* out, hipStream_t st,
                          int64_t* cache_ids = nullptr, uint32_t* cache_slots = nullptr) {
  hipLaunchKernelGGL(k_lookup<V>, dim3(grid_for(n, 32 * kIlpL)), dim3(256), 0, st, t->d, ids, n, out,
                     cache_ids, cache_slots);
}

template <int V, int OPTK>
static void launch_apply_vo(const Table* t, const ApplyArgs& a, hipStream_t st) {
  // (with compacted heads a sub-group beyond *nheads has nothing to do: when the caller knows a
  // bound on the distinct keys — the layout gradient's 1.4 M rows for 6.5 M uses — four fifths of
  // the workgroups need not be dispatched at all)
  const int64_t work = a.heads && a.max_heads > 0 ? std::min(a.n, a.max_heads) : a.n;
  const dim3 grid((unsigned)((work + 31) / 32));
  // no-filter path (slots resolved, heads compacted): the walking modes as instantiations of
  // their own; everything else (filter path, assign, import) through the run-time kernel
  if (!a.from_keys && a.heads && a.mode == MODE_OPT_DEDUP)
    hipLaunchKernelGGL((k_apply<V, OPTK, MODE_OPT_DEDUP>), grid, dim3(256), 0, st, t->d, a);
  else if (!a.from_keys && a.heads && a.mode == MODE_OPT_SEQ)
    hipLaunchKernelGGL((k_apply<V, OPTK, MODE_OPT_SEQ>), grid, dim3(256), 0, st, t->d, a);
  else if (!a.from_keys && a.heads && a.mode == MODE_ASSIGN_ADD)
    hipLaunchKernelGGL((k_apply<V, OPTK, MODE_ASSIGN_ADD>), grid, dim3(256), 0, st, t->d, a);
  else if (OPTK == OPTK_SGD && !a.from_keys && a.heads && a.mode == MODE_SET_SUM)
    // (the segment sums of the stateless ops: F2 gradient, layout gradient)
    hipLaunchKernelGGL((k_apply<V, OPTK == OPTK_SGD ? OPTK : OPTK_SGD, MODE_SET_SUM>), grid, dim3(256), 0, st,
                       t->d, a);
  else
    hipLaunchKernelGGL((k_apply<V, OPTK>), grid, dim3(256), 0, st, t->d, a);
}
template <int V>
static void launch_apply_v(const Table* t, const ApplyArgs& a, hipStream_t st) {
  switch (t->opt_tmpl) {
    case OPTK_SGD: launch_apply_vo<V, OPTK_SGD>(t, a, st); break;
    case OPTK_ADAGRAD: launch_apply_vo<V, OPTK_ADAGRAD>(t, a, st); break;
    case OPTK_FTRL: launch_apply_vo<V, OPTK_FTRL>(t, a, st); break;
    default: launch_apply_vo<1, OPTK_MULTI>(t, a, st); break;
  }
}
template <int OPTK>
static void launch_long_m(const Table* t, const ApplyArgs& a, unsigned grid, hipStream_t st) {
  switch (a.mode) {
    case MODE_ASSIGN_ADD:
      hipLaunchKernelGGL((k_apply_long<OPTK, MODE_ASSIGN_ADD>), dim3(grid), dim3(256), 0, st, t->d, a);
      break;
    case MODE_OPT_SEQ:
      hipLaunchKernelGGL((k_apply_long<OPTK, MODE_OPT_SEQ>), dim3(grid), dim3(256), 0, st, t->d, a);
      break;
    case MODE_SET_SUM:
      if (a.src_tab || a.src_ptrs)
        hipLaunchKernelGGL((k_apply_long<OPTK, MODE_SET_SUM, 4, 3072>), dim3(grid), dim3(256), 0, st, t->d, a);
      else
        hipLaunchKernelGGL((k_apply_long<OPTK, MODE_SET_SUM>), dim3(grid), dim3(256), 0, st, t->d, a);
      break;
    default:
      hipLaunchKernelGGL((k_apply_long<OPTK, MODE_OPT_DEDUP>), dim3(grid), dim3(256), 0, st, t->d, a);
      break;
  }
}
// Packed 64-row blocks a batch of n occurrences can take: every queue entry (a whole segment or
// one part of a split key) has more than kLongSeg rows and wastes less than one block.
static size_t packed_blocks_bound(int64_t n) {
  return (size_t)n / kPackRows + (size_t)n / kLongSeg + 2;
}
// ---- hot-key chains: pack pass + chain kernel (MONO_LONG_CHAIN=0 keeps k_apply_long for everything)
static bool chain_enabled() {
  static const bool on = [] {
    const char* e = getenv("MONO_LONG_CHAIN");
    return !e || atoi(e) != 0;
  }();
  return on;
}
static bool chain_ok(const Table* t, int mode) {
  const int D = t->d.dim;
  return chain_enabled() && (D == 8 || D == 16 || D == 32 || D == 64) && t->opt_tmpl != OPTK_MULTI &&
         (mode == MODE_ASSIGN_ADD || mode == MODE_OPT_DEDUP ||
          (mode == MODE_SET_SUM && t->opt_tmpl == OPTK_SGD));
}
// workgroups (= CUs taken) of the chain kernel beside a table's k_apply; the stateless sums have
// the chip to themselves
static unsigned chain_wgs(const Table* t) {
  static const int v = [] {
    const char* e = getenv("MONO_CHAIN_WGS");
    return e ? std::max(1, std::min(atoi(e), 256)) : 24;
  }();
  // The stateless sums (layout gradient, exchange gradient) have no table traffic to protect, but
  // their short runs are summed by k_apply on the CUs the chains leave free: 64 chain CUs beside
  // 192 for k_apply (config 3's layout gradient: 0.92 ms with 256, 0.85 with 128, 0.79 with 64,
  // 0.85 with 40). MONO_CHAIN_WGS_CTX overrides.
  static const int c = [] {
    const char* e = getenv("MONO_CHAIN_WGS_CTX");
    return e ? std::max(1, std::min(atoi(e), 256)) : 64;
  }();
  return t->d.keys ? (unsigned)v : (unsigned)c;
}
template <int OPTK, int MODE>
static int launch_chain_om(const Table* t, const ApplyArgs& a, hipStream_t st) {
  static bool attr_set = false;  // (one process per GPU)
  if (!attr_set) {
    MONO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_long_chain<OPTK, MODE>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, kChainLds));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_long_chain<OPTK, MODE>), dim3(chain_wgs(t)), dim3(1024), kChainLds, st, t->d, a);
  return 0;
}
template <int OPTK>
static int launch_chain_o(const Table* t, const ApplyArgs& a, hipStream_t st) {
  if (a.mode == MODE_ASSIGN_ADD) return launch_chain_om<OPTK, MODE_ASSIGN_ADD>(t, a, st);
  return launch_chain_om<OPTK, MODE_OPT_DEDUP>(t, a, st);
}
static int launch_pack(const Table* t, const ApplyArgs& a, hipStream_t st) {
  // one task per thread where the batch allows (every load of the pass then in flight at once)
  const int64_t tasks = (int64_t)packed_blocks_bound(a.n) * 4 * t->d.dim;
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((tasks + 255) / 256, 4096));
  if (a.src_ptrs || a.src_tab) hipLaunchKernelGGL(k_long_pack<true>, dim3(grid), dim3(256), 0, st, t->d, a);
  else hipLaunchKernelGGL(k_long_pack<false>, dim3(grid), dim3(256), 0, st, t->d, a);
  return 0;
}
static int launch_chain(const Table* t, const ApplyArgs& a, hipStream_t st) {
  if (a.mode == MODE_SET_SUM) return launch_chain_om<OPTK_SGD, MODE_SET_SUM>(t, a, st);
  switch (t->opt_tmpl) {
    case OPTK_SGD: return launch_chain_o<OPTK_SGD>(t, a, st);
    case OPTK_ADAGRAD: return launch_chain_o<OPTK_ADAGRAD>(t, a, st);
    default: return launch_chain_o<OPTK_FTRL>(t, a, st);
  }
}
static void launch_combine(const Table* t, const ApplyArgs& a, hipStream_t st) {
  const unsigned cgrid = (unsigned)std::min<int64_t>(a.n / a.chunk + 1, 256);
  if (!(a.mode != MODE_OPT_SEQ && a.n > a.chunk)) return;
  switch (t->opt_tmpl) {
    case OPTK_SGD: hipLaunchKernelGGL(k_combine<OPTK_SGD>, dim3(cgrid), dim3(64), 0, st, t->d, a); break;
    case OPTK_ADAGRAD: hipLaunchKernelGGL(k_combine<OPTK_ADAGRAD>, dim3(cgrid), dim3(64), 0, st, t->d, a); break;
    case OPTK_FTRL: hipLaunchKernelGGL(k_combine<OPTK_FTRL>, dim3(cgrid), dim3(64), 0, st, t->d, a); break;
    default: hipLaunchKernelGGL(k_combine<OPTK_MULTI>, dim3(cgrid), dim3(64), 0, st, t->d, a); break;
  }
}

// the long segments of a classified batch: pack + chains when the shape allows (a.packed set by
// launch_apply), k_apply_long otherwise; then the chunk sums of split keys. `after_pack`: an event
// to record behind the pack pass (the sub-group kernel is started behind it, see launch_apply).
static int launch_long_o(const Table* t, const ApplyArgs& a, unsigned grid, hipStream_t st,
                         hipEvent_t after_pack = nullptr, bool prepacked = false) {
  if (a.packed) {
    if (!prepacked) {
      MONO_TRY(launch_pack(t, a, st));
      if (after_pack) MONO_HIP(hipEventRecord(after_pack, st));
    }
    MONO_TRY(launch_chain(t, a, st));
    launch_combine(t, a, st);
    return 0;
  }
  if (after_pack) MONO_HIP(hipEventRecord(after_pack, st));
  switch (t->opt_tmpl) {
    case OPTK_SGD: launch_long_m<OPTK_SGD>(t, a, grid, st); break;
    case OPTK_ADAGRAD: launch_long_m<OPTK_ADAGRAD>(t, a, grid, st); break;
    case OPTK_FTRL: launch_long_m<OPTK_FTRL>(t, a, grid, st); break;
    default: launch_long_m<OPTK_MULTI>(t, a, grid, st); break;
  }
  launch_combine(t, a, st);
  return 0;
}
static void launch_short(const Table* t, const ApplyArgs& a, hipStream_t st) {
  if (t->opt_tmpl == OPTK_MULTI || t->vec == 1) launch_apply_v<1>(t, a, st);
  else if (t->vec == 2) launch_apply_v<2>(t, a, st);
  else launch_apply_v<4>(t, a, st);
}

// What k_classify's output depends on besides the sorted arrays: 0 = no long segments (assign,
// import), 1 = long segments, split into chunks (assign_add, dedup), 2 = long segments, whole
// (per-occurrence optimizer steps).
static int mode_class(int mode) {
  return mode == MODE_OPT_SEQ ? 2
         : (mode == MODE_ASSIGN_ADD || mode == MODE_OPT_DEDUP || mode == MODE_SET_SUM) ? 1 : 0;
}
static bool long_allowed(const Table* t, int mode, int64_t n) {
  // (a serving entry re-encodes after every add: its segments are walked whole, never chunked)
  return mode_class(mode) != 0 && n > kLongSeg && t->d.dim <= kLongMaxDim && !t->d.serving &&
         !t->d.any_sr;
}
static void launch_classify(const ApplyArgs& a, bool allow_long, uint4* heads, hipStream_t st) {
  const int64_t per_block = (int64_t)kClsThreads * kClsItems;
  hipLaunchKernelGGL(k_classify, dim3((unsigned)((a.n + per_block - 1) / per_block)),
                     dim3(kClsThreads), 0, st, a, allow_long ? 1 : 0, heads);
}

// ---- host-side call timing (MONO_HOST_PROF=1): where the launch path spends its microseconds ----
bool host_prof_on() {
  static const bool v = getenv("MONO_HOST_PROF") != nullptr;
  return v;
}
struct HostProfRow { const char* name; double us; long calls; double min_us; double max_us; };
static HostProfRow g_host_prof[64];
static int g_host_prof_n = 0;
double host_now() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
static void host_prof_dump() {
  // (the average leaves the largest sample out: the first launch of a translation unit's kernels
  // loads its code object, ~8 ms once, which would pass for 135 us per call over 60 calls)
  for (int i = 0; i < g_host_prof_n; ++i) {
    const HostProfRow& r = g_host_prof[i];
    const double avg = r.calls > 1 ? (r.us - r.max_us) / (double)(r.calls - 1) : r.us;
    fprintf(stderr, "hostprof %-40s calls %8ld  avg %8.2f us  min %8.2f us  max %10.2f us\n", r.name,
            r.calls, avg, r.min_us, r.max_us);
  }
}
// adds (now - *t0) to `name` and restarts *t0
void host_tick(const char* name, double* t0) {
  if (!host_prof_on()) return;
  const double now = host_now();
  int i = 0;
  for (; i < g_host_prof_n; ++i)
    if (g_host_prof[i].name == name) break;
  if (i == g_host_prof_n) {
    if (g_host_prof_n == 64) return;
    if (g_host_prof_n == 0) atexit(host_prof_dump);
    g_host_prof[g_host_prof_n++] = HostProfRow{name, 0.0, 0, 1e30, 0.0};
  }
  static const bool trace = getenv("MONO_HOST_PROF_TRACE") != nullptr;  // every tick, in order
  if (trace) fprintf(stderr, "hosttick %-44s %9.2f us\n", name, now - *t0);
  g_host_prof[i].us += now - *t0;
  g_host_prof[i].min_us = std::min(g_host_prof[i].min_us, now - *t0);
  g_host_prof[i].max_us = std::max(g_host_prof[i].max_us, now - *t0);
  g_host_prof[i].calls += 1;
  *t0 = host_now();
}

static int apply_streams(Table* t) {
  if (t->aux) return 0;
  MONO_HIP(hipStreamCreateWithFlags(&t->aux, hipStreamNonBlocking));
  MONO_HIP(hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming));
  MONO_HIP(hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming));
  MONO_HIP(hipEventCreateWithFlags(&t->ev_pack, hipEventDisableTiming));
  return 0;
}
// Scratch for the packed rows of the long segments (worst case: every one wastes a block), or
// a.packed = nullptr when this call's long segments go to k_apply_long.
static int chain_setup(Table* t, ApplyArgs& a, bool allow_long) {
  a.packed = nullptr;
  a.packed_bytes = 0;
  if (!allow_long || !chain_ok(t, a.mode)) return 0;
  const size_t need = packed_blocks_bound(a.n) * kPackRows * (size_t)t->d.dim * sizeof(float) + 65536;
  if (need >= 0xFFFF0000ull) return 0;  // (32-bit buffer offsets)
  // (the gated remainder of a write call runs beside the known ids' chains, on another stream:
  // a pack buffer of its own)
  Workspace& ws = a.from_keys ? t->ws_pack_gated : t->ws_pack;
  MONO_TRY(ws.ensure(need));
  a.packed = (float*)ws.ptr;
  a.packed_bytes = (uint32_t)need;
  return 0;
}

// classified: k_classify has already run on these sorted arrays (for this mode's class)
// prepacked: ... and k_long_pack too (try_prepared launches it ahead of its host round trip;
// t->ev_pack is recorded behind it)
// `ready`: an event of another stream the (already classified) input still waits for; both apply
// streams wait on it directly, so neither start pays a fork behind the other.
// `ready_old`: the event is most likely over already (a plan made long ago): one wait on the
// caller's stream ahead of the fork instead of one per apply stream (a cross-stream wait costs
// 5-17 us of host time each, MONO_HOST_PROF).
static int launch_apply(Table* t, ApplyArgs& a, uint4* heads, uint32_t* nheads, hipStream_t st,
                        bool classified = false, hipEvent_t ready = nullptr, bool ready_old = false,
                        bool prepacked = false) {
  const bool allow_long = long_allowed(t, a.mode, a.n);
  const unsigned lgrid = (unsigned)std::min<int64_t>(a.n / kLongSeg + 1, 1024);
  double hp = host_prof_on() ? host_now() : 0.0;
  MONO_TRY(chain_setup(t, a, allow_long));
  if (ready && ready_old) {
    MONO_HIP(hipStreamWaitEvent(st, ready, 0));
    ready = nullptr;
    host_tick("apply: wait old ready (st)", &hp);
  }
  if (a.from_keys) {
    // filter path: the sub-group kernel resolves / gates / claims and feeds the long queue
    launch_short(t, a, st);
    if (allow_long) MONO_TRY(launch_long_o(t, a, lgrid, st));
    return 0;
  }
  if (!classified) launch_classify(a, allow_long, heads, st);
  a.heads = heads;
  a.nheads = nheads;
  if (!allow_long) {
    if (ready) MONO_HIP(hipStreamWaitEvent(st, ready, 0));
    launch_short(t, a, st);
    return 0;
  }
  // long segments are latency-bound on a few CUs: run them beside the sub-group kernel
  MONO_TRY(apply_streams(t));
  static const bool serial = getenv("MONO_SERIAL_APPLY") != nullptr;  // profiling aid
  if (serial) {
    if (ready) MONO_HIP(hipStreamWaitEvent(st, ready, 0));
    MONO_TRY(launch_long_o(t, a, lgrid, st));
    launch_short(t, a, st);
    return 0;
  }
  // The long kernel goes first on the caller's stream so its few hundred workgroups (LDS-heavy,
  // long-lived) are resident before the sub-group kernel's ~27k short workgroups flood the
  // dispatcher from the auxiliary stream; the other order leaves the long kernel to run
  // mostly after the short one has drained.
  hp = host_prof_on() ? host_now() : 0.0;
  MONO_HIP(hipEventRecord(t->ev_fork, st));
  host_tick(a.mode == MODE_SET_SUM ? "gradsum: record fork" : "apply: record fork", &hp);
  if (ready) MONO_HIP(hipStreamWaitEvent(st, ready, 0));
  host_tick(a.mode == MODE_SET_SUM ? "gradsum: wait ready (st)" : "apply: wait ready (st)", &hp);
  // The chains' workgroups need whole CUs: the sub-group kernel starts behind the pack pass, when
  // the chain kernel is next in line on its stream, not before (MONO_CHAIN_GATE=0: at the fork).
  static const bool gate = [] {
    const char* e = getenv("MONO_CHAIN_GATE");
    return !e || atoi(e) != 0;
  }();
  const bool gated = gate && a.packed != nullptr;
  MONO_TRY(launch_long_o(t, a, lgrid, st, gated ? t->ev_pack : nullptr, prepacked && a.packed));
  host_tick(a.mode == MODE_SET_SUM ? "gradsum: launch long+combine" : "apply: launch long+combine", &hp);
  MONO_HIP(hipStreamWaitEvent(t->aux, gated ? t->ev_pack : t->ev_fork, 0));
  if (ready && !gated) MONO_HIP(hipStreamWaitEvent(t->aux, ready, 0));
  host_tick(a.mode == MODE_SET_SUM ? "gradsum: aux waits" : "apply: aux waits", &hp);
  if (gated && chain_wgs(t) < 256u)
    hipLaunchKernelGGL(k_chain_gate, dim3(1), dim3(64), 0, t->aux, a.qcount + 6, chain_wgs(t), 40u);
  if (t->profile) MONO_HIP(hipEventRecord(t->ev_k[0], t->aux));
  launch_short(t, a, t->aux);
  host_tick(a.mode == MODE_SET_SUM ? "gradsum: launch short" : "apply: launch short", &hp);
  if (t->profile) {
    MONO_HIP(hipEventRecord(t->ev_k[1], t->aux));
    t->ev_k_valid = true;
  }
  MONO_HIP(hipEventRecord(t->ev_join, t->aux));
  host_tick(a.mode == MODE_SET_SUM ? "gradsum: record join" : "apply: record join", &hp);
  MONO_HIP(hipStreamWaitEvent(st, t->ev_join, 0));
  host_tick(a.mode == MODE_SET_SUM ? "gradsum: wait join" : "apply: wait join", &hp);
  return 0;
}

// Scratch behind the apply kernels: counters, long-segment queue, split-key queue, chunk sums,
// compacted heads.
// Default: no key is ever split, every sum runs strictly in
// input order at any length — the reference's own order bit for bit, at the price of one
// dependent chain per hot key (deviation 1).
static int chunk_size() {  // MONO_CHUNK=<rows>: sum hot keys in chunks, clamped to [256, 1 << 20]
  static const int v = [] {
    const char* e = getenv("MONO_CHUNK");
    if (!e || atoi(e) <= 0) return 1 << 30;  // default: every sum in the reference's order
    return std::max(256, std::min(atoi(e), 1 << 20));
  }();
  return v;
}
struct QueueLayout {
  size_t lq, cq, pt, heads, blk, vl;
  uint32_t vl_cap;
};
static QueueLayout queue_layout(int64_t n, int dim) {
  QueueLayout q;
  const size_t n_al = ((size_t)n + 63) & ~(size_t)63;
  q.lq = (((size_t)n / kLongSeg + 2) * sizeof(LongSeg) + 255) & ~(size_t)255;
  const size_t chunk = (size_t)chunk_size();
  q.cq = (((size_t)n / chunk + 2) * sizeof(SplitSeg) + 255) & ~(size_t)255;
  q.pt = (((size_t)n / (chunk / 2) + 2) * (size_t)dim * sizeof(float) + 255) & ~(size_t)255;
  q.heads = n_al * sizeof(uint4);
  // packed blocks: a segment of len rows takes ceil(len / 64) of them, at most n / 64 + one per
  // long segment (+ one per extra part of a split key)
  q.blk = ((packed_blocks_bound(n) + 2) * sizeof(uint2) + 255) & ~(size_t)255;
  q.vl_cap = (uint32_t)((size_t)n / kVeryLong + 2);
  q.vl = ((size_t)kVlClasses * q.vl_cap * sizeof(uint32_t) + 255) & ~(size_t)255;
  return q;
}
static size_t queue_bytes(int64_t n, int dim) {
  QueueLayout q = queue_layout(n, dim);
  return 256 + q.lq + q.cq + q.pt + q.heads + q.blk + q.vl;
}
static int queue_setup(ApplyArgs* a, char* q_base, int64_t n, int dim, uint4** heads,
                       hipStream_t st, bool counters_cleared = false) {
  QueueLayout q = queue_layout(n, dim);
  a->chunk = chunk_size();
  a->qcount = (uint32_t*)q_base;
  a->queue = (LongSeg*)(q_base + 256);
  a->cqueue = (SplitSeg*)(q_base + 256 + q.lq);
  a->partials = (float*)(q_base + 256 + q.lq + q.cq);
  *heads = (uint4*)(q_base + 256 + q.lq + q.cq + q.pt);
  a->blk_src = (uint2*)(q_base + 256 + q.lq + q.cq + q.pt + q.heads);
  a->blk_cap = (uint32_t)(q.blk / sizeof(uint2));
  a->vlq = (uint32_t*)(q_base + 256 + q.lq + q.cq + q.pt + q.heads + q.blk);
  a->vl_cap = q.vl_cap;
  if (!counters_cleared) MONO_HIP(hipMemsetAsync(a->qcount, 0, 16 * sizeof(uint32_t), st));
  return 0;
}

static int bits_for(uint64_t max_value) {
  int b = 1;
  while (b < 64 && (max_value >> b) != 0) ++b;
  return b;
}

// Every call that may change the key set (inserts, rehash, evict, clear, import) drops a pending
// hint and leaves an event behind itself: the next hint's probe waits for it.
static int hint_drop(Table* t, hipStream_t st) {
  if (t->hint_count) ++t->path_counts[3];
  // (a hint's probe reads the key array, which a rehash frees: the dropping call waits for it)
  for (int k = 0; k < t->hint_count; ++k) {
    Table::Hint& h = t->hints[(t->hint_head + k) % Table::kHintDepth];
    MONO_HIP(hipStreamWaitEvent(st, h.p.ev_ready, 0));
    h.p.chain_pending = false;
    h.p.n = 0;
  }
  t->hint_count = 0;
  return 0;
}
static int keys_changed(Table* t, hipStream_t st) {
  if (!t->ev_keys) MONO_HIP(hipEventCreateWithFlags(&t->ev_keys, hipEventDisableTiming));
  MONO_HIP(hipEventRecord(t->ev_keys, st));
  t->ev_keys_set = true;
  return 0;
}

// ---- prepared update order (Table::Prep) ------------------------------------------------------
static int prep_mode() {  // MONO_PREP: 0 = never, 1 = when the calls follow the pattern (default), 2 = always
  static const int v = [] {
    const char* e = getenv("MONO_PREP");
    return e ? atoi(e) : 1;
  }();
  return v;
}

static int prep_setup(Table* t, Table::Prep& p) {
  if (p.stream) return 0;
  // (a high-priority stream changes nothing here: what delays the sort beside the gather is
  // the depth of the memory queues, not the dispatch order — profiles/r02_b_step_timeline.txt)
  hipStream_t& cs = t->chain_stream[t->preps_made & 1];
  // (nor does a low-priority one for the hinted chains, or a smaller grid for their probe: 307 to
  // 314 us per step whatever is set, profiles/r02_q_hint_variants.txt)
  if (!cs) MONO_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  ++t->preps_made;
  p.stream = cs;
  MONO_HIP(hipEventCreateWithFlags(&p.ev_probe, hipEventDisableTiming));
  MONO_HIP(hipEventCreateWithFlags(&p.ev_ready, hipEventDisableTiming));
  MONO_HIP(hipEventCreateWithFlags(&p.ev_verify, hipEventDisableTiming));
  MONO_HIP(hipEventCreateWithFlags(&p.ev_gated, hipEventDisableTiming));
  MONO_HIP(hipEventCreateWithFlags(&p.ev_consumed, hipEventDisableTiming));
  MONO_HIP(hipMalloc((void**)&p.d_flags, 256));
  MONO_HIP(hipMemset(p.d_flags, 0, 256));
  MONO_HIP(hipMalloc((void**)&p.d_blk_miss, kProbeMaxBlocks * sizeof(uint32_t)));
  MONO_HIP(hipMemset(p.d_blk_miss, 0, kProbeMaxBlocks * sizeof(uint32_t)));
  MONO_HIP(hipHostMalloc((void**)&p.h_flags, 64, hipHostMallocDefault));
  p.h_flags[0] = p.h_flags[1] = 0;
  return 0;
}

static void prep_destroy_one(Table::Prep& p) {
  if (p.ev_probe) (void)hipEventDestroy(p.ev_probe);
  if (p.ev_ready) (void)hipEventDestroy(p.ev_ready);
  if (p.ev_verify) (void)hipEventDestroy(p.ev_verify);
  if (p.ev_gated) (void)hipEventDestroy(p.ev_gated);
  if (p.ev_consumed) (void)hipEventDestroy(p.ev_consumed);
  if (p.d_flags) (void)hipFree(p.d_flags);
  if (p.d_blk_miss) (void)hipFree(p.d_blk_miss);
  if (p.h_flags) (void)hipHostFree(p.h_flags);
  p.ws.release();
  p = Table::Prep{};
}
static void prep_destroy(Table* t) {
  for (hipStream_t& cs : t->chain_stream)
    if (cs) (void)hipStreamSynchronize(cs);
  prep_destroy_one(t->prep);
  for (Table::Hint& h : t->hints) {
    prep_destroy_one(h.p);
    if (h.ids) (void)hipFree(h.ids);
    if (h.slots) (void)hipFree(h.slots);
    h.ids = nullptr; h.slots = nullptr; h.cap = 0;
  }
  t->hint_head = t->hint_count = 0;
  for (hipStream_t& cs : t->chain_stream) {
    if (cs) (void)hipStreamDestroy(cs);
    cs = nullptr;
  }
  t->preps_made = 0;
  if (t->ev_hint_src) (void)hipEventDestroy(t->ev_hint_src);
  if (t->ev_keys) (void)hipEventDestroy(t->ev_keys);
  t->ev_hint_src = t->ev_keys = nullptr;
  t->ev_keys_set = false;
}

// Row gather of a lookup from a slot record: `blocks` workgroups, `ilp` rows in flight per lane group.
static int launch_gather(Table* t, const uint32_t* slots, int64_t n, float* out, bool aligned,
                         int gather_blocks, int gather_ilp, hipStream_t gs) {
  const int D = t->d.dim;
  static const int gather_nt = [] {
    const char* e = getenv("MONO_GATHER_NT");
    return e ? atoi(e) : 0;
  }();
#define MONO_GATHER(LPR, ILP)                                                                       \
  hipLaunchKernelGGL((k_gather_rows4<LPR, ILP>), dim3(grid_for(n, (256 / LPR) * ILP, gather_blocks)), \
                     dim3(256), 0, gs, t->d, slots, n, out, gather_nt)
#define MONO_GATHER_L(LPR)                                  \
  do {                                                      \
    if (gather_ilp <= 1) MONO_GATHER(LPR, 1);               \
    else if (gather_ilp <= 2) MONO_GATHER(LPR, 2);          \
    else if (gather_ilp <= 4) MONO_GATHER(LPR, 4);          \
    else MONO_GATHER(LPR, 8);                               \
  } while (0)
  if (D % 4 == 0 && t->d.row_floats % 4 == 0 && aligned) {
    if (D <= 16) MONO_GATHER_L(4);
    else if (D <= 32) MONO_GATHER_L(8);
    else MONO_GATHER_L(16);
  } else {
    hipLaunchKernelGGL(k_gather_rows1, dim3(grid_for(n, 32, gather_blocks)), dim3(256), 0, gs, t->d,
                       slots, n, out);
  }
#undef MONO_GATHER_L
#undef MONO_GATHER
  MONO_HIP(hipGetLastError());
  return 0;
}

// mono_table_lookup, prepared form: k_probe on the caller's stream, then the row gather there
// and the update order (sort by slot, k_classify) on the table's own stream beside it.
// `hint`: mono_table_lookup_hint — the same for the NEXT lookup's ids, everything (probe included)
// on the Prep's own stream `p.stream`, no gather, the slot record written to rec_ids / rec_slots.
static int lookup_prepared(Table* t, Table::Prep& p, const int64_t* ids, int64_t n, float* out,
                           bool aligned, hipStream_t st, bool hint, int64_t* rec_ids,
                           uint32_t* rec_slots) {
  MONO_TRY(prep_setup(t, p));
  p.n = 0;
  p.hinted = hint;
  // (tried: the sort chain on the caller's stream and the gather beside it — the update's host
  // read-back then sits behind the whole chain: 376 us per step against 357)
  hipStream_t gs = st;        // row gather
  hipStream_t ss = p.stream;  // sort + classify
  hipStream_t ps = hint ? ss : st;  // probe
  // A lookup that follows another one with no update in between: the chain of the earlier
  // lookup may still be running, and k_probe below resets the counters that chain's k_classify
  // adds to (a stale increment would make the next k_apply read heads nobody wrote).
  if (p.chain_pending && !hint) MONO_HIP(hipStreamWaitEvent(st, p.ev_ready, 0));
  const size_t n_al = ((size_t)n + 63) & ~(size_t)63;
  // Behind a hash filter (the last write call was gated): sort keys of their own, absent ids
  // grouped by surrogate keys sur_base + position (k_probe). sur_base lies above the two
  // special-key slots cap and cap + 1.
  const uint64_t sur_base64 = t->d.cap + 2;
  const bool sur = t->gated_seen && sur_base64 + (uint64_t)n < 0xFFFFFFF0ull;
  uint64_t sur_cells_n = 0;
  if (sur) {
    sur_cells_n = 1024;
    while (sur_cells_n < 2 * (uint64_t)n) sur_cells_n <<= 1;
  }
  const int end_bit = sur ? bits_for(sur_base64 + (uint64_t)n) : bits_for(t->d.cap + 1);
  const size_t tmp_bytes = (sort_pairs_tmp_bytes((size_t)n, end_bit) + 255) & ~(size_t)255;
  const size_t q_bytes = (queue_bytes(n, t->d.dim) + 255) & ~(size_t)255;
  const size_t sur_bytes = sur ? (n_al + (size_t)sur_cells_n) * sizeof(uint32_t) : 0;
  MONO_TRY(p.ws.ensure(2 * n_al * sizeof(uint32_t) + tmp_bytes + q_bytes + sur_bytes));
  p.sslots = (uint32_t*)p.ws.ptr;
  p.sidx = p.sslots + n_al;
  void* tmp = p.sidx + n_al;
  p.q_base = (char*)tmp + tmp_bytes;
  uint32_t* sort_in = sur ? (uint32_t*)(p.q_base + q_bytes) : nullptr;
  uint32_t* sur_cells = sur ? sort_in + n_al : nullptr;
  p.grouped_tail = sur;
  p.sur_base = sur ? (uint32_t)sur_base64 : 0u;
  double hp = host_prof_on() ? host_now() : 0.0;
  p.probe_blocks = grid_for(n, 32 * kIlpP, kProbeMaxBlocks);
  if (sur) MONO_HIP(hipMemsetAsync(sur_cells, 0xFF, (size_t)sur_cells_n * sizeof(uint32_t), ps));
  hipLaunchKernelGGL(k_probe, dim3(p.probe_blocks), dim3(256), 0, ps, t->d, ids, n, rec_ids, rec_slots,
                     p.d_blk_miss, (uint32_t*)p.q_base, p.d_flags, sort_in, sur_cells,
                     (uint32_t)(sur_cells_n ? sur_cells_n - 1 : 0), (uint32_t)sur_base64);
  MONO_HIP(hipGetLastError());
  host_tick("lookup: launch probe", &hp);
  MONO_HIP(hipEventRecord(p.ev_probe, ps));
  host_tick("lookup: record probe", &hp);
  // Row gather on the caller's stream, deliberately throttled: one workgroup per CU, two rows in
  // flight per lane group. At full depth (4096 workgroups x 8 rows) the gather keeps hundreds of
  // thousands of requests queued in the memory pipeline, every load of the latency-bound sort
  // beside it waits behind them (its first kernel took 52 us instead of 7, stream priority or
  // not: profiles/r02_a, r02_b), and nothing overlaps. Throttled, the gather takes ~2x longer but
  // the sort runs inside it: 388 -> 365 us per step (profiles/r02_sweep_gather.txt).
  // Whether the throttle pays depends on what the caller does between this lookup and the update:
  // back to back (the plain training step) the update waits for the order, and the throttle wins;
  // with other work in between (the sharded pipeline: row exchange, F2 gather, gradient sum) the
  // order is ready long before, and the throttle only slows the gather down. try_prepared sees
  // which case it is (was the order complete when the update asked?) and flips gather_full.
  static const int env_blocks = [] {
    const char* e = getenv("MONO_GATHER_BLOCKS");
    return e ? atoi(e) : 0;
  }();
  static const int env_ilp = [] {
    const char* e = getenv("MONO_GATHER_ILP");
    return e ? atoi(e) : 0;
  }();
  const bool full = p.gather_full && env_blocks == 0 && env_ilp == 0;
  const int gather_blocks = env_blocks ? env_blocks : full ? 4096 : 256;
  const int gather_ilp = env_ilp ? env_ilp : full ? 8 : 2;
  // (out == nullptr: slots only — table_lookup_slots, the fused lookup + pool gathers the rows itself)
  if (out && !hint) MONO_TRY(launch_gather(t, rec_slots, n, out, aligned, gather_blocks, gather_ilp, gs));
  host_tick("lookup: launch gather", &hp);
  // update order beside it
  if (!hint) MONO_HIP(hipStreamWaitEvent(ss, p.ev_probe, 0));
  host_tick("lookup: ss wait probe", &hp);
  MONO_TRY(sort_pairs_u32(sur ? sort_in : rec_slots, p.sslots, nullptr, p.sidx, (size_t)n, end_bit, tmp,
                          tmp_bytes, ss));
  host_tick("lookup: sort launches", &hp);
  ApplyArgs a{};
  a.skip_slot = kNoSlot;
  a.sur_base = sur ? (uint32_t)sur_base64 : 0u;
  a.n = n;
  a.skeys = p.sslots;
  a.sidx = p.sidx;
  a.mode = t->last_mode_class == 2 ? MODE_OPT_SEQ : t->last_mode_class == 1 ? MODE_OPT_DEDUP : MODE_ASSIGN;
  a.sort_err = sort_error_word();
  uint4* heads = nullptr;
  MONO_TRY(queue_setup(&a, p.q_base, n, t->d.dim, &heads, ss, /*counters_cleared=*/true));
  launch_classify(a, long_allowed(t, a.mode, n), heads, ss);
  MONO_HIP(hipGetLastError());
  host_tick("lookup: classify", &hp);
  p.mode_class = t->last_mode_class;
  MONO_HIP(hipEventRecord(p.ev_ready, ss));
  host_tick("lookup: record ready", &hp);
  p.chain_pending = true;

  p.n = n;
  return 0;
}

// The write call's side: use the prepared order if it provably serves this call.
static int try_prepared(Table* t, const int64_t* ids, const float* src, int64_t n, int mode,
                        const float* lrs, int n_lrs, int64_t req_time, hipStream_t st,
                        int64_t global_step, bool* tried, bool* used, bool allow_missing = false,
                        int64_t* n_missing = nullptr) {
  *tried = *used = false;
  if (n_missing) *n_missing = 0;
  Table::Prep& p = t->prep;
  if (p.n == 0 || p.n != n || t->cache_n != n || mode == MODE_IMPORT) return 0;
  *tried = true;
  if (t->profile) MONO_HIP(hipEventRecord(t->ev[0], st));
  // device side: every id was found by the lookup, and the ids are still the same ones
  double hp = host_prof_on() ? host_now() : 0.0;
  hipLaunchKernelGGL(k_verify, dim3(grid_for(n, 1024, 1024)), dim3(256), 0, st, ids, t->cache_ids, n,
                     p.d_flags, (uint32_t*)p.q_base + 4, p.d_blk_miss, p.probe_blocks);
  MONO_HIP(hipGetLastError());
  host_tick("update: launch verify", &hp);
  MONO_HIP(hipMemcpyAsync(p.h_flags, p.d_flags, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  host_tick("update: memcpy flags", &hp);
  MONO_HIP(hipEventRecord(p.ev_verify, st));
  host_tick("update: record verify", &hp);
  // While the host waits for the verdict: the pack pass of the hot keys' rows (it reads the
  // prepared order and this call's gradients, and writes scratch only, so a "no" costs nothing
  // but the pass). Only when the prepared classification is this call's and already complete.
  bool prepacked = false;
  static const bool prepack_on = [] {
    const char* e = getenv("MONO_PREPACK");
    return !e || atoi(e) != 0;
  }();
  if (prepack_on && p.mode_class == mode_class(mode) && long_allowed(t, mode, n) && chain_ok(t, mode)) {
    const bool done = hipEventQuery(p.ev_ready) == hipSuccess;
    (void)hipGetLastError();
    if (done) {
      ApplyArgs pa{};
      pa.n = n;
      pa.src = src;
      pa.mode = mode;
      pa.skeys = p.sslots;
      pa.sur_base = p.sur_base;
      pa.sidx = p.sidx;
      uint4* heads0 = nullptr;
      MONO_TRY(queue_setup(&pa, p.q_base, n, t->d.dim, &heads0, st, /*counters_cleared=*/true));
      MONO_TRY(chain_setup(t, pa, true));
      if (pa.packed) {
        MONO_TRY(apply_streams(t));
        MONO_TRY(launch_pack(t, pa, st));
        MONO_HIP(hipEventRecord(t->ev_pack, st));
        prepacked = true;
      }
    }
    host_tick("update: pre-pack", &hp);
  }
  // MONO_SPEC_APPLY=1: the apply kernels are launched before the host has the verdict, reading the
  // same flags on the device and doing nothing on "no"; the host reads the verdict afterwards.
  static const bool spec_on = [] {
    const char* e = getenv("MONO_SPEC_APPLY");
    return e && atoi(e) != 0;
  }();
  bool speculated = false;
  if (spec_on && !allow_missing && !t->profile && p.mode_class == mode_class(mode)) {
    ApplyArgs a{};
    a.skip_slot = kNoSlot;
    a.n = n;
    a.src = src;
    a.mode = mode;
    a.ts = (uint32_t)req_time;
    a.global_step = global_step;
    a.sort_err = sort_error_word();
    for (int s = 0; s < kMaxSeg; ++s) a.lr[s] = s < n_lrs ? lrs[s] : 0.f;
    a.skeys = p.sslots;
    a.sur_base = p.sur_base;
    a.sidx = p.sidx;
    a.guard = p.d_flags;
    a.guard_seen = p.seen_changed;
    a.guard_hinted = p.hinted ? 1 : 0;
    uint4* heads = nullptr;
    MONO_TRY(queue_setup(&a, p.q_base, n, t->d.dim, &heads, st, /*counters_cleared=*/true));
    const bool rdy = hipEventQuery(p.ev_ready) == hipSuccess;
    (void)hipGetLastError();
    MONO_TRY(launch_apply(t, a, heads, a.qcount + 1, st, true, rdy ? nullptr : p.ev_ready, false,
                          prepacked));
    speculated = true;
    host_tick("update: speculative apply launches", &hp);
  }
  MONO_HIP(hipEventSynchronize(p.ev_verify));
  host_tick("update: sync verify (incl. GPU wait)", &hp);
  const bool changed = p.h_flags[1] != p.seen_changed;
  p.seen_changed = p.h_flags[1];
  // other ids, or unknown ids to insert (unless the caller handles those itself: the gated path,
  // where the prepared order serves the ids the table holds and skips the rest): full path
  const uint32_t missed = p.h_flags[0];  // of the lookup this order was built by
  // (a hinted order whose lookup met other ids than the hinted ones was sorted for the wrong slots)
  const bool hint_off = p.hinted && p.h_flags[2] != 0;
  if ((missed != 0 && !allow_missing) || changed || hint_off) {
    p.n = 0;
    // (the lookups have left the hinted sequence: what is still queued would miss as well)
    if (hint_off) MONO_TRY(hint_drop(t, st));
    return 0;
  }
  if (n_missing) *n_missing = (int64_t)missed;
  if (speculated) {  // the kernels are in flight and the verdict says they do the work
    p.chain_pending = false;
    MONO_HIP(hipGetLastError());
    *used = true;
    return 0;
  }
  // was the prepared order already complete? (see lookup_prepared: the row gather's throttle)
  bool slack = false;
  {
    slack = hipEventQuery(p.ev_ready) == hipSuccess;
    (void)hipGetLastError();  // "not ready" is an answer, not an error to find later
    host_tick("update: query ready", &hp);
    if (p.full_backoff > 0) --p.full_backoff;
    if (!p.gather_full) {
      p.slack_run = slack ? p.slack_run + 1 : 0;
      if (p.slack_run >= 4 && p.full_backoff == 0) p.gather_full = true;
    } else if (!slack) {
      p.gather_full = false;
      p.slack_run = 0;
      p.full_backoff = p.full_backoff_len;
      p.full_backoff_len = std::min(p.full_backoff_len * 2, 4096);
    }
  }
  if (t->profile) MONO_HIP(hipEventRecord(t->ev[1], st));
  const bool classified = p.mode_class == mode_class(mode);
  // a prepared classification is consumed as it is: both apply streams wait for the prepared
  // order themselves (launch_apply); otherwise this stream waits here and classifies again
  if ((!classified || t->profile) && !slack) MONO_HIP(hipStreamWaitEvent(st, p.ev_ready, 0));
  if (t->profile) MONO_HIP(hipEventRecord(t->ev[2], st));

  ApplyArgs a{};
  a.skip_slot = kNoSlot;
  a.n = n;
  a.src = src;
  a.mode = mode;
  a.ts = (uint32_t)req_time;
  a.global_step = global_step;
  a.sort_err = sort_error_word();
  for (int s = 0; s < kMaxSeg; ++s) a.lr[s] = s < n_lrs ? lrs[s] : 0.f;
  a.skeys = p.sslots;
  a.sur_base = p.sur_base;
  a.sidx = p.sidx;
  uint4* heads = nullptr;
  // (prepared classification: its counters stand; k_verify has reset the hand-out cursor)
  MONO_TRY(queue_setup(&a, p.q_base, n, t->d.dim, &heads, st, /*counters_cleared=*/classified));
  MONO_TRY(launch_apply(t, a, heads, a.qcount + 1, st, classified,
                        classified && !t->profile && !slack ? p.ev_ready : nullptr, false,
                        prepacked && classified));
  p.chain_pending = false;  // the caller's stream has waited for ev_ready on every path above
  if (!classified) p.mode_class = mode_class(mode);  // the queues now hold this class
  MONO_HIP(hipGetLastError());
  if (t->profile) MONO_HIP(hipEventRecord(t->ev[3], st));
  *used = true;
  return 0;
}

// The unknown remainder of a gated write call, U positions in input order: either `flag` marks
// them (k_resolve_known) and `cnt` = k_unk_count's count per chunk, or `order_tail` lists them
// (the last U entries of a prepared order). Grouped by the 64-bit key, walked by the gated head
// kernel (filter counts, admission, claims; an id the table holds after all — inserted since the
// order was prepared — is found there and updated like any known id, the filter untouched).
static int unk_count(const uint32_t* flag, int64_t n, int32_t* cnt, hipStream_t st) {
  const UnkGeom g = unk_geom(n);
  hipLaunchKernelGGL(k_unk_count, dim3(g.chunks), dim3(256), 0, st, flag, n, g.chunk, cnt);
  MONO_HIP(hipGetLastError());
  return 0;
}
static int gated_unknown(Table* t, const ApplyArgs& a, const int64_t* ids, int64_t n,
                         const uint32_t* flag, const int32_t* cnt, const uint32_t* order_tail,
                         int64_t U, hipStream_t st, bool tail_grouped = false) {
  if (U > 0) {
    const size_t u_al = ((size_t)U + 63) & ~(size_t)63;
    const size_t tmp64 = (group_keys64_tmp_bytes((size_t)U) + 255) & ~(size_t)255;
    const size_t q2 = (queue_bytes(U, t->d.dim) + 255) & ~(size_t)255;
    // a workspace of its own: the first part's arrays are still being read on the device
    MONO_TRY(t->ws_gated.ensure(2 * u_al * sizeof(int64_t) + 2 * u_al * sizeof(uint32_t) + tmp64 + q2));
    int64_t* unk_ids = (int64_t*)t->ws_gated.ptr;
    int64_t* key_out = unk_ids + u_al;
    uint32_t* unk_idx = (uint32_t*)(key_out + u_al);
    uint32_t* idx_out2 = unk_idx + u_al;
    char* tmp2 = (char*)(idx_out2 + u_al);
    char* q_base2 = tmp2 + tmp64;
    if (order_tail && tail_grouped) {
      // (the lookup's probe gave every absent id a sort key of its own: the tail is grouped by id
      // already, input order inside a group — it IS the walk's order)
      hipLaunchKernelGGL(k_unk_from_order, dim3(grid_for(U, 256)), dim3(256), 0, st, ids, order_tail, U,
                         n, key_out, idx_out2);
    } else if (order_tail) {
      hipLaunchKernelGGL(k_unk_from_order, dim3(grid_for(U, 256)), dim3(256), 0, st, ids, order_tail, U,
                         n, unk_ids, unk_idx);
    } else {
      const UnkGeom ug = unk_geom(n);
      hipLaunchKernelGGL(k_unk_emit, dim3(ug.chunks), dim3(256), 0, st, ids, flag, n, ug.chunk, cnt,
                         unk_ids, unk_idx, U);
    }
    MONO_HIP(hipGetLastError());
    // grouped by id (input order inside a group), not sorted: the gated walk only asks where a
    // group starts and ends
    if (!(order_tail && tail_grouped))
      MONO_TRY(group_keys64(unk_ids, unk_idx, key_out, idx_out2, (size_t)U, tmp2, tmp64, st));
    ApplyArgs g = a;
    g.n = U;
    g.from_keys = 1;
    g.skeys = key_out;
    g.sidx = idx_out2;
    g.sort_err = nullptr;
    uint4* heads = nullptr;
    MONO_TRY(queue_setup(&g, q_base2, U, t->d.dim, &heads, st));
    MONO_TRY(launch_apply(t, g, heads, g.qcount + 1, st));
    MONO_HIP(hipGetLastError());
  }
  return 0;
}

// Write path with a hash filter. Ids the table already holds never touch the filter, and in a
// trained table that is nearly all of them: they are resolved to slots and take the plain path
// (32-bit slot sort, k_classify, the apply kernels). Only the unknown remainder — compacted in
// input order — is grouped by its 64-bit key and walked by the gated head kernel (filter counts,
// admission, slot claims: the closed forms of SURVEY 8a). The two parts touch disjoint rows.
static int write_path_gated(Table* t, void* filter, ApplyArgs a, const int64_t* ids, int64_t n,
                            hipStream_t st) {
  const size_t n_al = ((size_t)n + 63) & ~(size_t)63;
  const uint32_t skip = (uint32_t)(t->d.cap + 2);
  const int end_bit = bits_for(t->d.cap + 2);
  const size_t tmp32 = (sort_pairs_tmp_bytes((size_t)n, end_bit) + 255) & ~(size_t)255;
  const size_t cnt_bytes = kUnkMaxChunks * sizeof(int32_t);
  const size_t q_bytes = (queue_bytes(n, t->d.dim) + 255) & ~(size_t)255;
  // [slot_in | unknown | slot_out | idx_out | sort scratch | chunk counts | queues]
  const size_t part1 = 4 * n_al * sizeof(uint32_t) + tmp32 + cnt_bytes + q_bytes;
  MONO_TRY(t->ws.ensure(part1));
  uint32_t* slot_in = (uint32_t*)t->ws.ptr;
  int32_t* unknown = (int32_t*)(slot_in + n_al);
  uint32_t* slot_out = (uint32_t*)(unknown + n_al);
  uint32_t* idx_out = slot_out + n_al;
  char* tmp = (char*)(idx_out + n_al);
  int32_t* cnt = (int32_t*)(tmp + tmp32);
  char* q_base = (char*)cnt + cnt_bytes;
  hipLaunchKernelGGL(k_resolve_known, dim3(grid_for(n, 1024)), dim3(256), 0, st, t->d, ids, n, slot_in,
                     unknown, skip, t->cache_ids, t->cache_slots, t->cache_n);
  MONO_HIP(hipGetLastError());
  MONO_TRY(unk_count((const uint32_t*)unknown, n, cnt, st));
  const int n_chunks = unk_geom(n).chunks;
  if (!t->h_unk_cnt)
    MONO_HIP(hipHostMalloc((void**)&t->h_unk_cnt, cnt_bytes, hipHostMallocDefault));
  int32_t* h_cnt = t->h_unk_cnt;
  MONO_HIP(hipMemcpyAsync(h_cnt, cnt, (size_t)n_chunks * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  MONO_TRY(prep_setup(t, t->prep));  // (its event marks the read-back below)
  MONO_HIP(hipEventRecord(t->prep.ev_verify, st));
  if (t->profile) MONO_HIP(hipEventRecord(t->ev[1], st));
  // known ids: the plain path; it runs while the host waits for the count of unknown ones
  {
    ApplyArgs k = a;
    k.from_keys = 0;
    k.filt = FilterDev{};
    k.skip_slot = skip;
    k.sort_err = sort_error_word();
    MONO_TRY(sort_pairs_u32(slot_in, slot_out, nullptr, idx_out, (size_t)n, end_bit, tmp, tmp32, st));
    k.skeys = slot_out;
    k.sidx = idx_out;
    uint4* heads = nullptr;
    MONO_TRY(queue_setup(&k, q_base, n, t->d.dim, &heads, st));
    if (t->profile) MONO_HIP(hipEventRecord(t->ev[2], st));
    MONO_TRY(launch_apply(t, k, heads, k.qcount + 1, st));
    MONO_HIP(hipGetLastError());
  }
  MONO_HIP(hipEventSynchronize(t->prep.ev_verify));
  int64_t U = 0;
  for (int c = 0; c < n_chunks; ++c) U += h_cnt[c];
  MONO_TRY(gated_unknown(t, a, ids, n, (const uint32_t*)unknown, cnt, nullptr, U, st));
  if (t->profile) MONO_HIP(hipEventRecord(t->ev[3], st));
  return filter_after_batch(filter, st, U);
}

// The gated write path behind a lookup's prepared order (try_prepared with allow_missing): the
// ids the table holds have just been applied through the prepared order — the positions the
// lookup did not find carry kNoSlot there and were skipped — and the U unknown ones are exactly
// those positions: the order's tail (k_unk_from_order), walked by the gated head kernel. No
// read-back of its own: U is the lookup's miss count k_verify's read-back brought.
static int gated_after_prepared(Table* t, void* filter, const ApplyArgs& a, const int64_t* ids,
                                int64_t n, int64_t U, hipStream_t st) {
  if (U > 0) {
    // On the table's own stream, BESIDE the apply kernels of the known ids that try_prepared has
    // just launched on the caller's: a dozen launch-bound kernels over a few per cent of the batch
    // (count, compaction, grouping by id, gated walk) that touch other rows, other scratch and the
    // filter, which the known part never reads. The caller's stream joins before the filter's
    // end-of-batch step.
    Table::Prep& p = t->prep;
    hipStream_t gs = p.stream;
    MONO_HIP(hipStreamWaitEvent(gs, p.ev_verify, 0));  // (the ids are final, the slot record read)
    MONO_HIP(hipStreamWaitEvent(gs, p.ev_ready, 0));   // (the order is complete)
    MONO_TRY(gated_unknown(t, a, ids, n, nullptr, nullptr, p.sidx + (n - U), U, gs, p.grouped_tail));
    MONO_HIP(hipEventRecord(p.ev_gated, gs));
    MONO_HIP(hipStreamWaitEvent(st, p.ev_gated, 0));
  }
  if (t->profile) MONO_HIP(hipEventRecord(t->ev[3], st));
  return filter_after_batch(filter, st, U);
}

// The whole write path. `filter` gates inserts of unknown ids (nullptr / dummy = no gate).
static int write_path(Table* t, void* filter, const int64_t* ids, const float* src,
                      const uint32_t* src_ts, int64_t n, int mode, const float* lrs, int n_lrs,
                      int64_t req_time, int assign_add_semantics, hipStream_t st,
                      int64_t global_step = 0) {
  if (n == 0) return 0;
  MONO_REQUIRE(n < (int64_t)0xFFFFFFF0ll, "batch of %lld ids exceeds the 32-bit index range",
               (long long)n);
  FilterDev fd{};
  if (filter) MONO_TRY(filter_dev_view(filter, &fd, st));
  const bool from_keys = fd.kind != 0;
  t->gated_seen = from_keys;
  // "one lookup, then an update of as many ids" is what makes the next lookup prepare the update
  // order; two prepared orders in a row that could not be used (other ids, ids to insert) switch
  // it off again for a while
  const bool pattern = mode != MODE_IMPORT && t->cache_n == n && t->lookups_since_update == 1;
  t->lookups_since_update = 0;
  if (mode != MODE_IMPORT) t->last_mode_class = mode_class(mode);
  bool tried = false, prepared = false;
  int64_t missing = 0;
  // With a filter the prepared order serves the ids the table holds and the unknown ones are
  // gated afterwards; those may be inserted, so the order is only used while the table cannot
  // need a rehash for this batch (a rehash moves every row the order points at).
  const bool room = !from_keys || t->used_upper + n <= (int64_t)(kMaxLoad * (double)t->d.cap);
  if (room)
    MONO_TRY(try_prepared(t, ids, src, n, mode, lrs, n_lrs, req_time, st, global_step, &tried,
                          &prepared, /*allow_missing=*/from_keys, &missing));
  if (prepared) t->prep_strikes = 0;
  else if (tried) ++t->prep_strikes;
  if (prepared) ++t->path_counts[0];
  else if (tried) ++t->path_counts[1];
  if (++t->updates_since_strike >= 64) t->updates_since_strike = t->prep_strikes = 0;
  t->expect_update = pattern && t->prep_strikes < 2;
  if (prepared && !from_keys) {
    MONO_HIP(hipEventRecord(t->prep.ev_consumed, st));
    t->prep.consumed_set = true;
    return 0;
  }
  // Everything below may insert keys (and move them: rehash): a pending hint is stale, and the
  // next one waits for what this call enqueues. The exception is the gated remainder behind a
  // prepared order: the table has room (no rehash, checked above), an insertion moves no row, and
  // the only thing a queued hint can have wrong afterwards is an id it recorded as absent —
  // which k_hint_fix probes again. Those hints stay.
  const bool keep_hints = prepared && from_keys && keep_hints_enabled();
  if (!keep_hints) MONO_TRY(hint_drop(t, st));
  else t->hints_kept = true;
  struct KeysDone {
    Table* t; hipStream_t st; bool on;
    ~KeysDone() { if (on) (void)keys_changed(t, st); }
  } keys_done{t, st, !keep_hints};
  if (!prepared) MONO_TRY(ensure_room(t, n, st));

  ApplyArgs a{};
  a.skip_slot = kNoSlot;
  a.n = n;
  a.src = src;
  a.src_ts = src_ts;
  a.mode = mode;
  a.from_keys = from_keys;
  a.ts = (uint32_t)req_time;
  a.filt = fd;
  a.assign_add_semantics = assign_add_semantics;
  a.global_step = global_step;
  a.sort_err = from_keys ? nullptr : sort_error_word();
  for (int s = 0; s < kMaxSeg; ++s) a.lr[s] = s < n_lrs ? lrs[s] : 0.f;
  if (prepared) {  // (from_keys) the known ids are applied; the lookup's misses remain
    // (this order is used up: a second update on it would add the miss counts to flags[0] again,
    // and the unknown ids are being inserted now)
    t->prep.n = 0;
    t->used_upper += missing;
    MONO_TRY(gated_after_prepared(t, filter, a, ids, n, missing, st));
    MONO_HIP(hipEventRecord(t->prep.ev_consumed, st));
    t->prep.consumed_set = true;
    return 0;
  }

  const size_t n_al = ((size_t)n + 63) & ~(size_t)63;
  const size_t q_bytes = queue_bytes(n, t->d.dim);
  char* q_base = nullptr;
  if (t->profile) MONO_HIP(hipEventRecord(t->ev[0], st));
  if (!from_keys) {
    // (slot, idx) pairs -> stable radix sort on the bits a slot can occupy
    const int end_bit = bits_for(t->d.cap + 1);
    size_t tmp_bytes = sort_pairs_tmp_bytes((size_t)n, end_bit);
    tmp_bytes = (tmp_bytes + 255) & ~(size_t)255;
    MONO_TRY(t->ws.ensure(4 * n_al * sizeof(uint32_t) + tmp_bytes + q_bytes));
    uint32_t* slot_in = (uint32_t*)t->ws.ptr;
    q_base = (char*)t->ws.ptr + 4 * n_al * sizeof(uint32_t) + tmp_bytes;
    uint32_t* idx_in = slot_in + n_al;
    uint32_t* slot_out = idx_in + n_al;
    uint32_t* idx_out = slot_out + n_al;
    void* tmp = idx_out + n_al;
    if (t->cache_n > 0)
      hipLaunchKernelGGL(k_claim_pending, dim3(grid_for(n, 1024)), dim3(256), 0, st, t->d, ids, n,
                         slot_in, (uint32_t*)q_base, t->cache_ids, t->cache_slots, t->cache_n);
    else
      hipLaunchKernelGGL(k_find_or_claim, dim3(grid_for(n, 32 * kIlp)), dim3(256), 0, st, t->d, ids, n,
                         slot_in, (uint32_t*)q_base);
    MONO_HIP(hipGetLastError());
    if (t->profile) MONO_HIP(hipEventRecord(t->ev[1], st));
    // values = input positions (the sort's identity default): no index array to write and read
    (void)idx_in;
    MONO_TRY(sort_pairs_u32(slot_in, slot_out, nullptr, idx_out, (size_t)n, end_bit, tmp, tmp_bytes,
                            st));
    a.skeys = slot_out;
    a.sidx = idx_out;
  } else {
    return write_path_gated(t, filter, a, ids, n, st);
  }
  uint4* heads = nullptr;
  MONO_TRY(queue_setup(&a, q_base, n, t->d.dim, &heads, st, !from_keys));
  if (t->profile) MONO_HIP(hipEventRecord(t->ev[2], st));
  MONO_TRY(launch_apply(t, a, heads, a.qcount + 1, st));
  MONO_HIP(hipGetLastError());
  if (t->profile) MONO_HIP(hipEventRecord(t->ev[3], st));
  if (from_keys) MONO_TRY(filter_after_batch(filter, st, n));
  return 0;
}

// In-order segmented row sum for the stateless ops (F2 gradient): out[key .. key + D) +=
// src[sidx[p]] for every sorted position p of the segment `key`, left to right. It is the
// table's apply machinery (sub-group kernel + long-segment kernel + chunk combine) pointed at a
// flat float buffer: "slot" = float offset, row stride 1, no optimizer state, no timestamps.
int segment_add_rows(const uint32_t* skeys, const uint32_t* sidx, int64_t n, const float* src,
                     int D, bool aligned4, float* out, hipStream_t st) {
  static Table ctx;  // scratch + aux stream of the stateless ops (one host thread per GPU)
  if (n == 0) return 0;
  MONO_REQUIRE(D > 0 && D <= kTileFloats, "segment_add_rows: dim %d exceeds %d", D, kTileFloats);
  Table* t = &ctx;
  t->d = TableDev{};
  t->d.rows = out;
  t->d.dim = D;
  t->d.row_floats = 1;
  t->d.n_seg = 1;
  t->d.seg[0].dim = D;
  t->d.seg[0].opt_kind = MONO_OPT_SGD;
  t->d.seg[0].st_off = D;
  t->opt_tmpl = OPTK_SGD;
  t->vec = 1;
  if (aligned4) t->vec = D % 32 == 0 ? 4 : D % 16 == 0 ? 2 : 1;
  ApplyArgs a{};
  a.skip_slot = kNoSlot;
  a.skeys = skeys;
  a.sidx = sidx;
  a.n = n;
  a.src = src;
  a.mode = MODE_ASSIGN_ADD;
  a.sort_err = sort_error_word();
  MONO_TRY(t->ws.ensure(queue_bytes(n, D)));
  uint4* heads = nullptr;
  MONO_TRY(queue_setup(&a, (char*)t->ws.ptr, n, D, &heads, st));
  MONO_TRY(launch_apply(t, a, heads, a.qcount + 1, st));
  MONO_HIP(hipGetLastError());
  return 0;
}

// out[key .. key + D) = src_ptrs[first occurrence] + the other occurrences in input order, for
// every segment of the sorted keys whose class has width D (layout gradient: classes = slices).
// Keys equal to skip_key name no destination. Same machinery as segment_add_rows.
int segment_set_sum_rows(const uint32_t* skeys, const uint32_t* sidx, int64_t n,
                         const float* const* src_ptrs, const uint32_t* cls, const int32_t* cls_dim,
                         int D, bool aligned4, uint32_t skip_key, float* out, hipStream_t st,
                         int key_unit) {
  static Table ctx;
  if (n == 0) return 0;
  MONO_REQUIRE(D > 0 && D <= kTileFloats, "segment_set_sum_rows: dim %d exceeds %d", D, kTileFloats);
  Table* t = &ctx;
  t->d = TableDev{};
  t->d.rows = out;
  t->d.dim = D;
  t->d.row_floats = key_unit;
  t->d.n_seg = 1;
  t->d.seg[0].dim = D;
  t->d.seg[0].opt_kind = MONO_OPT_SGD;
  t->d.seg[0].st_off = D;
  t->opt_tmpl = OPTK_SGD;
  t->vec = 1;
  if (aligned4) t->vec = D % 32 == 0 ? 4 : D % 16 == 0 ? 2 : 1;
  ApplyArgs a{};
  a.skeys = skeys;
  a.sidx = sidx;
  a.n = n;
  a.src_ptrs = src_ptrs;
  a.src_aligned4 = aligned4 ? 1 : 0;
  a.cls = cls;
  a.cls_dim = cls_dim;
  a.want_dim = D;
  a.skip_slot = skip_key;
  a.mode = MODE_SET_SUM;
  a.sort_err = sort_error_word();
  MONO_TRY(t->ws.ensure(queue_bytes(n, D)));
  uint4* heads = nullptr;
  MONO_TRY(queue_setup(&a, (char*)t->ws.ptr, n, D, &heads, st));
  MONO_TRY(launch_apply(t, a, heads, a.qcount + 1, st));
  MONO_HIP(hipGetLastError());
  return 0;
}

int segment_set_sum_packed(const uint32_t* skeys, const uint32_t* svals, int64_t n, const SrcTab* tab,
                           int sbits, int bbits, const int32_t* cls_dim, int D, int key_unit,
                           bool aligned4, uint32_t skip_key, float* out, hipStream_t st) {
  static Table ctx;
  if (n == 0) return 0;
  MONO_REQUIRE(D > 0 && D <= kTileFloats, "segment_set_sum_packed: dim %d exceeds %d", D, kTileFloats);
  MONO_REQUIRE(tab && key_unit >= 1 && sbits >= 0 && bbits >= 0 && sbits + bbits <= 32,
               "segment_set_sum_packed: bad packing");
  Table* t = &ctx;
  t->d = TableDev{};
  t->d.rows = out;
  t->d.dim = D;
  t->d.row_floats = key_unit;
  t->d.n_seg = 1;
  t->d.seg[0].dim = D;
  t->d.seg[0].opt_kind = MONO_OPT_SGD;
  t->d.seg[0].st_off = D;
  t->opt_tmpl = OPTK_SGD;
  t->vec = 1;
  if (aligned4) t->vec = D % 32 == 0 ? 4 : D % 16 == 0 ? 2 : 1;
  ApplyArgs a{};
  a.skeys = skeys;
  a.sidx = svals;
  a.n = n;
  a.src_tab = tab;
  a.pk_sbits = sbits;
  a.pk_smask = sbits >= 32 ? 0xFFFFFFFFu : ((1u << sbits) - 1u);
  a.pk_bmask = bbits >= 32 ? 0xFFFFFFFFu : ((1u << bbits) - 1u);
  a.pk_qshift = std::min(sbits + bbits, 31);  // (no seq bits: every seq_stride is 0)
  a.src_aligned4 = aligned4 ? 1 : 0;
  a.cls_dim = cls_dim;
  a.want_dim = D;
  a.skip_slot = skip_key;
  a.max_heads = (int64_t)skip_key;  // keys are < skip_key
  a.mode = MODE_SET_SUM;
  a.sort_err = sort_error_word();
  MONO_TRY(t->ws.ensure(queue_bytes(n, D)));
  uint4* heads = nullptr;
  MONO_TRY(queue_setup(&a, (char*)t->ws.ptr, n, D, &heads, st));
  MONO_TRY(launch_apply(t, a, heads, a.qcount + 1, st));
  MONO_HIP(hipGetLastError());
  return 0;
}

// The same sum with the classification done ahead of time (F2 gradient planned by F1, ops.hip):
// segment_plan_classify runs k_classify on the sorted (float offset, occurrence) pairs into
// `q_base` (segment_plan_bytes(n, D) bytes) on any stream; segment_set_sum_planned later launches
// only the apply kernels, which wait for `ready` themselves. out[key .. key + D) is SET, not added
// to: the caller zero-fills whatever no key covers.
size_t segment_plan_bytes(int64_t n, int D) { return (queue_bytes(n, D) + 255) & ~(size_t)255; }

static Table* segment_ctx(float* out, int D, bool aligned4) {
  static Table ctx;
  Table* t = &ctx;
  t->d = TableDev{};
  t->d.rows = out;
  t->d.dim = D;
  t->d.row_floats = 1;
  t->d.n_seg = 1;
  t->d.seg[0].dim = D;
  t->d.seg[0].opt_kind = MONO_OPT_SGD;
  t->d.seg[0].st_off = D;
  t->opt_tmpl = OPTK_SGD;
  t->vec = 1;
  if (aligned4) t->vec = D % 32 == 0 ? 4 : D % 16 == 0 ? 2 : 1;
  return t;
}

int segment_plan_classify(const uint32_t* skeys, const uint32_t* sidx, int64_t n, int D, char* q_base,
                          hipStream_t st) {
  if (n == 0) return 0;
  MONO_REQUIRE(D > 0 && D <= kTileFloats, "segment_plan_classify: dim %d exceeds %d", D, kTileFloats);
  Table* t = segment_ctx(nullptr, D, false);
  ApplyArgs a{};
  a.skip_slot = kNoSlot;
  a.skeys = skeys;
  a.sidx = sidx;
  a.n = n;
  a.mode = MODE_SET_SUM;
  a.sort_err = sort_error_word();
  uint4* heads = nullptr;
  MONO_TRY(queue_setup(&a, q_base, n, D, &heads, st));
  launch_classify(a, long_allowed(t, a.mode, n), heads, st);
  MONO_HIP(hipGetLastError());
  return 0;
}

int segment_set_sum_planned(const uint32_t* skeys, const uint32_t* sidx, int64_t n, const float* src,
                            int D, bool aligned4, float* out, char* q_base, hipEvent_t ready,
                            hipStream_t st) {
  if (n == 0) return 0;
  Table* t = segment_ctx(out, D, aligned4);
  ApplyArgs a{};
  a.skip_slot = kNoSlot;
  a.skeys = skeys;
  a.sidx = sidx;
  a.n = n;
  a.src = src;
  a.mode = MODE_SET_SUM;
  a.sort_err = sort_error_word();
  uint4* heads = nullptr;
  MONO_TRY(queue_setup(&a, q_base, n, D, &heads, st, /*counters_cleared=*/true));
  // the long queue's hand-out cursor restarts (a plan may be applied more than once)
  double hp = host_prof_on() ? host_now() : 0.0;
  MONO_HIP(hipMemsetAsync(a.qcount + 4, 0, 3 * sizeof(uint32_t), st));
  host_tick("planned: memset cursor", &hp);
  MONO_TRY(launch_apply(t, a, heads, a.qcount + 1, st, /*classified=*/true, ready, /*ready_old=*/true));
  host_tick("planned: launch_apply total", &hp);
  MONO_HIP(hipGetLastError());
  host_tick("planned: get last error", &hp);
  return 0;
}

// Slots of `ids` (kNoSlot for ids the table does not hold), nothing else touched: the resolve
// half of the fused lookup + pool (layout.hip, mono_lookup_to_layout).
__global__ void __launch_bounds__(256) k_resolve(TableDev t, const int64_t* __restrict__ ids, int64_t n,
                                                 uint32_t* __restrict__ slots,
                                                 int64_t* __restrict__ cache_ids = nullptr) {
  const int sub = lane_sub(), gshift = lane_gshift();
  const int64_t ngroups = (int64_t)gridDim.x * (blockDim.x / 8);
  for (int64_t i0 = ((int64_t)blockIdx.x * (blockDim.x / 8) + threadIdx.x / 8) * kIlpP; i0 < n;
       i0 += ngroups * kIlpP) {
    const int64_t mine = i0 + sub < n ? ids[i0 + sub] : kEmpty;
    int64_t key[kIlpP], k0[kIlpP];
    uint64_t b[kIlpP];
#pragma unroll
    for (int q = 0; q < kIlpP; ++q) key[q] = __shfl(mine, q, 8);
#pragma unroll
    for (int q = 0; q < kIlpP; ++q) {
      b[q] = mix64((uint64_t)key[q]) & t.bucket_mask;
      k0[q] = t.keys[b[q] * kBucket + sub];
    }
    uint32_t my_slot = kNoSlot;
#pragma unroll
    for (int q = 0; q < kIlpP; ++q) {
      const uint32_t sl = group_find_pre(t, key[q], b[q], k0[q], sub, gshift);
      if (sub == q) my_slot = sl;
    }
    if (i0 + sub < n) {
      slots[i0 + sub] = my_slot;
      if (cache_ids) cache_ids[i0 + sub] = mine;
    }
  }
}

int table_resolve_slots(void* table, const int64_t* ids, int64_t n, uint32_t* slots, hipStream_t st) {
  MONO_REQUIRE(table, "table_resolve_slots: null table");
  if (n == 0) return 0;
  Table* t = (Table*)table;
  hipLaunchKernelGGL(k_resolve, dim3(grid_for(n, 32 * kIlpP)), dim3(256), 0, st, t->d, ids, n, slots);
  MONO_HIP(hipGetLastError());
  return 0;
}

// mono_table_lookup; out == nullptr: resolve and record the slots only (the caller reads
// t->cache_slots and gathers the rows itself), with the same book-keeping, so that the update
// which follows finds the slot record / the prepared order of a lookup.
static int lookup_impl(Table* t, const int64_t* ids, int64_t n, float* out, hipStream_t st) {
  const bool aligned = ((uintptr_t)out & 15) == 0;
  if (slot_cache_enabled() && n < (int64_t)0xFFFFFFF0ll) {
    if (n > t->cache_cap) {  // grows once per batch size; (re)allocation synchronises
      if (t->cache_ids) (void)hipFree(t->cache_ids);
      if (t->cache_slots) (void)hipFree(t->cache_slots);
      t->cache_ids = nullptr; t->cache_slots = nullptr; t->cache_cap = 0; t->cache_n = 0;
      MONO_HIP(hipMalloc((void**)&t->cache_ids, (size_t)n * sizeof(int64_t)));
      MONO_HIP(hipMalloc((void**)&t->cache_slots, (size_t)n * sizeof(uint32_t)));
      t->cache_cap = n;
    }
    t->cache_n = n;
  } else {
    t->cache_n = 0;
  }
  MONO_REQUIRE(out || t->cache_n, "lookup_impl: a slots-only lookup needs the slot record");
  // a second lookup with no update in between: not the training pattern (any more)
  if (++t->lookups_since_update > 1) t->expect_update = false;
  Table::Hint* oldest = t->hint_count ? &t->hints[t->hint_head] : nullptr;
  if (t->cache_n && oldest && oldest->p.n == n && oldest->cap >= n) {
    // A hint for a lookup of this size: its slot record and update order take the place of the
    // current ones (the buffers swap roles), the ids are checked against the hinted ones on the
    // device (k_hint_fix corrects the record where they differ and counts: the order is then not
    // used), and all that is left of the lookup is the row gather, at full depth.
    std::swap(t->prep, oldest->p);
    std::swap(t->cache_ids, oldest->ids);
    std::swap(t->cache_slots, oldest->slots);
    std::swap(t->cache_cap, oldest->cap);
    oldest->p.n = 0;
    t->hint_head = (t->hint_head + 1) % Table::kHintDepth;
    --t->hint_count;
    ++t->path_counts[2];
    Table::Prep& p = t->prep;
    // (the whole chain, not just the probe: k_hint_fix may rewrite slots the sort is reading. In
    // steady state the chain finished long ago, beside the update before this lookup.)
    MONO_HIP(hipStreamWaitEvent(st, p.ev_ready, 0));
    p.chain_pending = false;
    // (under churn a few per cent of the positions are probed again: one round per wave then,
    // instead of six; 18-25 us per call on config 5 when the hint was right)
    hipLaunchKernelGGL(k_hint_fix, dim3(t->hints_kept ? grid_for(n, 256, 8192) : grid_for(n, 1024, 1024)),
                       dim3(256), 0, st, t->d, ids, n,
                       t->cache_ids, t->cache_slots, p.d_flags);
    MONO_HIP(hipGetLastError());
    if (out) MONO_TRY(launch_gather(t, t->cache_slots, n, out, aligned, 4096, 8, st));
    return 0;
  }
  if (t->hint_count) MONO_TRY(hint_drop(t, st));  // (a lookup of another size: not the hinted one)
  t->prep.n = 0;  // the record it was built from is being overwritten
  if (t->cache_n && prep_mode() != 0 && (t->expect_update || prep_mode() == 2))
    return lookup_prepared(t, t->prep, ids, n, out, aligned, st, /*hint=*/false, t->cache_ids,
                           t->cache_slots);
  if (!out) {
    hipLaunchKernelGGL(k_resolve, dim3(grid_for(n, 32 * kIlpP)), dim3(256), 0, st, t->d, ids, n,
                       t->cache_slots, t->cache_ids);
    MONO_HIP(hipGetLastError());
    return 0;
  }
  int64_t* ci = t->cache_n ? t->cache_ids : nullptr;
  uint32_t* cs = t->cache_n ? t->cache_slots : nullptr;
  if (t->d.dim % 4 == 0 && t->d.row_floats % 4 == 0 && aligned) launch_lookup<4>(t, ids, n, out, st, ci, cs);
  else if (t->d.dim % 2 == 0 && t->d.row_floats % 2 == 0 && aligned) launch_lookup<2>(t, ids, n, out, st, ci, cs);
  else launch_lookup<1>(t, ids, n, out, st, ci, cs);
  MONO_HIP(hipGetLastError());
  return 0;
}

// mono_table_lookup_hint: `ids` (ready on `hs`) are the ids of the mono_table_lookup `ahead` calls
// from now on this table (1 = the next one). No-op unless the calls follow the training pattern
// (lookup, update, lookup, ...).
static int lookup_hint(Table* t, const int64_t* ids, int64_t n, int ahead, hipStream_t hs) {
  if (n == 0 || !slot_cache_enabled() || prep_mode() == 0 || n >= (int64_t)0xFFFFFFF0ll) return 0;
  if (!t->expect_update && prep_mode() != 2) return 0;
  // only a hint that extends the queue by one is taken: the caller can announce "next" and "the
  // one after" at every step without keeping track of what the table dropped or ignored
  if (ahead != t->hint_count + 1 || ahead > Table::kHintDepth) return 0;
  Table::Hint& h = t->hints[(t->hint_head + t->hint_count) % Table::kHintDepth];
  if (n > h.cap) {  // (hipFree synchronises the device: nothing is reading the old buffers)
    if (h.ids) (void)hipFree(h.ids);
    if (h.slots) (void)hipFree(h.slots);
    h.ids = nullptr; h.slots = nullptr; h.cap = 0;
    MONO_HIP(hipMalloc((void**)&h.ids, (size_t)n * sizeof(int64_t)));
    MONO_HIP(hipMalloc((void**)&h.slots, (size_t)n * sizeof(uint32_t)));
    h.cap = n;
  }
  Table::Prep& p = h.p;
  MONO_TRY(prep_setup(t, p));
  if (!t->ev_hint_src) MONO_HIP(hipEventCreateWithFlags(&t->ev_hint_src, hipEventDisableTiming));
  // The probe reads `ids` on the chain stream, some time after this call: they have to stay valid
  // until the lookup they announce has been issued (the caller holds that batch anyway; a copy on
  // the caller's stream would put 13 us into every step: profiles/r02_q_hint_timeline_copy.txt).
  // It also reads the key array: it waits for the last call that may have changed the key set,
  // NOT for updates through the prepared order, which never do.
  MONO_HIP(hipEventRecord(t->ev_hint_src, hs));
  MONO_HIP(hipStreamWaitEvent(p.stream, t->ev_hint_src, 0));
  if (p.consumed_set) MONO_HIP(hipStreamWaitEvent(p.stream, p.ev_consumed, 0));
  if (t->ev_keys_set) MONO_HIP(hipStreamWaitEvent(p.stream, t->ev_keys, 0));
  MONO_TRY(lookup_prepared(t, p, ids, n, nullptr, false, hs, /*hint=*/true, h.ids, h.slots));
  ++t->hint_count;
  return 0;
}

// The slots of `ids` as a LOOKUP of the table: *slots points into the table's slot record (valid
// until the table's next lookup), and an update of the same ids that follows is served like one
// that follows mono_table_lookup. nullptr in *slots: the record is switched off, resolve plainly.
int table_lookup_slots(void* table, const int64_t* ids, int64_t n, const uint32_t** slots,
                       hipStream_t st) {
  MONO_REQUIRE(table && slots, "table_lookup_slots: null argument");
  *slots = nullptr;
  Table* t = (Table*)table;
  if (n == 0 || !slot_cache_enabled() || n >= (int64_t)0xFFFFFFF0ll) return 0;
  MONO_TRY(lookup_impl(t, ids, n, nullptr, st));
  *slots = t->cache_slots;
  return 0;
}

int table_row_view(void* table, const float** rows, int* dim, int* row_floats) {
  MONO_REQUIRE(table, "table_row_view: null table");
  Table* t = (Table*)table;
  *rows = t->d.rows;
  *dim = t->d.dim;
  *row_floats = t->d.row_floats;
  return 0;
}

}  // namespace mono

using namespace mono;

extern "C" {

int mono_table_create(const mono_table_cfg* cfg, void** out_table) {
  MONO_REQUIRE(cfg && out_table, "mono_table_create: null argument");
  MONO_REQUIRE(cfg->n_segments >= 1 && cfg->n_segments <= kMaxSeg,
               "mono_table_create: %d segments (supported: 1..%d)", cfg->n_segments, kMaxSeg);
  {
    int64_t total_dim = 0;
    for (int s = 0; s < cfg->n_segments; ++s) total_dim += cfg->segments[s].dim;
    MONO_REQUIRE(total_dim <= kTileFloats, "mono_table_create: dim %lld exceeds %d",
                 (long long)total_dim, kTileFloats);
  }
  const bool serving = cfg->entry_type == MONO_ENTRY_SERVING;
  MONO_REQUIRE(cfg->entry_type == 0 || cfg->entry_type == MONO_ENTRY_TRAINING || serving,
               "mono_table_create: unknown entry type %d", cfg->entry_type);
  Table* t = new Table();
  int w = 0;
  for (int s = 0; s < cfg->n_segments; ++s) {
    mono_segment_cfg c = cfg->segments[s];
    if (serving) {  // entry_accessor.cc:315-318: a serving entry is its compressor, nothing else
      c.opt_kind = MONO_OPT_SGD;
      c.init_kind = MONO_INIT_ZEROS;
      if (c.comp_kind < MONO_COMP_FP32 || c.comp_kind > MONO_COMP_ONE_BIT) {
        delete t;
        return fail("Unknown tpye of float compressor. %d", c.comp_kind);  // float_compressor.cc:195
      }
    }
    if (c.dim <= 0 || c.opt_kind < 0 || c.opt_kind > MONO_OPT_BATCH_SOFTMAX || c.init_kind < 0 ||
        c.init_kind > MONO_INIT_UNIFORM ||
        (c.opt_kind == MONO_OPT_BATCH_SOFTMAX && c.dim != 1)) {  // DCHECK_EQ(dim_size, 1)
      delete t;
      return fail("mono_table_create: bad segment %d (dim %d opt %d init %d)", s, c.dim,
                  c.opt_kind, c.init_kind);
    }
    t->segs.push_back(c);
    w += c.dim;
  }
  TableDev& d = t->d;
  d.dim = w;
  d.n_seg = cfg->n_segments;
  int st = w, wo = 0;
  d.serving = serving ? 1 : 0;
  for (int s = 0; s < cfg->n_segments; ++s) {
    const mono_segment_cfg& c = t->segs[s];
    SegDev& sd = d.seg[s];
    sd.dim = c.dim; sd.opt_kind = c.opt_kind; sd.w_off = wo; sd.st_off = st;
    sd.init_kind = c.init_kind; sd.init_acc = c.init_acc; sd.wd = c.weight_decay;
    sd.l1 = c.l1; sd.l2 = c.l2; sd.beta = c.beta; sd.init_a = c.init_a; sd.init_b = c.init_b;
    sd.momentum = c.momentum; sd.epsilon = c.epsilon; sd.rho = c.averaging_ratio;
    sd.conf_lr = c.learning_rate; sd.nesterov = c.use_nesterov;
    sd.beta1 = c.beta1; sd.beta2 = c.beta2;
    sd.sr_post = (!serving && c.stochastic_rounding_float16) ? 1 : 0;
    sd.sr_pre = -1;
    for (int z = s - 1; z >= 0; --z)
      if (d.seg[z].sr_post) sd.sr_pre = z;  // ends at the FIRST decorated segment before s
    if (sd.sr_post) d.any_sr = 1;
    sd.comp_kind = serving ? c.comp_kind : MONO_COMP_FP32;
    sd.cq0 = sd.cq1 = 0.f;
    if (sd.comp_kind == MONO_COMP_FIXED_R8) {  // fake_quantizer.h:28-29: step = r / 128, half step
      sd.cq0 = c.comp_a / 128;
      sd.cq1 = sd.cq0 / 2;
    } else if (sd.comp_kind == MONO_COMP_ONE_BIT) {
      sd.cq0 = c.comp_a;
      sd.cq1 = c.comp_b;
    }
    wo += c.dim;
    st += opt_state_floats(c.opt_kind, c.dim);
  }
  d.row_floats = st;
  d.seed = cfg->seed;
  if (cfg->enable_feature_eviction) t->evict_every_sec = (int64_t)cfg->feature_evict_every_n_hours * 3600;
  d.default_expire_days = cfg->default_expire_days;
  // the three reference-config optimizers have specialised kernels; the rest (and mixes) take
  // the generic per-element path
  t->opt_tmpl = !serving && !d.any_sr && cfg->n_segments == 1 &&
                        cfg->segments[0].opt_kind <= MONO_OPT_FTRL
                    ? cfg->segments[0].opt_kind : OPTK_MULTI;
  // floats per lane: the widest access that still keeps all 8 lanes of a sub-group busy
  t->vec = 1;
  if (cfg->n_segments == 1) {
    if (w % 32 == 0) t->vec = 4;
    else if (w % 16 == 0) t->vec = 2;
    else if (w % 8 == 0) t->vec = 1;
    else t->vec = w % 4 == 0 ? 4 : w % 2 == 0 ? 2 : 1;
  }
  if (cfg->n_slot_expire > 0) {
    std::vector<int32_t> tbl(1 << 15, cfg->default_expire_days);
    for (int i = 0; i < cfg->n_slot_expire; ++i)
      if (cfg->slot_ids[i] >= 0 && cfg->slot_ids[i] < (1 << 15))
        tbl[cfg->slot_ids[i]] = cfg->slot_expire_days[i];
    if (hipMalloc((void**)&t->slot_expire_dev, tbl.size() * sizeof(int32_t)) != hipSuccess ||
        hipMemcpy(t->slot_expire_dev, tbl.data(), tbl.size() * sizeof(int32_t),
                  hipMemcpyHostToDevice) != hipSuccess) {
      delete t;
      return fail("mono_table_create: cannot allocate the slot-expire table");
    }
    d.slot_expire = t->slot_expire_dev;
  }
  int64_t want_rows = cfg->initial_capacity > 0 ? cfg->initial_capacity : 1;
  uint64_t cap = pow2_at_least((uint64_t)((double)want_rows / (kMaxLoad * 0.95)) + 1);
  int rc = alloc_arrays(&d, cap);
  if (rc == 0) rc = reset_arrays(&d, nullptr);
  if (rc == 0 && hipStreamSynchronize(nullptr) != hipSuccess) rc = fail("mono_table_create: sync failed");
  if (rc != 0) {
    free_arrays(&d);
    if (t->slot_expire_dev) (void)hipFree(t->slot_expire_dev);
    delete t;
    return rc;
  }
  *out_table = t;
  return 0;
}

int mono_table_destroy(void* table) {
  if (!table) return 0;
  Table* t = (Table*)table;
  free_arrays(&t->d);
  if (t->slot_expire_dev) (void)hipFree(t->slot_expire_dev);
  if (t->cache_ids) (void)hipFree(t->cache_ids);
  if (t->cache_slots) (void)hipFree(t->cache_slots);
  prep_destroy(t);
  t->ws.release();
  t->ws_gated.release();
  t->ws_pack.release();
  t->ws_pack_gated.release();
  if (t->h_unk_cnt) (void)hipHostFree(t->h_unk_cnt);
  for (int i = 0; i < 4; ++i)
    if (t->ev[i]) (void)hipEventDestroy(t->ev[i]);
  for (int i = 0; i < 2; ++i)
    if (t->ev_k[i]) (void)hipEventDestroy(t->ev_k[i]);
  if (t->ev_fork) (void)hipEventDestroy(t->ev_fork);
  if (t->ev_join) (void)hipEventDestroy(t->ev_join);
  if (t->ev_pack) (void)hipEventDestroy(t->ev_pack);
  if (t->aux) (void)hipStreamDestroy(t->aux);
  delete t;
  return 0;
}

int mono_table_profile(void* table, int enable) {
  MONO_REQUIRE(table, "mono_table_profile: null table");
  Table* t = (Table*)table;
  if (enable && !t->ev[0])
    for (int i = 0; i < 4; ++i) MONO_HIP(hipEventCreate(&t->ev[i]));
  if (enable && !t->ev_k[0])
    for (int i = 0; i < 2; ++i) MONO_HIP(hipEventCreate(&t->ev_k[i]));
  t->profile = enable != 0;
  t->ev_k_valid = false;
  return 0;
}

int mono_table_last_phase_ms(void* table, float* h_ms) {
  MONO_REQUIRE(table && h_ms, "mono_table_last_phase_ms: null argument");
  Table* t = (Table*)table;
  MONO_REQUIRE(t->profile && t->ev[0], "mono_table_last_phase_ms: profiling is off");
  MONO_HIP(hipEventSynchronize(t->ev[3]));
  for (int i = 0; i < 3; ++i) MONO_HIP(hipEventElapsedTime(&h_ms[i], t->ev[i], t->ev[i + 1]));
  return 0;
}

int mono_table_last_apply_kernel_ms(void* table, float* h_ms) {
  MONO_REQUIRE(table && h_ms, "mono_table_last_apply_kernel_ms: null argument");
  Table* t = (Table*)table;
  MONO_REQUIRE(t->profile && t->ev_k_valid,
               "mono_table_last_apply_kernel_ms: no timed k_apply launch (profiling off, or no "
               "write call since it was switched on forked k_apply: 96 ids or fewer, a filter)");
  MONO_HIP(hipEventSynchronize(t->ev_k[1]));
  MONO_HIP(hipEventElapsedTime(h_ms, t->ev_k[0], t->ev_k[1]));
  return 0;
}

int mono_table_path_counts(void* table, int64_t* h_counts) {
  MONO_REQUIRE(table && h_counts, "mono_table_path_counts: null argument");
  Table* t = (Table*)table;
  for (int k = 0; k < 4; ++k) h_counts[k] = t->path_counts[k];
  return 0;
}

int mono_table_dim(void* table) { return ((Table*)table)->d.dim; }
int mono_table_slice_size(void* table) { return ((Table*)table)->d.n_seg; }
int mono_table_row_floats(void* table) { return ((Table*)table)->d.row_floats; }

int mono_table_lookup(void* table, const int64_t* ids, int64_t n, float* out, void* stream) {
  MONO_REQUIRE(table, "mono_table_lookup: null table");
  if (n == 0) return 0;
  MONO_REQUIRE(ids && out, "mono_table_lookup: null buffer");
  return lookup_impl((Table*)table, ids, n, out, (hipStream_t)stream);
}

int mono_table_lookup_hint(void* table, const int64_t* ids, int64_t n, int ahead, void* stream) {
  MONO_REQUIRE(table && (n == 0 || ids), "mono_table_lookup_hint: null argument");
  MONO_REQUIRE(ahead >= 1, "mono_table_lookup_hint: ahead = %d (1 = the next lookup)", ahead);
  return lookup_hint((Table*)table, ids, n, ahead, (hipStream_t)stream);
}

// embedding_hash_table_tf_bridge.cc:69-100. The bridge's thread looks every 10 s; the first look
// only notes the time, later ones evict once feature_evict_every_n_hours have passed since the last
// eviction (or since that first look), with the largest req_time seen so far. Here the look happens
// at the entry of a write call — between two calls, where the thread's would fall as well.
static int periodic_evict(Table* t, void* stream) {
  if (t->evict_every_sec < 0) return 0;
  const int64_t now = t->test_clock >= 0 ? t->test_clock : (int64_t)time(nullptr);
  if (t->last_evict_wall == 0) {
    t->last_evict_wall = now;
    return 0;
  }
  if (now - t->last_evict_wall < t->evict_every_sec) return 0;
  t->last_evict_wall = now;
  ++t->periodic_evictions;
  return mono_table_evict(t, t->max_update_ts, stream);
}

int mono_table_set_clock(void* table, int64_t now_sec, int64_t* h_evictions) {
  MONO_REQUIRE(table, "mono_table_set_clock: null table");
  Table* t = (Table*)table;
  t->test_clock = now_sec >= 0 ? now_sec : -1;
  if (h_evictions) *h_evictions = t->periodic_evictions;
  return 0;
}

int mono_table_assign(void* table, void* filter, const int64_t* ids, const float* values, int64_t n,
                      int64_t req_time, void* stream) {
  MONO_REQUIRE(table, "mono_table_assign: null table");
  Table* t = (Table*)table;
  MONO_TRY(periodic_evict(t, stream));
  t->max_update_ts = std::max(t->max_update_ts, req_time);
  return write_path(t, filter, ids, values, nullptr, n, MODE_ASSIGN, nullptr, 0, req_time, 0,
                    (hipStream_t)stream);
}

int mono_table_assign_add(void* table, void* filter, const int64_t* ids, const float* values,
                          int64_t n, int64_t req_time, void* stream) {
  MONO_REQUIRE(table, "mono_table_assign_add: null table");
  Table* t = (Table*)table;
  MONO_TRY(periodic_evict(t, stream));
  if (n > 0) t->max_update_ts = std::max(t->max_update_ts, req_time);
  return write_path(t, filter, ids, values, nullptr, n, MODE_ASSIGN_ADD, nullptr, 0, req_time, 1,
                    (hipStream_t)stream);
}

int mono_table_apply_gradients(void* table, void* filter, const int64_t* ids, const float* grads,
                               int64_t n, const float* h_learning_rates, int n_learning_rates,
                               int64_t req_time, int64_t global_step, int enable_dedup,
                               void* stream) {
  MONO_REQUIRE(table, "mono_table_apply_gradients: null table");
  Table* t = (Table*)table;
  MONO_REQUIRE(!t->d.serving, "ServingEntryAccessor doesn't support Optimize");  // entry_accessor.cc:79-83
  for (int s = 0; s < t->d.n_seg; ++s)  // batch_softmax_optimizer.cc:49-52 (LOG(FATAL) there)
    MONO_REQUIRE(t->d.seg[s].opt_kind != MONO_OPT_BATCH_SOFTMAX || global_step >= 0,
                 "global_step=%lld is negative, please investigate!", (long long)global_step);
  MONO_REQUIRE(n_learning_rates == t->d.n_seg,
               "mono_table_apply_gradients: %d learning rates for %d segments", n_learning_rates,
               t->d.n_seg);
  MONO_TRY(periodic_evict(t, stream));
  t->max_update_ts = std::max(t->max_update_ts, req_time);
  return write_path(t, filter, ids, grads, nullptr, n, enable_dedup ? MODE_OPT_DEDUP : MODE_OPT_SEQ,
                    h_learning_rates, n_learning_rates, req_time, 0, (hipStream_t)stream,
                    global_step);
}

int mono_table_import(void* table, const int64_t* ids, const float* rows, const uint32_t* ts,
                      int64_t n, void* stream) {
  MONO_REQUIRE(table, "mono_table_import: null table");
  Table* t = (Table*)table;
  if (n == 0) return 0;
  // max_update_ts follows the restored timestamps (bridge :344-367)
  std::vector<uint32_t> h(n);
  MONO_HIP(hipMemcpyAsync(h.data(), ts, n * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  MONO_HIP(hipStreamSynchronize((hipStream_t)stream));
  for (uint32_t v : h) t->max_update_ts = std::max<int64_t>(t->max_update_ts, v);
  return write_path(t, nullptr, ids, rows, ts, n, MODE_IMPORT, nullptr, 0, 0, 0, (hipStream_t)stream);
}

int mono_table_size(void* table, int64_t* h_size, void* stream) {
  MONO_REQUIRE(table && h_size, "mono_table_size: null argument");
  int64_t live, tombs;
  MONO_TRY(read_counters((Table*)table, (hipStream_t)stream, &live, &tombs));
  *h_size = live;
  return 0;
}

int mono_table_contains(void* table, const int64_t* ids, int64_t n, uint8_t* out, void* stream) {
  MONO_REQUIRE(table, "mono_table_contains: null table");
  if (n == 0) return 0;
  Table* t = (Table*)table;
  hipLaunchKernelGGL(k_contains, dim3(grid_for(n, 32)), dim3(256), 0, (hipStream_t)stream, t->d, ids,
                     n, out);
  MONO_HIP(hipGetLastError());
  return 0;
}

int mono_table_lookup_entry(void* table, const int64_t* ids, int64_t n, float* rows, uint32_t* ts,
                            uint8_t* found, void* stream) {
  MONO_REQUIRE(table && (n == 0 || (ids && rows && ts && found)), "mono_table_lookup_entry: null argument");
  if (n == 0) return 0;
  Table* t = (Table*)table;
  hipLaunchKernelGGL(k_lookup_entry, dim3(grid_for(n, 32)), dim3(256), 0, (hipStream_t)stream, t->d,
                     ids, n, rows, ts, found);
  MONO_HIP(hipGetLastError());
  return 0;
}

int mono_table_clear(void* table, void* stream) {
  MONO_REQUIRE(table, "mono_table_clear: null table");
  Table* t = (Table*)table;
  MONO_TRY(hint_drop(t, (hipStream_t)stream));
  MONO_TRY(reset_arrays(&t->d, (hipStream_t)stream));
  t->used_upper = 0;
  t->cache_n = 0;
  t->prep.n = 0;
  return keys_changed(t, (hipStream_t)stream);
}

int mono_table_evict(void* table, int64_t max_update_time, void* stream) {
  MONO_REQUIRE(table, "mono_table_evict: null table");
  Table* t = (Table*)table;
  t->cache_n = 0;  // evicted slots may be claimed by other keys
  t->prep.n = 0;
  MONO_TRY(hint_drop(t, (hipStream_t)stream));
  hipLaunchKernelGGL(k_evict, dim3(grid_for((int64_t)t->d.cap + 2, 256)), dim3(256), 0,
                     (hipStream_t)stream, t->d, max_update_time);
  MONO_HIP(hipGetLastError());
  return keys_changed(t, (hipStream_t)stream);
}

int mono_table_max_update_ts(void* table, int64_t* h_out) {
  MONO_REQUIRE(table && h_out, "mono_table_max_update_ts: null argument");
  *h_out = ((Table*)table)->max_update_ts;
  return 0;
}

int mono_table_reserve(void* table, int64_t rows, void* stream) {
  MONO_REQUIRE(table, "mono_table_reserve: null table");
  Table* t = (Table*)table;
  hipStream_t st = (hipStream_t)stream;
  int64_t live, tombs;
  MONO_TRY(read_counters(t, st, &live, &tombs));
  const int64_t limit = (int64_t)(kMaxLoad * 0.95 * (double)t->d.cap);
  if (std::max(rows, live) + tombs > limit) {
    uint64_t want = pow2_at_least((uint64_t)((double)std::max(rows, live) / (kMaxLoad * 0.95)) + 1);
    MONO_TRY(rehash(t, std::max<uint64_t>(want, 1024), st));
    tombs = 0;
  }
  t->used_upper = live + tombs;
  return 0;
}

int mono_table_export(void* table, int64_t max_rows, int apply_expiry, int64_t* ids, float* rows,
                      uint32_t* ts, int64_t* h_count, void* stream) {
  MONO_REQUIRE(table && h_count, "mono_table_export: null argument");
  Table* t = (Table*)table;
  hipStream_t st = (hipStream_t)stream;
  MONO_HIP(hipMemsetAsync(&t->d.counters[2], 0, sizeof(unsigned long long), st));
  hipLaunchKernelGGL(k_export, dim3(grid_for((int64_t)t->d.cap + 2, 256)), dim3(256), 0, st, t->d,
                     max_rows, apply_expiry, t->max_update_ts, ids, rows, ts);
  MONO_HIP(hipGetLastError());
  unsigned long long cnt = 0;
  MONO_HIP(hipMemcpyAsync(&cnt, &t->d.counters[2], sizeof(cnt), hipMemcpyDeviceToHost, st));
  MONO_HIP(hipStreamSynchronize(st));
  *h_count = std::min<int64_t>((int64_t)cnt, max_rows);
  return 0;
}

int mono_tables_fused_lookup(void* const* tables, int K, const int64_t* ids,
                             const int32_t* h_fused_slot_size, int N, float* out_emb,
                             int32_t* h_emb_splits, int32_t* h_id_offsets, int32_t* h_emb_offsets,
                             int32_t* h_emb_sizes, void* stream) {
  MONO_REQUIRE(tables && K > 0 && N > 0 && h_fused_slot_size, "mono_tables_fused_lookup: bad argument");
  MONO_REQUIRE(h_emb_splits && h_id_offsets && h_emb_offsets && h_emb_sizes,
               "mono_tables_fused_lookup: null metadata output");
  // ops/hash_table_lookup_op.cc:178-203
  const int cnt = K * N;
  int64_t output_dim = 0, prev = 0;
  for (int i = 0; i < cnt; ++i) h_id_offsets[i] = h_emb_offsets[i] = h_emb_sizes[i] = 0;
  for (int shard = 0; shard < N; ++shard) {
    for (int k = 0; k < K; ++k) {
      int idx = K * shard + k;
      MONO_REQUIRE(h_fused_slot_size[idx] >= 0, "mono_tables_fused_lookup: negative slot size");
      int64_t seg_dim = (int64_t)((Table*)tables[k])->d.dim * h_fused_slot_size[idx];
      output_dim += seg_dim;
      MONO_REQUIRE(output_dim < (int64_t)INT32_MAX, "mono_tables_fused_lookup: int32 offset overflow");
      h_emb_sizes[idx] = (int32_t)seg_dim;
      if (idx > 0) {
        h_emb_offsets[idx] = h_emb_offsets[idx - 1] + h_emb_sizes[idx - 1];
        h_id_offsets[idx] = h_id_offsets[idx - 1] + h_fused_slot_size[idx - 1];
      }
    }
    h_emb_splits[shard] = (int32_t)(output_dim - prev);
    prev = output_dim;
  }
  if (!out_emb) return 0;
  if (K == 1) {  // one table: the shard-major cells are one contiguous id list
    const int64_t total = (int64_t)h_id_offsets[cnt - 1] + h_fused_slot_size[cnt - 1];
    if (total == 0) return 0;
    return mono_table_lookup(tables[0], ids, total, out_emb, stream);
  }
  for (int idx = 0; idx < cnt; ++idx) {
    if (h_fused_slot_size[idx] == 0) continue;
    MONO_TRY(mono_table_lookup(tables[idx % K], ids + h_id_offsets[idx], h_fused_slot_size[idx],
                               out_emb + h_emb_offsets[idx], stream));
  }
  return 0;
}

int mono_tables_fused_lookup_hint(void* const* tables, int K, const int64_t* ids,
                                  const int32_t* h_fused_slot_size, int N, int ahead, void* stream) {
  MONO_REQUIRE(tables && K > 0 && N > 0 && h_fused_slot_size,
               "mono_tables_fused_lookup_hint: bad argument");
  // One table: the fused lookup is ONE mono_table_lookup over the shard-major id list, and that
  // is what is announced. K > 1 runs one lookup per (shard, table) cell — up to N per table and
  // step, more than a hint queue holds: not announced (the lookups resolve their ids themselves).
  if (K != 1) return 0;
  int64_t total = 0;
  for (int idx = 0; idx < N; ++idx) {
    MONO_REQUIRE(h_fused_slot_size[idx] >= 0, "mono_tables_fused_lookup_hint: negative slot size");
    total += h_fused_slot_size[idx];
  }
  if (total == 0) return 0;
  return mono_table_lookup_hint(tables[0], ids, total, ahead, stream);
}

int mono_tables_fused_apply_gradient(void* const* tables, void* const* filters, int K,
                                     const int64_t* ids, const int32_t* h_fused_slot_size,
                                     const float* grads, const int32_t* h_id_offsets,
                                     const int32_t* h_grad_offsets, const float* h_learning_rates,
                                     const int32_t* h_lr_lengths, int64_t req_time,
                                     int64_t global_step, int N, int enable_grad_accumulation,
                                     void* stream) {
  MONO_REQUIRE(tables && K > 0 && N > 0, "mono_tables_fused_apply_gradient: bad argument");
  // ops/hash_table_update_op.cc:209-251; shard-major order is the canonical one
  std::vector<int> lr_off(K, 0);
  for (int k = 1; k < K; ++k) lr_off[k] = lr_off[k - 1] + h_lr_lengths[k - 1];
  bool gated = false;  // a real (non-dummy) filter gates every cell against the table state it finds
  if (filters && filters[0]) {
    FilterDev fd{};
    MONO_TRY(filter_dev_view(filters[0], &fd, (hipStream_t)stream));
    gated = fd.kind != 0;
  }
  if (K == 1 && !enable_grad_accumulation && !gated) {
    // One table, no per-cell accumulation: "every cell in shard-major order, every occurrence a
    // separate step" is one sequential pass over the concatenated id list, provided the cells
    // are laid out back to back (they are when the offsets come from mono_tables_fused_lookup).
    Table* t0 = (Table*)tables[0];
    const int64_t D = t0->d.dim;
    bool contiguous = true;
    int64_t total = 0;
    for (int idx = 0; idx < N; ++idx) {
      if (h_fused_slot_size[idx] > 0 &&
          (h_id_offsets[idx] != total || (int64_t)h_grad_offsets[idx] != total * D))
        contiguous = false;
      total += h_fused_slot_size[idx];
    }
    if (contiguous) {
      t0->max_update_ts = std::max(t0->max_update_ts, req_time);
      if (total == 0) return 0;
      return mono_table_apply_gradients(tables[0], filters ? filters[0] : nullptr, ids, grads,
                                        total, h_learning_rates, h_lr_lengths[0], req_time,
                                        global_step, 0, stream);
    }
  }
  for (int shard = 0; shard < N; ++shard)
    for (int k = 0; k < K; ++k) {
      int idx = shard * K + k;
      Table* tk = (Table*)tables[k];
      tk->max_update_ts = std::max(tk->max_update_ts, req_time);  // bridge :233-234
      if (h_fused_slot_size[idx] == 0) continue;
      MONO_TRY(mono_table_apply_gradients(tables[k], filters ? filters[k] : nullptr,
                                          ids + h_id_offsets[idx], grads + h_grad_offsets[idx],
                                          h_fused_slot_size[idx], h_learning_rates + lr_off[k],
                                          h_lr_lengths[k], req_time, global_step,
                                          enable_grad_accumulation, stream));
    }
  return 0;
}

}  // extern "C"
