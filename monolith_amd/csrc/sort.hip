// sort.hip — stable LSD radix sort of (uint32 key, uint32 value) pairs for gfx950.
//
// Every grouping step of the hot path (occurrences by table slot, by fused-embedding offset, by
// shard * M + slot, by pooling row) is a stable sort of at most a few million 8-byte pairs on
// 1..29 key bits. At that size a general-purpose library sort is launch-bound (buffer fills and
// look-back state between passes). Two paths, both 8-bit digits over 8192-element tiles:
//
// up to 1024 tiles (8.39 M pairs: every sort of the benchmarked workloads) — pass 0 as below (its
// hist kernel also zeroes the scratch of the later passes, its scatter kernel accumulates pass
// 1's digit totals), then ONE kernel per pass:
//   k_sort_fused    ranks its tile, publishes the tile's digit counts as self-validating words
//                   (count | ready bit: no fence, an agent-scope fence here writes back a whole
//                   XCD L2), sums the counts of the tiles before it (tile numbers come from an
//                   atomic ticket, so those tiles are already running and never wait before they
//                   publish), accumulates the NEXT pass's digit totals, scatters
// beyond that — three small kernels per pass, no spinning:
//   k_sort_hist     per-tile digit histogram (LDS)
//   k_sort_scan     per digit: running count over the preceding tiles, and the digit total
//   k_sort_scatter  stable in-tile ranking with wave ballots, tile staged in digit order in LDS,
//                   then every digit's run written as contiguous (coalesced) stores
//
// A tile is 8192 consecutive elements owned by one 512-thread workgroup; wave w owns elements
// [1024 w, 1024 w + 1024) and walks them 64 at a time, so "rank among equal digits" is:
// equal-digit lanes below me in this round (ballot match) + what my wave counted in earlier
// rounds (LDS counter per wave and digit) + what lower waves counted (scan over waves). 8-bit
// digits keep the runs a tile contributes to one digit at ~32 elements (128 B): with 10-bit
// digits the scatter degenerates into 16-byte partial-line writes and runs 2x slower.
#include <algorithm>
#include <cstdlib>

#include "mono_common.h"

namespace mono {

constexpr int kSortThreads = 512;
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kSortItems = 16;
constexpr int kSortTile = kSortThreads * kSortItems;  // 8192 pairs = 64 KB of LDS staging
constexpr int kSortMaxBits = 8;
constexpr int kSortMaxDigits = 1 << kSortMaxBits;

__global__ void __launch_bounds__(kSortThreads) k_sort_hist(const uint32_t* __restrict__ keys,
                                                            int64_t n, int shift, int bits,
                                                            uint32_t* __restrict__ tile_hist,
                                                            uint32_t* __restrict__ zero_base,
                                                            int64_t zero_words) {
  __shared__ uint32_t h[kSortMaxDigits];
  const int nd = 1 << bits;
  // the fused passes that follow need their scratch zeroed: done here, one slice per workgroup,
  // instead of a memset between the kernels
  for (int64_t i = (int64_t)blockIdx.x * kSortThreads + threadIdx.x; i < zero_words;
       i += (int64_t)gridDim.x * kSortThreads)
    zero_base[i] = 0u;
  for (int d = threadIdx.x; d < nd; d += kSortThreads) h[d] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortTile;
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const int64_t i = base + r * kSortThreads + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & (uint32_t)(nd - 1)], 1u);
  }
  __syncthreads();
  uint32_t* out = tile_hist + (size_t)blockIdx.x * nd;
  for (int d = threadIdx.x; d < nd; d += kSortThreads) out[d] = h[d];
}

// One workgroup per 32 digits: 32 tile groups x 32 digits. In place: tile_hist[t][d] becomes the
// number of elements with digit d in the tiles before t; dtot[d] = all elements with digit d.
__global__ void __launch_bounds__(1024) k_sort_scan(uint32_t* __restrict__ tile_hist,
                                                    uint32_t* __restrict__ dtot, int tiles,
                                                    int bits) {
  __shared__ uint32_t part[32][33];
  const int nd = 1 << bits;
  const int dl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + dl;
  const int per = (tiles + 31) / 32;
  const int t0 = min(tiles, g * per), t1 = min(tiles, t0 + per);
  uint32_t sum = 0;
  if (d < nd)
    for (int t = t0; t < t1; ++t) sum += tile_hist[(size_t)t * nd + d];
  part[g][dl] = sum;
  __syncthreads();
  if (d < nd) {
    uint32_t run = 0;
    for (int gg = 0; gg < g; ++gg) run += part[gg][dl];
    for (int t = t0; t < t1; ++t) {
      const uint32_t c = tile_hist[(size_t)t * nd + d];
      tile_hist[(size_t)t * nd + d] = run;
      run += c;
    }
    if (g == 31) dtot[d] = run;
  }
}

// Exclusive scan of v[0 .. 256) by one wave (4 values per lane).
__device__ __forceinline__ void wave_excl_scan_256(const uint32_t* in, uint32_t* out, int nd,
                                                   int lane) {
  uint32_t a[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = lane * 4 + k;
    a[k] = d < nd ? in[d] : 0u;
    s += a[k];
  }
  uint32_t inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  uint32_t run = inc - s;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = lane * 4 + k;
    if (d < nd) out[d] = run;
    run += a[k];
  }
}

// Stable rank of a wave's 16 x 64 elements among equal digits of the wave's span. Per round the
// lanes with equal digits find each other through LDS: every lane ORs its lane bit into the
// (wave, digit) mask word and reads the word back (one wave's LDS operations execute in order, and
// OR commutes, so the result does not depend on how the hardware serialises the lanes); the lowest
// peer then adds the group to the running (wave, digit) count and clears the mask for the next
// round. (Eight ballots per element with per-lane 64-bit mask arithmetic cost 6 us of a 13 us
// pass.) mask: this wave's kSortMaxDigits words, zero on entry and on return.
__device__ __forceinline__ void rank_wave_span(const uint32_t (&key)[kSortItems], int64_t base,
                                               int64_t n, int shift, uint32_t dmask, int lane,
                                               uint32_t* cnt_w, unsigned long long* mask_w,
                                               uint32_t (&rank)[kSortItems]) {
  const unsigned long long bitl = 1ull << lane, lt = bitl - 1ull;
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const bool valid = base + r * 64 + lane < n;
    const uint32_t d = (key[r] >> shift) & dmask;
    if (valid) atomicOr(&mask_w[d], bitl);
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      const unsigned long long peers = mask_w[d];
      const uint32_t prev = cnt_w[d];  // all peers read both words before the leader updates them
      rank[r] = prev + (uint32_t)__popcll(peers & lt);
      if ((peers & lt) == 0) {
        cnt_w[d] = prev + (uint32_t)__popcll(peers);
        mask_w[d] = 0ull;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void __launch_bounds__(kSortThreads) k_sort_scatter(
    const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin, int64_t n, int shift,
    int bits, const uint32_t* __restrict__ tile_off, const uint32_t* __restrict__ dtot,
    uint32_t* __restrict__ kout, uint32_t* __restrict__ vout, int next_shift, int next_bits,
    uint32_t* dhist_next) {
  __shared__ uint32_t nh[kSortMaxDigits];  // next pass's digit histogram of this tile (optional)
  __shared__ uint32_t cnt[kSortWaves][kSortMaxDigits];
  __shared__ uint32_t ltot[kSortMaxDigits];   // elements of this tile per digit
  __shared__ uint32_t lexc[kSortMaxDigits];   // ... exclusive scan = start of the digit's run
  __shared__ uint32_t gpre[kSortMaxDigits];   // global start of the digit
  // staging of the tile in digit order; before that (ranking) its first 16 KB hold the match masks
  __shared__ __align__(8) uint32_t stage[2 * kSortTile];
  uint32_t* const sk = stage;
  uint32_t* const sv = stage + kSortTile;
  unsigned long long* const match = reinterpret_cast<unsigned long long*>(stage);
  const int nd = 1 << bits;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < kSortWaves * kSortMaxDigits; i += kSortThreads) {
    (&cnt[0][0])[i] = 0;
    match[i] = 0ull;
  }
  for (int i = threadIdx.x; i < kSortMaxDigits; i += kSortThreads) nh[i] = 0;
  __syncthreads();
  const int64_t tile0 = (int64_t)blockIdx.x * kSortTile;
  const int64_t base = tile0 + (int64_t)w * (64 * kSortItems);
  uint32_t key[kSortItems], val[kSortItems], rank[kSortItems];
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const int64_t i = base + r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? kin[i] : 0u;
    val[r] = valid ? (vin ? vin[i] : (uint32_t)i) : 0u;
  }
  // stable rank inside the wave's 1024-element span
  rank_wave_span(key, base, n, shift, (uint32_t)(nd - 1), lane, cnt[w],
                 match + (size_t)w * kSortMaxDigits, rank);
  if (next_bits > 0) {
#pragma unroll
    for (int r = 0; r < kSortItems; ++r)
      if (base + r * 64 + lane < n)
        atomicAdd(&nh[(key[r] >> next_shift) & (uint32_t)((1 << next_bits) - 1)], 1u);
  }
  __syncthreads();
  if (next_bits > 0)
    for (int i = threadIdx.x; i < (1 << next_bits); i += kSortThreads)
      if (nh[i]) atomicAdd(&dhist_next[i], nh[i]);
  // per digit: start of every wave's share inside the digit's run, and the tile total
  for (int d = threadIdx.x; d < nd; d += kSortThreads) {
    uint32_t run = 0;
#pragma unroll
    for (int ww = 0; ww < kSortWaves; ++ww) {
      const uint32_t c = cnt[ww][d];
      cnt[ww][d] = run;
      run += c;
    }
    ltot[d] = run;
  }
  __syncthreads();
  if (w == 0) wave_excl_scan_256(ltot, lexc, nd, lane);
  if (w == 1) wave_excl_scan_256(dtot, gpre, nd, lane);
  __syncthreads();
  // stage the tile in digit order
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    if (base + r * 64 + lane < n) {
      const uint32_t d = (key[r] >> shift) & (uint32_t)(nd - 1);
      const uint32_t lp = lexc[d] + cnt[w][d] + rank[r];
      sk[lp] = key[r];
      sv[lp] = val[r];
    }
  }
  __syncthreads();
  // runs of equal digits go out as contiguous, coalesced stores
  const uint32_t* off = tile_off + (size_t)blockIdx.x * nd;
  const int count = (int)min((int64_t)kSortTile, n - tile0);
  for (int j = threadIdx.x; j < count; j += kSortThreads) {
    const uint32_t k = sk[j];
    const uint32_t d = (k >> shift) & (uint32_t)(nd - 1);
    const uint32_t pos = gpre[d] + off[d] + ((uint32_t)j - lexc[d]);
    kout[pos] = k;
    vout[pos] = sv[j];
  }
}


// ------------------------------------------------------------------------------ fused passes
// For up to kFusedMaxTiles tiles one kernel per pass replaces hist + scan + scatter ("onesweep"
// without the chained scan): the digit totals of a pass are accumulated by the pass before it;
// inside a pass every workgroup ranks its tile, publishes the tile's digit counts
// and a ready flag, waits for the flags of the tiles before it, and sums their counts itself.
// Tile numbers are handed out by an atomic ticket, so every tile a workgroup waits for has
// already started and publishes before it waits on anything: the wait always ends. The spin is
// bounded all the same; a time-out raises err[0] (mono_sort_error) instead of hanging the GPU.
constexpr int kFusedMaxTiles = 1024;  // 8.4 M pairs
// Beyond this many tiles the three plain kernels of a pass are cheaper than the fused one (800
// tiles, config 3's 6.5 M uses: 8 + 10 + 44 us against 64-82 us — the look-back chain grows with
// the tile count, the plain scatter does not).
constexpr int kFusedPreferTiles = 512;
constexpr int kFusedMaxPasses = 4;
constexpr uint32_t kSpinLimit = 1u << 22;
constexpr uint32_t kRowReady = 0x80000000u;  // a tile holds at most 8192 elements

constexpr int kLbGroup = 16;                                // tiles per look-back group
constexpr int kLbMaxGroups = kFusedMaxTiles / kLbGroup;     // 32

// two published words (count | kRowReady each) per 8-byte relaxed load; a word validates itself
__device__ __forceinline__ unsigned long long lb_load(const uint32_t* row, int l2) {
  return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(row) + l2, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long lb_wait(unsigned long long v, const uint32_t* row,
                                                      int l2, uint32_t* err, int* timed_out) {
  const unsigned long long both = ((unsigned long long)kRowReady << 32) | kRowReady;
  uint32_t spins = 0;
  while ((v & both) != both) {
    __builtin_amdgcn_s_sleep(2);
    if (++spins > kSpinLimit) {
      atomicExch(err, 1u);
      *timed_out = 1;
      break;
    }
    v = lb_load(row, l2);
  }
  return v;
}

struct FusedPass {
  int shift[kFusedMaxPasses];
  int bits[kFusedMaxPasses];
  int passes;
};

__global__ void __launch_bounds__(kSortThreads) k_sort_fused(
    const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin, int64_t n, int shift,
    int bits, int tiles, const uint32_t* __restrict__ dhist, uint32_t* rows, uint32_t* grows,
    int next_shift,
    int next_bits, uint32_t* dhist_next, uint32_t* ticket, uint32_t* err,
    uint32_t* __restrict__ kout, uint32_t* __restrict__ vout) {
  __shared__ uint32_t nh[kSortMaxDigits];  // this tile's histogram of the NEXT pass's digit
  __shared__ uint32_t cnt[kSortWaves][kSortMaxDigits];
  __shared__ uint32_t ltot[kSortMaxDigits];
  __shared__ uint32_t lexc[kSortMaxDigits];
  __shared__ uint32_t gpre[kSortMaxDigits];
  __shared__ uint32_t tpre[4][kSortMaxDigits];  // counts of the preceding tiles (four thread sets)
  // staging of the tile in digit order; before that (ranking) its first 16 KB hold the match masks
  __shared__ __align__(8) uint32_t stage[2 * kSortTile];
  uint32_t* const sk = stage;
  uint32_t* const sv = stage + kSortTile;
  unsigned long long* const match = reinterpret_cast<unsigned long long*>(stage);
  __shared__ uint32_t s_tile;
  __shared__ int s_timed_out;
  const int nd = 1 << bits;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) {
    s_tile = atomicAdd(ticket, 1u);
    s_timed_out = 0;
  }
  for (int i = threadIdx.x; i < kSortWaves * kSortMaxDigits; i += kSortThreads) {
    (&cnt[0][0])[i] = 0;
    match[i] = 0ull;
  }
  for (int i = threadIdx.x; i < kSortMaxDigits; i += kSortThreads) nh[i] = 0;
  __syncthreads();
  const int tile = (int)s_tile;
  if (tile >= tiles) {  // a ticket word that did not start at zero: never index past the rows
    if (threadIdx.x == 0) atomicExch(err, 2u);
    return;
  }
  const int64_t tile0 = (int64_t)tile * kSortTile;
  const int64_t base = tile0 + (int64_t)w * (64 * kSortItems);
  uint32_t key[kSortItems], val[kSortItems], rank[kSortItems];
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const int64_t i = base + r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? kin[i] : 0u;
    val[r] = valid ? (vin ? vin[i] : (uint32_t)i) : 0u;
  }
  rank_wave_span(key, base, n, shift, (uint32_t)(nd - 1), lane, cnt[w],
                 match + (size_t)w * kSortMaxDigits, rank);
  __syncthreads();
  for (int d = threadIdx.x; d < nd; d += kSortThreads) {
    uint32_t run = 0;
#pragma unroll
    for (int ww = 0; ww < kSortWaves; ++ww) {
      const uint32_t c = cnt[ww][d];
      cnt[ww][d] = run;
      run += c;
    }
    ltot[d] = run;
    // publish this tile's digit count. Every word validates itself (kRowReady), so no fence is
    // needed: an agent-scope fence on this chip writes back / invalidates a whole XCD L2, which
    // costs far more than the sort itself
    __hip_atomic_store(&rows[(size_t)tile * kSortMaxDigits + d], run | kRowReady, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (w == 0) wave_excl_scan_256(ltot, lexc, nd, lane);
  if (w == 1) wave_excl_scan_256(dhist, gpre, nd, lane);
  // digit totals of the next pass (its kernel reads them): done here, after publishing, so the
  // tiles behind this one are not kept waiting
  if (next_bits > 0) {
#pragma unroll
    for (int r = 0; r < kSortItems; ++r)
      if (base + r * 64 + lane < n)
        atomicAdd(&nh[(key[r] >> next_shift) & (uint32_t)((1 << next_bits) - 1)], 1u);
  }
  // counts of the tiles before this one, in two levels: tiles form groups of kLbGroup; the last
  // tile of a group publishes the group's total (its own look-back inside the group + its count),
  // so a tile reads at most kLbGroup - 1 tile rows and tiles / kLbGroup group rows instead of
  // every tile before it. Smaller tickets are already running and publish before they wait, and
  // a group total waits on tile rows only: every wait ends. Four sets of 128 threads split the
  // rows; a thread reads two digits per 8-byte load, all its loads of a level in flight.
  int timed_out = 0;
  {
    const int set = threadIdx.x >> 7, l2 = threadIdx.x & 127;
    const int g = tile / kLbGroup, g0 = g * kLbGroup;
    const bool act = 2 * l2 < nd;
    uint32_t sum0 = 0, sum1 = 0;
    if (act) {
      unsigned long long v[kLbGroup / 4];
#pragma unroll
      for (int u = 0; u < kLbGroup / 4; ++u) {
        const int t = g0 + set + 4 * u;
        v[u] = t < tile ? lb_load(rows + (size_t)t * kSortMaxDigits, l2) : 0ull;
      }
#pragma unroll
      for (int u = 0; u < kLbGroup / 4; ++u) {
        const int t = g0 + set + 4 * u;
        if (t < tile) {
          v[u] = lb_wait(v[u], rows + (size_t)t * kSortMaxDigits, l2, err, &timed_out);
          sum0 += (uint32_t)v[u] & ~kRowReady;
          sum1 += (uint32_t)(v[u] >> 32) & ~kRowReady;
        }
      }
    }
    tpre[set][2 * l2] = sum0;
    tpre[set][2 * l2 + 1] = sum1;
    if (timed_out) s_timed_out = 1;
    __syncthreads();
    // (a group total built from a wait that gave up is never published: the tiles behind it time
    // out in turn and raise the error word themselves)
    if (tile % kLbGroup == kLbGroup - 1 && !s_timed_out)
      for (int d = threadIdx.x; d < nd; d += kSortThreads)
        __hip_atomic_store(&grows[(size_t)g * kSortMaxDigits + d],
                           (tpre[0][d] + tpre[1][d] + tpre[2][d] + tpre[3][d] + ltot[d]) | kRowReady,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (act) {
      unsigned long long v[kLbMaxGroups / 4];
#pragma unroll
      for (int u = 0; u < kLbMaxGroups / 4; ++u) {
        const int gg = set + 4 * u;
        v[u] = gg < g ? lb_load(grows + (size_t)gg * kSortMaxDigits, l2) : 0ull;
      }
#pragma unroll
      for (int u = 0; u < kLbMaxGroups / 4; ++u) {
        const int gg = set + 4 * u;
        if (gg < g) {
          v[u] = lb_wait(v[u], grows + (size_t)gg * kSortMaxDigits, l2, err, &timed_out);
          sum0 += (uint32_t)v[u] & ~kRowReady;
          sum1 += (uint32_t)(v[u] >> 32) & ~kRowReady;
        }
      }
    }
    tpre[set][2 * l2] = sum0;
    tpre[set][2 * l2 + 1] = sum1;
  }
  if (timed_out) s_timed_out = 1;
  __syncthreads();
  // a wait gave up (error word raised, mono_debug_sort_error): the sums are not to be trusted,
  // so this tile writes nothing rather than somewhere
  if (s_timed_out) return;
  if (next_bits > 0)
    for (int i = threadIdx.x; i < (1 << next_bits); i += kSortThreads)
      if (nh[i]) atomicAdd(&dhist_next[i], nh[i]);
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    if (base + r * 64 + lane < n) {
      const uint32_t d = (key[r] >> shift) & (uint32_t)(nd - 1);
      const uint32_t lp = lexc[d] + cnt[w][d] + rank[r];
      sk[lp] = key[r];
      sv[lp] = val[r];
    }
  }
  __syncthreads();
  const int count = (int)min((int64_t)kSortTile, n - tile0);
  for (int j = threadIdx.x; j < count; j += kSortThreads) {
    const uint32_t k = sk[j];
    const uint32_t d = (k >> shift) & (uint32_t)(nd - 1);
    const uint32_t pos = gpre[d] + tpre[0][d] + tpre[1][d] + tpre[2][d] + tpre[3][d] +
                         ((uint32_t)j - lexc[d]);
    kout[pos] = k;
    vout[pos] = sv[j];
  }
}

// Sticky error word of the fused passes (1 = a look-back wait gave up, 2 = a ticket word was not
// zero): a __device__ symbol, so no call ever allocates it; every GPU holds its own copy, the
// pointer is resolved once per device. Consumers of a sort's output read it before they act on
// that output (table.hip k_classify), and every host read-back of the product path reports it
// (sort_error_check): a failed sort never turns into rows updated at the wrong slots.
static __device__ uint32_t g_sort_err_word[64];
constexpr int kMaxDevices = 64;
static uint32_t* g_sort_err_ptr[kMaxDevices] = {};

uint32_t* sort_error_word() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  if (!g_sort_err_ptr[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_sort_err_word)) != hipSuccess) return nullptr;
    g_sort_err_ptr[dev] = (uint32_t*)p;
  }
  return g_sort_err_ptr[dev];
}

// `h_word` holds the error word as copied to the host by the caller's own read-back.
int sort_error_report(uint32_t h_word) {
  if (h_word == 0) return 0;
  return fail("radix sort failed on the device (code %u: %s); the batch it served was skipped",
              h_word, h_word == 1 ? "a look-back wait timed out" : "a control word was not zero");
}

static bool fused_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("MONO_SORT_FUSED");
    on = e ? atoi(e) : 1;
  }
  return on != 0;
}

size_t sort_pairs_tmp_bytes(size_t n, int end_bit) {
  const size_t n_al = (n + 63) & ~(size_t)63;
  const size_t tiles = (n + kSortTile - 1) / kSortTile;
  const int passes = (std::min(std::max(end_bit, 1), 32) + kSortMaxBits - 1) / kSortMaxBits;
  size_t b = 256 + kSortMaxDigits * sizeof(uint32_t);                       // digit totals
  b += ((tiles * kSortMaxDigits * sizeof(uint32_t)) + 255) & ~(size_t)255;  // tile histograms
  if (passes > 1) b += 2 * n_al * sizeof(uint32_t);                         // ping-pong buffer
  // fused passes: digit histograms, flags, tickets, error word, per-pass tile rows
  if (tiles <= (size_t)kFusedMaxTiles && passes <= kFusedMaxPasses)
    b += 256 + kFusedMaxPasses * (kSortMaxDigits + 16) * sizeof(uint32_t) +
         (size_t)passes * (tiles + kLbMaxGroups) * kSortMaxDigits * sizeof(uint32_t);
  return b;
}

int sort_pairs_u32(const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout,
                   size_t n, int end_bit, void* tmp, size_t tmp_bytes, hipStream_t st) {
  if (n == 0) return 0;
  MONO_REQUIRE(n < 0xFFFFFFF0ull, "sort_pairs_u32: %zu elements exceed the 32-bit index range", n);
  MONO_REQUIRE(tmp && tmp_bytes >= sort_pairs_tmp_bytes(n, end_bit),
               "sort_pairs_u32: scratch too small");
  if (end_bit < 1) end_bit = 1;
  if (end_bit > 32) end_bit = 32;
  const int passes = (end_bit + kSortMaxBits - 1) / kSortMaxBits;
  const int tiles = (int)((n + kSortTile - 1) / kSortTile);
  const size_t n_al = (n + 63) & ~(size_t)63;
  char* p = (char*)tmp;
  uint32_t* dtot = (uint32_t*)p;
  p += 256 + kSortMaxDigits * sizeof(uint32_t);
  uint32_t* thist = (uint32_t*)p;
  p += (((size_t)tiles * kSortMaxDigits * sizeof(uint32_t)) + 255) & ~(size_t)255;
  uint32_t* tk = (uint32_t*)p;
  uint32_t* tv = tk + n_al;
  if (fused_enabled() && passes >= 2 && tiles <= kFusedPreferTiles && passes <= kFusedMaxPasses) {
    uint32_t* const g_sort_err = sort_error_word();
    MONO_REQUIRE(g_sort_err, "sort_pairs_u32: cannot resolve the error word of this device");
    char* q = (char*)(tv + n_al);
    q = (char*)(((uintptr_t)q + 255) & ~(uintptr_t)255);
    uint32_t* dhist = (uint32_t*)q;                                   // [passes][256]
    uint32_t* ctl = dhist + kFusedMaxPasses * kSortMaxDigits;         // tickets [4], err
    uint32_t* rows = ctl + 16 * kFusedMaxPasses;                      // [passes][tiles][256]
    uint32_t* grows = rows + (size_t)passes * tiles * kSortMaxDigits;  // [passes][32][256]
    const int64_t zero_words = (int64_t)(grows - dhist) + (int64_t)passes * kLbMaxGroups * kSortMaxDigits;
    FusedPass fp{};
    fp.passes = passes;
    int sh = 0;
    for (int ps = 0; ps < passes; ++ps) {
      const int left = passes - ps;
      fp.shift[ps] = sh;
      fp.bits[ps] = (end_bit - sh + left - 1) / left;
      sh += fp.bits[ps];
    }
    // pass 0: its tile histograms exist before any tile is ranked, so the plain kernels are the
    // cheapest form; the hist kernel also zeroes the scratch of the fused passes and the scatter
    // kernel accumulates pass 1's digit totals
    const bool out0 = ((passes - 1) & 1) == 0;
    uint32_t* d0k = out0 ? kout : tk;
    uint32_t* d0v = out0 ? vout : tv;
    hipLaunchKernelGGL(k_sort_hist, dim3(tiles), dim3(kSortThreads), 0, st, kin, (int64_t)n,
                       fp.shift[0], fp.bits[0], thist, dhist, zero_words);
    hipLaunchKernelGGL(k_sort_scan, dim3(((1 << fp.bits[0]) + 31) / 32), dim3(1024), 0, st, thist,
                       dtot, tiles, fp.bits[0]);
    hipLaunchKernelGGL(k_sort_scatter, dim3(tiles), dim3(kSortThreads), 0, st, kin, vin, (int64_t)n,
                       fp.shift[0], fp.bits[0], thist, dtot, d0k, d0v, fp.shift[1], fp.bits[1],
                       dhist + kSortMaxDigits);
    const uint32_t* src_k = d0k;
    const uint32_t* src_v = d0v;
    for (int ps = 1; ps < passes; ++ps) {
      const bool to_out = ((passes - 1 - ps) & 1) == 0;
      uint32_t* dst_k = to_out ? kout : tk;
      uint32_t* dst_v = to_out ? vout : tv;
      const bool more = ps + 1 < passes;
      hipLaunchKernelGGL(k_sort_fused, dim3(tiles), dim3(kSortThreads), 0, st, src_k, src_v,
                         (int64_t)n, fp.shift[ps], fp.bits[ps], tiles,
                         dhist + ps * kSortMaxDigits,
                         rows + (size_t)ps * tiles * kSortMaxDigits,
                         grows + (size_t)ps * kLbMaxGroups * kSortMaxDigits,
                         more ? fp.shift[ps + 1] : 0, more ? fp.bits[ps + 1] : 0,
                         dhist + (more ? ps + 1 : ps) * kSortMaxDigits, ctl + ps,
                         g_sort_err, dst_k, dst_v);
      src_k = dst_k;
      src_v = dst_v;
    }
    MONO_HIP(hipGetLastError());
    return 0;
  }
  // even split of the key bits over the passes (29 bits -> 8 + 7 + 7 + 7)
  int shift = 0;
  const uint32_t* src_k = kin;
  const uint32_t* src_v = vin;
  for (int ps = 0; ps < passes; ++ps) {
    const int left = passes - ps;
    const int bits = (end_bit - shift + left - 1) / left;
    // the last pass must land in (kout, vout): passes alternate backwards from there
    const bool to_out = ((passes - 1 - ps) & 1) == 0;
    uint32_t* dst_k = to_out ? kout : tk;
    uint32_t* dst_v = to_out ? vout : tv;
    hipLaunchKernelGGL(k_sort_hist, dim3(tiles), dim3(kSortThreads), 0, st, src_k, (int64_t)n,
                       shift, bits, thist, (uint32_t*)nullptr, (int64_t)0);
    hipLaunchKernelGGL(k_sort_scan, dim3(((1 << bits) + 31) / 32), dim3(1024), 0, st, thist, dtot,
                       tiles, bits);
    hipLaunchKernelGGL(k_sort_scatter, dim3(tiles), dim3(kSortThreads), 0, st, src_k, src_v,
                       (int64_t)n, shift, bits, thist, dtot, dst_k, dst_v, 0, 0,
                       (uint32_t*)nullptr);
    src_k = dst_k;
    src_v = dst_v;
    shift += bits;
  }
  MONO_HIP(hipGetLastError());
  return 0;
}


// ---- stable grouping of 64-bit keys -------------------------------------------------------------
// The gated write path and the hash filter need a batch's ids GROUPED — equal ids next to each
// other, in input order inside a group — not sorted: nobody reads the order of the groups. A
// 64-bit radix sort (eight passes) is therefore replaced by a 32-bit surrogate: every id claims
// a cell of a scratch hash table (a 4-byte cell holds the position of the occurrence that won it;
// the id itself is keys[that position]), all occurrences of an id meet in the same cell, and the
// cell's index is sorted with the 32-bit sort above on bits_for(cells) bits — stable, so a
// group keeps its input order.
__global__ void __launch_bounds__(256) k_group_claim(const int64_t* __restrict__ keys, int64_t n,
                                                     uint32_t* __restrict__ cells, uint32_t mask,
                                                     uint32_t* __restrict__ sur) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t key = keys[i];
    // (not the table's and the filter's own hash: groups that follow each other in cell order would
    // then hit neighbouring buckets and filter cells, and their atomics queue up on one line —
    // the gated walk took 115 us instead of 49)
    uint32_t h = (uint32_t)(mix64((uint64_t)key ^ 0x9E3779B97F4A7C15ull) >> 24) & mask;
    // cells only ever go EMPTY -> position: a stale EMPTY is settled by the CAS, nothing else can
    // be stale. The table is at most half full, so an EMPTY cell ends every probe sequence.
    for (uint32_t probes = 0; probes <= mask; ++probes) {
      uint32_t c = cells[h];
      if (c == 0xFFFFFFFFu) {
        c = atomicCAS(&cells[h], 0xFFFFFFFFu, (uint32_t)i);
        if (c == 0xFFFFFFFFu) break;  // claimed
      }
      if (keys[c] == key) break;
      h = (h + 1) & mask;
    }
    sur[i] = h;
  }
}
__global__ void __launch_bounds__(256) k_group_gather(const int64_t* __restrict__ keys,
                                                      const uint32_t* __restrict__ vin,
                                                      const uint32_t* __restrict__ perm, int64_t n,
                                                      int64_t* __restrict__ key_out,
                                                      uint32_t* __restrict__ val_out) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n;
       j += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t p = perm[j];
    key_out[j] = keys[p];
    val_out[j] = vin ? vin[p] : p;
  }
}
static uint32_t group_cells(size_t n) {
  uint32_t c = 64;
  while ((size_t)c < 2 * n) c <<= 1;
  return c;
}
static int group_bits(uint32_t cells) {
  int b = 1;
  while ((1u << b) < cells) ++b;
  return b;
}
size_t group_keys64_tmp_bytes(size_t n) {
  const size_t n_al = (n + 63) & ~(size_t)63;
  const uint32_t cells = group_cells(n);
  return (size_t)cells * 4 + 3 * n_al * 4 + ((sort_pairs_tmp_bytes(n, group_bits(cells)) + 255) & ~(size_t)255) + 256;
}
int group_keys64(const int64_t* keys, const uint32_t* vin, int64_t* key_out, uint32_t* val_out, size_t n,
                 void* tmp, size_t tmp_bytes, hipStream_t st) {
  if (n == 0) return 0;
  MONO_REQUIRE(n < 0x7FFFFFF0ull, "group_keys64: %zu keys", n);
  MONO_REQUIRE(tmp_bytes >= group_keys64_tmp_bytes(n), "group_keys64: scratch too small");
  const size_t n_al = (n + 63) & ~(size_t)63;
  const uint32_t cells = group_cells(n);
  uint32_t* d_cells = (uint32_t*)tmp;
  uint32_t* sur = d_cells + cells;
  uint32_t* sur_sorted = sur + n_al;
  uint32_t* perm = sur_sorted + n_al;
  void* stmp = perm + n_al;
  const int bits = group_bits(cells);
  const size_t stmp_bytes = sort_pairs_tmp_bytes(n, bits);
  MONO_HIP(hipMemsetAsync(d_cells, 0xFF, (size_t)cells * 4, st));
  hipLaunchKernelGGL(k_group_claim, dim3(grid_for((int64_t)n, 256)), dim3(256), 0, st, keys, (int64_t)n,
                     d_cells, cells - 1, sur);
  MONO_HIP(hipGetLastError());
  MONO_TRY(sort_pairs_u32(sur, sur_sorted, nullptr, perm, n, bits, stmp, stmp_bytes, st));
  hipLaunchKernelGGL(k_group_gather, dim3(grid_for((int64_t)n, 256)), dim3(256), 0, st, keys, vin, perm,
                     (int64_t)n, key_out, val_out);
  MONO_HIP(hipGetLastError());
  return 0;
}

}  // namespace mono

extern "C" int mono_debug_sort_error(void) {
  uint32_t* w = mono::sort_error_word();
  if (!w) return -1;
  uint32_t v = 0;
  if (hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(&v, w, 4, hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  return (int)v;
}

// Test / measurement hook: sort n pairs on the low end_bit key bits (vin NULL = identity values),
// reps times, average milliseconds per sort (HIP events on the stream) in *avg_ms.
static int debug_sort_run(const uint32_t* kin, const uint32_t* vin, int64_t n, int end_bit,
                          uint32_t* kout, uint32_t* vout, int reps, float* avg_ms, hipStream_t st,
                          void* tmp, size_t bytes, hipEvent_t e0, hipEvent_t e1) {
  MONO_HIP(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r)
    MONO_TRY(mono::sort_pairs_u32(kin, kout, vin, vout, (size_t)n, end_bit, tmp, bytes, st));
  MONO_HIP(hipEventRecord(e1, st));
  MONO_HIP(hipStreamSynchronize(st));
  float ms = 0.f;
  MONO_HIP(hipEventElapsedTime(&ms, e0, e1));
  if (avg_ms) *avg_ms = ms / (float)reps;
  return 0;
}

extern "C" int mono_debug_sort_pairs(const uint32_t* kin, const uint32_t* vin, int64_t n,
                                     int end_bit, uint32_t* kout, uint32_t* vout, int reps,
                                     float* avg_ms, void* stream) {
  MONO_REQUIRE(n >= 0 && reps >= 1, "mono_debug_sort_pairs: bad arguments");
  const size_t bytes = mono::sort_pairs_tmp_bytes((size_t)n, end_bit);
  void* tmp = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = 0;
  if (hipMalloc(&tmp, bytes) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
      hipEventCreate(&e1) != hipSuccess)
    rc = mono::fail("mono_debug_sort_pairs: cannot allocate %zu bytes of scratch / events", bytes);
  else
    rc = debug_sort_run(kin, vin, n, end_bit, kout, vout, reps, avg_ms, (hipStream_t)stream, tmp,
                        bytes, e0, e1);
  if (e0) (void)hipEventDestroy(e0);  // whatever happened above, nothing stays allocated
  if (e1) (void)hipEventDestroy(e1);
  if (tmp) (void)hipFree(tmp);
  return rc;
}
