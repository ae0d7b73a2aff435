// layout.hip — fused embedding-to-layout (G1) and its gradient (G2) for MI355X (gfx950).
//
//   forward   ops/fused_embedding_to_layout.cc:403-695
//   gradient  ops/fused_embedding_to_layout.cc:703-976
//
// Forward: one thread per output element (group, sample, column): it walks the sample's FID run
// in order and, for an ADDN layout, the layout's slices in order — the reference's summation
// order exactly, so the result is bit-identical and independent of scheduling. Consecutive
// threads read consecutive columns of the same embedding row (coalesced row gathers); the FID
// descriptor (8 B) is a broadcast within the row.
//
// Gradient: every use of an embedding slice (layout, slice, sample, fid) is a contribution to
// one (unique fid, column range). Contributions are enumerated in the reference's order, keyed
// by the float offset of their destination, stably sorted by key (sort.hip), and each segment is
// summed left to right by one 8-lane sub-group — no float atomics, first touch = copy.
#include <cstring>
#include <algorithm>
#include <vector>

#include "mono_common.h"

namespace mono {

static Workspace g_lws;       // descriptors + contribution arrays
static Workspace g_lws_sort;  // sort scratch

struct SliceDev {
  int start, dim, pooling, max_seq;
  int out_index, out_stride, out_offset, addn, first;
  int shared, f0, feature_num;  // decoded nfl_offset entry of the slice's feature
};
struct GroupDev {  // a unit of forward work: one non-ADDN slice, or all slices of an ADDN layout
  int s0, s1;
  int width;  // floats per sample: dim, or max_seq * dim for FIRSTN
  long long work_base;
};
struct BufDev {
  float* ptr;
  int dim;
  long long floats;
  long long gbase;  // offset of the buffer in the virtual concatenation of all buffers
  // fused lookup + pool (mono_lookup_to_layout): the "buffer" is a table. Row i of the buffer is
  // the table row of slot slots[i] (kNoSlotL = the id is not in the table: a row of zeros, as
  // mono_table_lookup would have written), rows `stride` floats apart. nullptr: a dense buffer.
  const uint32_t* slots;
  int stride;
  long long gbase_u;  // gradient: gbase and dim in units of the destination keys
  int dim_u;
};
constexpr uint32_t kNoSlotL = 0xFFFFFFFFu;

struct LayoutArgs {
  const SliceDev* slices;
  const GroupDev* groups;
  const BufDev* bufs;
  float* const* outs;
  int n_slices, n_groups, n_bufs;
  const uint64_t* fid_offset;
  long long n_fids;
  const int32_t* feature_offset;
  long long n_features;
  int batch;
  long long total_work;
};

__device__ __forceinline__ int fid_count_dev(const LayoutArgs& a, int f) {
  return f < a.n_features - 1 ? a.feature_offset[f + 1] - a.feature_offset[f]
                              : (int)a.n_fids - a.feature_offset[f];
}

// address of columns [start, start + dim) of the row a fid_offset names, or nullptr when the
// descriptor points outside its buffer (corrupt input must not fault the GPU)
__device__ __forceinline__ const float* slice_src(const LayoutArgs& a, uint64_t fo, int start,
                                                  int dim) {
  const uint32_t i1 = (uint32_t)(fo >> 32), i2 = (uint32_t)fo;
  if (i1 >= (uint32_t)a.n_bufs) return nullptr;
  const BufDev b = a.bufs[i1];
  const long long off = (long long)i2 * b.dim + start;
  if (off + dim > b.floats) return nullptr;
  if (b.slots) {
    const uint32_t slot = b.slots[i2];
    return slot == kNoSlotL ? nullptr : b.ptr + (unsigned long long)slot * b.stride + start;
  }
  return b.ptr + off;
}

// V floats of one row, moved as one access (V = 4: 16 bytes; the host checks alignment)
template <int V>
struct Val {
  float x[V];
};
template <int V>
__device__ __forceinline__ Val<V> val_load(const float* p) {
  Val<V> r;
  if (p == nullptr) {
#pragma unroll
    for (int i = 0; i < V; ++i) r.x[i] = 0.f;
  } else if (V == 4) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    r.x[0] = q.x; r.x[1] = q.y; r.x[2 % V] = q.z; r.x[3 % V] = q.w;
  } else {
#pragma unroll
    for (int i = 0; i < V; ++i) r.x[i] = p[i];
  }
  return r;
}
template <int V>
__device__ __forceinline__ void val_store(float* p, const Val<V>& r) {
  if (V == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r.x[0], r.x[1 % V], r.x[2 % V], r.x[3 % V]);
  } else {
#pragma unroll
    for (int i = 0; i < V; ++i) p[i] = r.x[i];
  }
}
// r = v + r, component-wise (the reference's "row + accumulator" order)
template <int V>
__device__ __forceinline__ void val_add_into(Val<V>& r, const Val<V>& v) {
#pragma unroll
  for (int i = 0; i < V; ++i) r.x[i] = v.x[i] + r.x[i];
}

// pooled value of columns [col, col + V) of slice `sc` for feature run f (SUM / MEAN, "copy
// first"). Four descriptors, then four rows, are in flight at a time; the adds stay in FID order.
template <int V>
__device__ __forceinline__ Val<V> pooled(const LayoutArgs& a, const SliceDev& sc, int f, int col,
                                         int* fnum_out) {
  const int fnum = fid_count_dev(a, f);
  *fnum_out = fnum;
  Val<V> acc = val_load<V>(nullptr);
  if (fnum == 0) return acc;
  const int p0 = a.feature_offset[f];
  int q = 0;
  for (; q + 4 <= fnum; q += 4) {
    uint64_t fo[4];
    const float* src[4];
    Val<V> v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) fo[u] = a.fid_offset[p0 + q + u];
#pragma unroll
    for (int u = 0; u < 4; ++u) src[u] = slice_src(a, fo[u], sc.start, sc.dim);
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = val_load<V>(src[u] ? src[u] + col : nullptr);
    if (q == 0) acc = v[0]; else val_add_into<V>(acc, v[0]);
    val_add_into<V>(acc, v[1]);
    val_add_into<V>(acc, v[2]);
    val_add_into<V>(acc, v[3]);
  }
  for (; q < fnum; ++q) {
    const float* src = slice_src(a, a.fid_offset[p0 + q], sc.start, sc.dim);
    const Val<V> v = val_load<V>(src ? src + col : nullptr);
    if (q == 0) acc = v; else val_add_into<V>(acc, v);
  }
  if (sc.pooling == MONO_POOL_MEAN) {
#pragma unroll
    for (int i = 0; i < V; ++i) acc.x[i] = acc.x[i] / fnum;
  }
  return acc;
}

// One thread per V consecutive output columns of one (group, sample). total_work and the
// groups' work_base / width are in units of V floats.
template <int V>
__global__ void __launch_bounds__(256) k_layout_forward(LayoutArgs a) {
  for (long long wi = (long long)blockIdx.x * blockDim.x + threadIdx.x; wi < a.total_work;
       wi += (long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = a.n_groups - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (a.groups[mid].work_base <= wi) lo = mid; else hi = mid - 1;
    }
    const GroupDev g = a.groups[lo];
    const long long local = wi - g.work_base;
    const int b = (int)(local / g.width);
    const int k = (int)(local - (long long)b * g.width) * V;  // first column of this thread
    const SliceDev s0 = a.slices[g.s0];
    float* out = a.outs[s0.out_index] + (long long)b * s0.out_stride + s0.out_offset;
    if (s0.pooling == MONO_POOL_FIRSTN) {  // :657-662: out[seq * dim + col], zero padded
      const int seq = k / s0.dim, col = k - seq * s0.dim;
      const int f = s0.shared ? s0.f0 : s0.f0 + b;
      if (seq < fid_count_dev(a, f)) {
        const float* src = slice_src(a, a.fid_offset[a.feature_offset[f] + seq], s0.start, s0.dim);
        if (src) val_store<V>(out + k, val_load<V>(src + col));
      }
      continue;
    }
    if (!s0.addn) {
      int fnum;
      const Val<V> v = pooled<V>(a, s0, s0.shared ? s0.f0 : s0.f0 + b, k, &fnum);
      if (fnum) val_store<V>(out + k, v);
      continue;
    }
    // ADDN (:553-598): the layout's slices accumulate into the zero-filled output in order
    Val<V> cur = val_load<V>(nullptr);
    for (int si = g.s0; si < g.s1; ++si) {
      const SliceDev sc = a.slices[si];
      if (!sc.feature_num) continue;
      int f = sc.shared ? sc.f0 : sc.f0 + b;
      if (sc.pooling == MONO_POOL_MEAN || sc.shared) {  // pooled into a temporary first
        // The reference's temporary lives across the samples of a slice and an empty run leaves
        // it untouched (:553-583, GatherEmb returns early), so an empty run contributes the
        // pooled value of the nearest earlier non-empty sample. Reproduced.
        while (f > sc.f0 && fid_count_dev(a, f) == 0) --f;
        int fnum;
        const Val<V> t = pooled<V>(a, sc, f, k, &fnum);
        if (sc.first) cur = t; else val_add_into<V>(cur, t);
      } else {  // SUM: every fid is added straight into the output
        const int fnum = fid_count_dev(a, f);
        const int p0 = a.feature_offset[f];
        for (int q = 0; q < fnum; ++q) {
          const float* src = slice_src(a, a.fid_offset[p0 + q], sc.start, sc.dim);
          val_add_into<V>(cur, val_load<V>(src ? src + k : nullptr));
        }
      }
    }
    val_store<V>(out + k, cur);
  }
}

// ---------------------------------------------------------------------------------- gradient
struct GradSliceDev {
  long long cbase;  // first contribution index of the slice
  int p0;           // first fid position of the slice's feature range
  int len0;         // shared: length of the (single) run
  int start_u;      // the slice's first column, in key units
};

struct GradArgs {
  LayoutArgs l;
  const GradSliceDev* gs;
  const float* const* out_grads;
  long long n_contrib;
  uint32_t* key;       // destination, in units of key_unit floats
  uint32_t* val;       // packed form: slice | sample << sbits | seq << (sbits + bbits)
  uint32_t* cslice;    // pointer form (the packing does not fit 32 bits): slice of contribution c
  const float** srcp;  //               ... and its source row; the sort then carries c itself
  uint32_t nokey;  // key of a use that scatters nothing (FIRSTN beyond max_seq); sorts last
  int key_unit;
  int packed, sbits, bbits;
};

// One thread per use; a wave owns kEnumSpan consecutive uses and walks them 64 at a time. Those
// are consecutive fid positions of (nearly always) one slice, so the feature run of a round's
// first use is found once — by a search every lane runs on the same values (scalar loads) in the
// first round, and from the run the previous round ended in afterwards — and each lane walks
// forward from there. (A 14-step search per use, then per wave and round, kept this kernel at
// the cost of a sort pass: it is a chain of dependent loads, not bandwidth.)
constexpr int kEnumRounds = 8;
constexpr int kEnumSpan = 64 * kEnumRounds;
__global__ void __launch_bounds__(256) k_layout_grad_enumerate(GradArgs a) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long span0 = wave * kEnumSpan; span0 < a.n_contrib; span0 += nwaves * kEnumSpan) {
    int carry_slice = -1, carry_f = 0;
    for (int k = 0; k < kEnumRounds; ++k) {
      const long long c = span0 + (long long)k * 64 + lane;
      if (c - lane >= a.n_contrib) break;
      const bool act = c < a.n_contrib;
      const long long cc = act ? c : a.n_contrib - 1;
      int lo = 0, hi = a.l.n_slices - 1;
      while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (a.gs[mid].cbase <= cc) lo = mid; else hi = mid - 1;
      }
      const SliceDev sc = a.l.slices[lo];
      const GradSliceDev gs = a.gs[lo];
      const long long local = cc - gs.cbase;
      int b, seq, p;
      if (sc.shared) {  // every sample scatters to the one shared run (:874-876)
        b = (int)(local / gs.len0);
        seq = (int)(local - (long long)b * gs.len0);
        p = gs.p0 + seq;
        carry_slice = -1;
      } else {
        p = gs.p0 + (int)local;
        // feature run holding fid position p: last f in [f0, f0 + batch) with feature_offset[f] <= p
        const int lo1 = __builtin_amdgcn_readfirstlane(lo);
        const int p1 = __builtin_amdgcn_readfirstlane(p);
        const int f01 = __builtin_amdgcn_readfirstlane(sc.f0);
        int flo = f01, fhi = f01 + a.l.batch - 1;
        if (lo1 == carry_slice) {
          flo = carry_f;  // the first lane's position lies at or behind the last round's last run
        } else {
          while (flo < fhi) {  // the first lane's run (uniform)
            int mid = (flo + fhi + 1) >> 1;
            if (a.l.feature_offset[mid] <= p1) flo = mid; else fhi = mid - 1;
          }
        }
        if (lo == lo1) {  // same slice, p >= p1: forward from the first lane's run
          const int last = sc.f0 + a.l.batch - 1;
          while (flo < last && a.l.feature_offset[flo + 1] <= p) ++flo;
        } else {  // a round that straddles two slices: search on its own
          flo = sc.f0;
          fhi = sc.f0 + a.l.batch - 1;
          while (flo < fhi) {
            int mid = (flo + fhi + 1) >> 1;
            if (a.l.feature_offset[mid] <= p) flo = mid; else fhi = mid - 1;
          }
        }
        b = flo - sc.f0;
        seq = p - a.l.feature_offset[flo];
        // the next round starts where lane 63 ended (only if it was still in the first lane's slice)
        const int lo63 = __builtin_amdgcn_readlane(lo, 63);
        carry_f = __builtin_amdgcn_readlane(flo, 63);
        carry_slice = lo63 == lo1 ? lo1 : -1;
      }
      if (!act) continue;
      uint32_t key = a.nokey;
      const float* src = nullptr;
      const bool firstn = sc.pooling == MONO_POOL_FIRSTN;
      const uint64_t fo = a.l.fid_offset[p];
      const uint32_t i1 = (uint32_t)(fo >> 32), i2 = (uint32_t)fo;
      if (i1 < (uint32_t)a.l.n_bufs && !(firstn && seq >= sc.max_seq)) {
        const BufDev bd = a.l.bufs[i1];
        const long long off = (long long)i2 * bd.dim + sc.start;
        if (off + sc.dim <= bd.floats) {
          key = (uint32_t)(bd.gbase_u + (long long)i2 * bd.dim_u + gs.start_u);
          if (!a.packed)
            src = a.out_grads[sc.out_index] + (long long)b * sc.out_stride + sc.out_offset +
                  (firstn ? seq * sc.dim : 0);
        }
      }
      a.key[c] = key;
      if (a.packed) {
        a.val[c] = key == a.nokey ? 0u
                                  : ((uint32_t)lo | ((uint32_t)b << a.sbits) |
                                     (firstn ? (uint32_t)seq << (a.sbits + a.bbits) : 0u));
      } else {
        a.cslice[c] = (uint32_t)lo;
        a.srcp[c] = src;
      }
    }
  }
}

// dst[i] = idx[i] < 0 ? fill : src[idx[i]] (the few feature_offset entries the host needs)
__global__ void k_pick_i32(const int32_t* __restrict__ src, const int32_t* __restrict__ idx, int n,
                           int32_t fill, int32_t* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = idx[i] < 0 ? fill : src[idx[i]];
}

static int bits_for_l(uint64_t max_value) {
  int b = 1;
  while (b < 64 && (max_value >> b) != 0) ++b;
  return b;
}

struct HostPlan {
  std::vector<SliceDev> slices;
  std::vector<GroupDev> groups;
  std::vector<BufDev> bufs;
  long long total_work = 0;  // forward work items, in units of `vec` floats
  int vec = 1;               // floats per forward thread (4 when every access is 16-byte aligned)
};

static int build_plan(const mono_layout_slice* h_slices, int n_slices, float* const* emb_ptrs,
                      const int32_t* h_emb_dims, const int64_t* h_emb_floats, int n_bufs,
                      const uint32_t* h_nfl, int n_nfl, int64_t n_features, int32_t batch,
                      int n_outs, const int64_t* h_out_floats, HostPlan* hp, const char* who) {
  MONO_REQUIRE(h_slices && n_slices >= 0 && n_bufs >= 0 && batch >= 0 && n_nfl >= 0,
               "%s: bad argument", who);
  hp->slices.resize(n_slices);
  for (int i = 0; i < n_slices; ++i) {
    const mono_layout_slice& s = h_slices[i];
    MONO_REQUIRE(s.feature_idx >= 0 && s.feature_idx < n_nfl, "%s: slice %d names feature %d of %d",
                 who, i, s.feature_idx, n_nfl);
    MONO_REQUIRE(s.dim > 0 && s.start >= 0 && s.out_index >= 0 && s.out_index < n_outs,
                 "%s: slice %d is malformed", who, i);
    MONO_REQUIRE(s.pooling >= MONO_POOL_SUM && s.pooling <= MONO_POOL_FIRSTN, "%s: pooling %d", who,
                 s.pooling);
    MONO_REQUIRE(!(s.addn && s.pooling == MONO_POOL_FIRSTN), "%s: ADDN over FIRSTN is unsupported",
                 who);
    const int width = s.pooling == MONO_POOL_FIRSTN ? s.max_seq * s.dim : s.dim;
    MONO_REQUIRE(batch == 0 || (int64_t)(batch - 1) * s.out_stride + s.out_offset + width <=
                                   h_out_floats[s.out_index],
                 "%s: slice %d overruns output %d", who, i, s.out_index);
    SliceDev& d = hp->slices[i];
    d.start = s.start; d.dim = s.dim; d.pooling = s.pooling; d.max_seq = s.max_seq;
    d.out_index = s.out_index; d.out_stride = s.out_stride; d.out_offset = s.out_offset;
    d.addn = s.addn; d.first = s.first;
    // GetFeatureInfo (:369-383)
    const uint32_t e = h_nfl[s.feature_idx];
    d.shared = (int)(e >> 31);
    d.f0 = (int)(e & 0x7fffffffu);
    d.feature_num = s.feature_idx < n_nfl - 1
                        ? (int)(h_nfl[s.feature_idx + 1] & 0x7fffffffu) - d.f0
                        : (int)n_features - d.f0;
    MONO_REQUIRE(d.feature_num >= 0 && d.f0 + d.feature_num <= n_features,
                 "%s: nfl_offset[%d] is outside feature_offset", who, s.feature_idx);
    MONO_REQUIRE(d.feature_num == 0 || d.shared || d.feature_num >= batch,
                 "%s: feature %d has %d runs for a batch of %d", who, s.feature_idx, d.feature_num,
                 batch);
  }
  long long gb = 0;
  hp->bufs.resize(n_bufs);
  for (int i = 0; i < n_bufs; ++i) {
    hp->bufs[i] = BufDev{emb_ptrs[i], h_emb_dims[i], (long long)h_emb_floats[i], gb, nullptr,
                         h_emb_dims[i], 0, 0};
    gb += h_emb_floats[i];
  }
  return 0;
}

// Groups of forward work (after hp->bufs is final: their alignment decides the vector width).
static int finish_plan(HostPlan* hp, int32_t batch, const char* who, float* const* h_outs = nullptr) {
  const int n_slices = (int)hp->slices.size();
  // 16-byte accesses need every column range, output offset and row start to be a multiple of 4
  // floats and every base pointer 16-byte aligned
  int vec = h_outs ? 4 : 1;
  for (const SliceDev& sl : hp->slices) {
    if ((sl.start | sl.dim | sl.out_offset | sl.out_stride) & 3) vec = 1;
    if (h_outs && (reinterpret_cast<uintptr_t>(h_outs[sl.out_index]) & 15)) vec = 1;
  }
  for (const BufDev& bd : hp->bufs)
    if ((bd.dim & 3) || (bd.stride & 3) || (reinterpret_cast<uintptr_t>(bd.ptr) & 15)) vec = 1;
  hp->vec = vec;
  hp->groups.clear();
  // groups: an ADDN layout is one group, every other slice its own
  long long work = 0;
  for (int i = 0; i < n_slices;) {
    int j = i + 1;
    if (hp->slices[i].addn) {
      while (j < n_slices && hp->slices[j].addn && !hp->slices[j].first) {
        MONO_REQUIRE(hp->slices[j].dim == hp->slices[i].dim &&
                         hp->slices[j].out_index == hp->slices[i].out_index,
                     "%s: ADDN slices of one layout must agree in dim and output", who);
        ++j;
      }
    }
    const SliceDev& s = hp->slices[i];
    bool live = false;
    for (int k = i; k < j; ++k) live = live || hp->slices[k].feature_num > 0;
    if (live) {
      GroupDev g;
      g.s0 = i; g.s1 = j;
      g.width = (s.pooling == MONO_POOL_FIRSTN ? s.max_seq * s.dim : s.dim) / vec;
      g.work_base = work;
      work += (long long)batch * g.width;
      if (g.width > 0) hp->groups.push_back(g);
    }
    i = j;
  }
  hp->total_work = work;
  return 0;
}

// uploads the plan; returns device pointers inside g_lws (256-byte aligned pieces)
struct DevPlan {
  SliceDev* slices;
  GroupDev* groups;
  BufDev* bufs;
  float** outs;
  char* rest;  // first free byte
};

static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// One host image of every small array a launch needs, uploaded with ONE copy (seven separate
// copies of a few hundred bytes cost ~140 us of stream time in front of the gradient's first
// kernel). `more`: further (pointer, bytes) pieces, returned in dp->more in the same order.
struct Piece { const void* ptr; size_t bytes; };
static int upload_plan(const HostPlan& hp, const void* const* h_outs, int n_outs, size_t extra,
                       DevPlan* dp, hipStream_t st, const std::vector<Piece>& more = {},
                       std::vector<char*>* more_dev = nullptr) {
  std::vector<Piece> pieces = {
      {hp.slices.data(), hp.slices.size() * sizeof(SliceDev)},
      {hp.groups.data(), hp.groups.size() * sizeof(GroupDev)},
      {hp.bufs.data(), hp.bufs.size() * sizeof(BufDev)},
      {h_outs, (size_t)n_outs * sizeof(void*)}};
  pieces.insert(pieces.end(), more.begin(), more.end());
  std::vector<size_t> off(pieces.size());
  size_t total = 0;
  for (size_t i = 0; i < pieces.size(); ++i) {
    off[i] = total;
    total += al256(pieces[i].bytes + 1);
  }
  MONO_TRY(g_lws.ensure(total + extra + 256));
  static std::vector<char> image;  // (one host thread per GPU)
  image.assign(total, 0);
  for (size_t i = 0; i < pieces.size(); ++i)
    if (pieces[i].bytes) memcpy(image.data() + off[i], pieces[i].ptr, pieces[i].bytes);
  char* p = (char*)g_lws.ptr;
  // (through the pinned staging ring: the call enqueues and returns, `image` may be reused)
  MONO_TRY(upload_async(p, image.data(), total, st));
  dp->slices = (SliceDev*)(p + off[0]);
  dp->groups = (GroupDev*)(p + off[1]);
  dp->bufs = (BufDev*)(p + off[2]);
  dp->outs = (float**)(p + off[3]);
  if (more_dev) {
    more_dev->clear();
    for (size_t i = 4; i < pieces.size(); ++i) more_dev->push_back(p + off[i]);
  }
  dp->rest = p + total;
  return 0;
}

// (shared by mono_embedding_to_layout and mono_lookup_to_layout)
static int layout_forward(const HostPlan& hp, int n_slices, int n_bufs, const uint64_t* fid_offset,
                               int64_t n_fids, const int32_t* feature_offset, int64_t n_features,
                               int32_t batch, float* const* h_outs, const int64_t* h_out_floats,
                               int n_outs, hipStream_t st) {
  for (int o = 0; o < n_outs; ++o)
    if (h_out_floats[o] > 0)
      MONO_HIP(hipMemsetAsync(h_outs[o], 0, (size_t)h_out_floats[o] * sizeof(float), st));
  if (hp.total_work == 0) return 0;
  DevPlan dp;
  MONO_TRY(upload_plan(hp, (const void* const*)h_outs, n_outs, 0, &dp, st));
  LayoutArgs a{};
  a.slices = dp.slices; a.groups = dp.groups; a.bufs = dp.bufs; a.outs = dp.outs;
  a.n_slices = n_slices; a.n_groups = (int)hp.groups.size(); a.n_bufs = n_bufs;
  a.fid_offset = fid_offset; a.n_fids = n_fids;
  a.feature_offset = feature_offset; a.n_features = n_features;
  a.batch = batch; a.total_work = hp.total_work;
  if (hp.vec == 4)
    hipLaunchKernelGGL(k_layout_forward<4>, dim3(grid_for(hp.total_work, 256)), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(k_layout_forward<1>, dim3(grid_for(hp.total_work, 256)), dim3(256), 0, st, a);
  MONO_HIP(hipGetLastError());
  return 0;
}

}  // namespace mono

using namespace mono;

extern "C" {

int mono_embedding_to_layout(const mono_layout_slice* h_slices, int n_slices,
                             const float* const* h_emb_ptrs, const int32_t* h_emb_dims,
                             const int64_t* h_emb_floats, int n_bufs, const uint64_t* fid_offset,
                             int64_t n_fids, const int32_t* feature_offset, int64_t n_features,
                             const uint32_t* h_nfl_offset, int n_nfl, int32_t batch,
                             float* const* h_outs, const int64_t* h_out_floats, int n_outs,
                             void* stream) {
  hipStream_t st = (hipStream_t)stream;
  MONO_REQUIRE(n_fids < (int64_t)0x7FFFFFF0ll && n_features < (int64_t)0x7FFFFFF0ll,
               "mono_embedding_to_layout: more than 2^31 fids / features");
  HostPlan hp;
  MONO_TRY(build_plan(h_slices, n_slices, const_cast<float* const*>(h_emb_ptrs), h_emb_dims,
                      h_emb_floats, n_bufs, h_nfl_offset, n_nfl, n_features, batch, n_outs,
                      h_out_floats, &hp, "mono_embedding_to_layout"));
  MONO_TRY(finish_plan(&hp, batch, "mono_embedding_to_layout", h_outs));
  return layout_forward(hp, n_slices, n_bufs, fid_offset, n_fids, feature_offset, n_features, batch,
                        h_outs, h_out_floats, n_outs, st);
}

// Fused lookup + pool: the embedding "buffers" are tables and the FID lists that would have been
// looked up into them. Every list is resolved to slots once (4 bytes per unique FID instead of a
// row of the [n_unique, dim] lookup result), and the pooling kernel gathers the table rows
// themselves: hash_table_lookup_op.cc:36-73 composed with fused_embedding_to_layout.cc:624-676.
int mono_lookup_to_layout(const mono_layout_slice* h_slices, int n_slices, void* const* h_tables,
                          const int64_t* const* h_fid_lists, const int64_t* h_fid_counts, int n_bufs,
                          const uint64_t* fid_offset, int64_t n_fids, const int32_t* feature_offset,
                          int64_t n_features, const uint32_t* h_nfl_offset, int n_nfl, int32_t batch,
                          float* const* h_outs, const int64_t* h_out_floats, int n_outs,
                          void* stream) {
  hipStream_t st = (hipStream_t)stream;
  MONO_REQUIRE(n_fids < (int64_t)0x7FFFFFF0ll && n_features < (int64_t)0x7FFFFFF0ll,
               "mono_lookup_to_layout: more than 2^31 fids / features");
  MONO_REQUIRE(n_bufs >= 0 && (n_bufs == 0 || (h_tables && h_fid_lists && h_fid_counts)),
               "mono_lookup_to_layout: null argument");
  // slots of every list, back to back
  static Workspace slot_ws;
  int64_t total = 0;
  for (int i = 0; i < n_bufs; ++i) {
    MONO_REQUIRE(h_tables[i] && h_fid_counts[i] >= 0 && (h_fid_counts[i] == 0 || h_fid_lists[i]),
                 "mono_lookup_to_layout: bad buffer %d", i);
    total += (h_fid_counts[i] + 63) & ~(int64_t)63;
  }
  MONO_TRY(slot_ws.ensure((size_t)total * sizeof(uint32_t) + 256));
  std::vector<float*> ptrs(n_bufs);
  std::vector<int32_t> dims(n_bufs), strides(n_bufs);
  std::vector<int64_t> floats(n_bufs);
  std::vector<const uint32_t*> slots(n_bufs);
  // Lists that follow each other in memory and name the same table (the V2 per-feature sub-ranges
  // of one fid_list tensor) are resolved by ONE launch: eight 9-us launches in a row otherwise.
  int64_t off = 0;
  for (int i = 0; i < n_bufs;) {
    int j = i + 1;
    int64_t run = h_fid_counts[i];
    while (j < n_bufs && h_tables[j] == h_tables[i] && h_fid_lists[j] == h_fid_lists[i] + run) {
      run += h_fid_counts[j];
      ++j;
    }
    // a table named by ONE run is looked up through its slot record: the update of these ids that
    // follows (apply_gradients on the same fid list) then finds a lookup's prepared order
    bool once = true;
    for (int k = 0; k < n_bufs; ++k) once = once && (k >= i && k < j ? true : h_tables[k] != h_tables[i]);
    const uint32_t* sl = nullptr;
    if (once) MONO_TRY(table_lookup_slots(h_tables[i], h_fid_lists[i], run, &sl, st));
    if (!sl) {
      uint32_t* own = (uint32_t*)slot_ws.ptr + off;
      MONO_TRY(table_resolve_slots(h_tables[i], h_fid_lists[i], run, own, st));
      sl = own;
    }
    const float* rows = nullptr;
    int dim = 0, stride = 0;
    MONO_TRY(table_row_view(h_tables[i], &rows, &dim, &stride));
    int64_t at = 0;
    for (int k = i; k < j; ++k) {
      ptrs[k] = const_cast<float*>(rows);
      dims[k] = dim;
      strides[k] = stride;
      floats[k] = h_fid_counts[k] * (int64_t)dim;
      slots[k] = sl + at;
      at += h_fid_counts[k];
    }
    off += (run + 63) & ~(int64_t)63;
    i = j;
  }
  HostPlan hp;
  MONO_TRY(build_plan(h_slices, n_slices, ptrs.data(), dims.data(), floats.data(), n_bufs,
                      h_nfl_offset, n_nfl, n_features, batch, n_outs, h_out_floats, &hp,
                      "mono_lookup_to_layout"));
  for (int i = 0; i < n_bufs; ++i) {
    hp.bufs[i].slots = slots[i];
    hp.bufs[i].stride = strides[i];
  }
  MONO_TRY(finish_plan(&hp, batch, "mono_lookup_to_layout", h_outs));
  return layout_forward(hp, n_slices, n_bufs, fid_offset, n_fids, feature_offset, n_features, batch,
                        h_outs, h_out_floats, n_outs, st);
}

int mono_embedding_to_layout_grad(const mono_layout_slice* h_slices, int n_slices,
                                  float* const* h_emb_grads, const int32_t* h_emb_dims,
                                  const int64_t* h_emb_floats, int n_bufs,
                                  const uint64_t* fid_offset, int64_t n_fids,
                                  const int32_t* feature_offset, int64_t n_features,
                                  const uint32_t* h_nfl_offset, int n_nfl, int32_t batch,
                                  const float* const* h_out_grads, const int64_t* h_out_floats,
                                  int n_outs, int mean_divides, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  MONO_REQUIRE(n_fids < (int64_t)0x7FFFFFF0ll && n_features < (int64_t)0x7FFFFFF0ll,
               "mono_embedding_to_layout_grad: more than 2^31 fids / features");
  MONO_REQUIRE(!mean_divides, "mono_embedding_to_layout_grad: mean_divides is not implemented "
                              "(the reference scatters the undivided gradient)");
  HostPlan hp;
  MONO_TRY(build_plan(h_slices, n_slices, h_emb_grads, h_emb_dims, h_emb_floats, n_bufs,
                      h_nfl_offset, n_nfl, n_features, batch, n_outs, h_out_floats, &hp,
                      "mono_embedding_to_layout_grad"));
  MONO_TRY(finish_plan(&hp, batch, "mono_embedding_to_layout_grad",
                       const_cast<float* const*>(h_out_grads)));
  long long total_floats = 0;
  bool contiguous = true;  // the buffers follow each other in memory (views of one allocation)
  for (int i = 0; i < n_bufs; ++i) {
    if (i + 1 < n_bufs) contiguous = contiguous && h_emb_grads[i + 1] == h_emb_grads[i] + h_emb_floats[i];
    total_floats += h_emb_floats[i];
  }
  if (contiguous && n_bufs > 0 && total_floats > 0) {
    MONO_HIP(hipMemsetAsync(h_emb_grads[0], 0, (size_t)total_floats * sizeof(float), st));
  } else {
    for (int i = 0; i < n_bufs; ++i)
      if (h_emb_floats[i] > 0)
        MONO_HIP(hipMemsetAsync(h_emb_grads[i], 0, (size_t)h_emb_floats[i] * sizeof(float), st));
  }
  MONO_REQUIRE(total_floats < (long long)0xFFFFFFF0ll,
               "mono_embedding_to_layout_grad: gradient buffers exceed 2^32 floats");
  if (n_slices == 0 || batch == 0 || n_fids == 0) return 0;
  // contribution ranges: two feature_offset entries per slice (start of its first run, end of its
  // last), picked on the device and read back in one small copy
  std::vector<int32_t> pick_idx((size_t)2 * n_slices, -1), pick((size_t)2 * n_slices, 0);
  for (int i = 0; i < n_slices; ++i) {
    const SliceDev& s = hp.slices[i];
    if (!s.feature_num) continue;
    pick_idx[2 * i] = s.f0;
    const long long e = s.shared ? (long long)s.f0 + 1 : (long long)s.f0 + batch;  // run_end(e - 1)
    pick_idx[2 * i + 1] = e < n_features ? (int32_t)e : -1;
  }
  static Workspace pick_ws;
  MONO_TRY(pick_ws.ensure((size_t)4 * n_slices * sizeof(int32_t) + 256));
  int32_t* d_pick_idx = (int32_t*)pick_ws.ptr;
  int32_t* d_pick = d_pick_idx + 2 * n_slices;
  MONO_HIP(hipMemcpyAsync(d_pick_idx, pick_idx.data(), pick_idx.size() * sizeof(int32_t),
                          hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_pick_i32, dim3((2 * n_slices + 255) / 256), dim3(256), 0, st, feature_offset,
                     d_pick_idx, 2 * n_slices, (int32_t)n_fids, d_pick);
  MONO_HIP(hipGetLastError());
  MONO_HIP(hipMemcpyAsync(pick.data(), d_pick, pick.size() * sizeof(int32_t), hipMemcpyDeviceToHost,
                          st));
  MONO_HIP(hipStreamSynchronize(st));
  std::vector<GradSliceDev> gs(n_slices);
  long long nc = 0;
  for (int i = 0; i < n_slices; ++i) {
    const SliceDev& s = hp.slices[i];
    gs[i].cbase = nc;
    gs[i].p0 = 0;
    gs[i].len0 = 1;
    if (!s.feature_num) continue;
    gs[i].p0 = pick[2 * i];
    const long long len = (long long)pick[2 * i + 1] - pick[2 * i];
    MONO_REQUIRE(len >= 0, "mono_embedding_to_layout_grad: feature_offset is not ascending");
    if (s.shared) {
      gs[i].len0 = (int)len;
      nc += (long long)batch * gs[i].len0;
      if (gs[i].len0 == 0) gs[i].len0 = 1;
    } else {
      nc += len;
    }
  }
  if (nc == 0) return 0;
  MONO_REQUIRE(nc < (long long)0xFFFFFFF0ll, "mono_embedding_to_layout_grad: too many uses");
  // keys count units of key_unit floats: every destination offset is gbase + row * dim + start
  auto gcd = [](long long x, long long y) {
    while (y) { const long long r = x % y; x = y; y = r; }
    return x;
  };
  long long unit = 0;
  for (const BufDev& bd : hp.bufs)
    if (bd.floats > 0) unit = gcd(unit, bd.dim);
  for (const SliceDev& sl : hp.slices) unit = gcd(unit, sl.start);
  if (unit <= 0) unit = 1;
  // packed source descriptors when slice | sample | seq fit one 32-bit value
  auto nbits = [](long long v) { return v <= 0 ? 0 : bits_for_l((uint64_t)v); };
  const int sbits = nbits(n_slices - 1), bbits = nbits(batch - 1);
  int qbits = 0;
  for (const SliceDev& sl : hp.slices)
    if (sl.pooling == MONO_POOL_FIRSTN) qbits = std::max(qbits, nbits(sl.max_seq - 1));
  static const bool force_ptrs = getenv("MONO_LAYOUT_GRAD_PTRS") != nullptr;  // test switch
  const bool packed = sbits + bbits + qbits <= 32 && !force_ptrs;
  const size_t n_al = ((size_t)nc + 63) & ~(size_t)63;
  const size_t extra = 5 * n_al * sizeof(uint32_t) +
                       (packed ? 0 : n_al * sizeof(void*)) + 1024;
  std::vector<int32_t> h_dims(n_slices);
  std::vector<SrcTab> h_tab(n_slices);
  for (int i = 0; i < n_slices; ++i) {
    const SliceDev& sl = hp.slices[i];
    h_dims[i] = sl.dim;
    h_tab[i].base = h_out_grads[sl.out_index] + sl.out_offset;
    h_tab[i].stride = sl.out_stride;
    h_tab[i].seq_stride = sl.pooling == MONO_POOL_FIRSTN ? sl.dim : 0;
  }
  // destinations in key units (no division on the device)
  for (BufDev& bd : hp.bufs) {
    bd.gbase_u = bd.gbase / unit;
    bd.dim_u = (int)(bd.dim / unit);
  }
  for (int i = 0; i < n_slices; ++i) gs[i].start_u = (int)(hp.slices[i].start / unit);
  DevPlan dp;
  std::vector<char*> more_dev;
  MONO_TRY(upload_plan(hp, (const void* const*)h_out_grads, n_outs, extra, &dp, st,
                       {{gs.data(), (size_t)n_slices * sizeof(GradSliceDev)},
                        {h_dims.data(), (size_t)n_slices * sizeof(int32_t)},
                        {h_tab.data(), (size_t)n_slices * sizeof(SrcTab)}},
                       &more_dev));
  GradSliceDev* d_gs = (GradSliceDev*)more_dev[0];
  int32_t* d_dims = (int32_t*)more_dev[1];
  SrcTab* d_tab = (SrcTab*)more_dev[2];
  char* p = dp.rest;
  const float** srcp = nullptr;
  if (!packed) { srcp = (const float**)p; p += al256(n_al * sizeof(void*)); }
  uint32_t* key = (uint32_t*)p; p += n_al * sizeof(uint32_t);
  uint32_t* val = (uint32_t*)p; p += n_al * sizeof(uint32_t);   // packed values / slices
  uint32_t* skey = (uint32_t*)p; p += n_al * sizeof(uint32_t);
  uint32_t* sval = (uint32_t*)p;
  const long long n_units = (total_floats + unit - 1) / unit;
  GradArgs a{};
  a.l.slices = dp.slices; a.l.groups = dp.groups; a.l.bufs = dp.bufs; a.l.outs = nullptr;
  a.l.n_slices = n_slices; a.l.n_groups = (int)hp.groups.size(); a.l.n_bufs = n_bufs;
  a.l.fid_offset = fid_offset; a.l.n_fids = n_fids;
  a.l.feature_offset = feature_offset; a.l.n_features = n_features;
  a.l.batch = batch;
  a.gs = d_gs;
  a.out_grads = (const float* const*)dp.outs;
  a.n_contrib = nc;
  a.key = key; a.val = val; a.cslice = val; a.srcp = srcp;
  a.nokey = (uint32_t)n_units;
  a.key_unit = (int)unit;
  a.packed = packed ? 1 : 0; a.sbits = sbits; a.bbits = bbits;
  hipLaunchKernelGGL(k_layout_grad_enumerate, dim3(grid_for(nc, 4 * kEnumSpan, 256 * 64)), dim3(256), 0, st,
                     a);
  MONO_HIP(hipGetLastError());
  const int end_bit = bits_for_l((uint64_t)n_units);  // nokey = n_units sorts last
  const size_t tmp = sort_pairs_tmp_bytes((size_t)nc, end_bit);
  MONO_TRY(g_lws_sort.ensure(tmp));
  // packed: the sort carries the descriptors; pointer form: the contribution indices
  MONO_TRY(sort_pairs_u32(key, skey, packed ? val : nullptr, sval, (size_t)nc, end_bit,
                          g_lws_sort.ptr, tmp, st));
  // Every destination's uses are now one run of the sorted keys, in the reference's order: summed
  // by the table's segment machinery (sub-group walk, long-segment kernel, chunk combine), one
  // pass per distinct slice width. (One sub-group walking a run on its own took 64 ms for a
  // config-3 batch: a Zipf head is hundreds of thousands of uses of one FID.) The result is built
  // in one contiguous image of all buffers: the caller's own memory when its buffers follow each
  // other (views of one allocation, the V2 sub-ranges of one tensor), else a scratch copied out.
  static Workspace grad_img;
  float* img = n_bufs > 0 ? h_emb_grads[0] : nullptr;
  if (!contiguous) {
    MONO_TRY(grad_img.ensure((size_t)total_floats * sizeof(float) + 256));
    img = (float*)grad_img.ptr;
    MONO_HIP(hipMemsetAsync(img, 0, (size_t)total_floats * sizeof(float), st));
  }
  std::vector<int> widths;
  for (int i = 0; i < n_slices; ++i)
    if (std::find(widths.begin(), widths.end(), hp.slices[i].dim) == widths.end())
      widths.push_back(hp.slices[i].dim);
  for (int w : widths) {
    if (packed)
      MONO_TRY(segment_set_sum_packed(skey, sval, nc, d_tab, sbits, bbits, d_dims, w, (int)unit,
                                      hp.vec == 4, (uint32_t)n_units, img, st));
    else
      MONO_TRY(segment_set_sum_rows(skey, sval, nc, srcp, val, d_dims, w, hp.vec == 4,
                                    (uint32_t)n_units, img, st, (int)unit));
  }
  if (!contiguous) {
    long long gb = 0;
    for (int i = 0; i < n_bufs; ++i) {
      if (h_emb_floats[i] > 0)
        MONO_HIP(hipMemcpyAsync(h_emb_grads[i], img + gb, (size_t)h_emb_floats[i] * sizeof(float),
                                hipMemcpyDeviceToDevice, st));
      gb += h_emb_floats[i];
    }
  }
  return 0;
}

}  // extern "C"
