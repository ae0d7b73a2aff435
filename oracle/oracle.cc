// oracle.cc — CPU restatement of Monolith's sparse-embedding hot path.
//
// TEST INFRASTRUCTURE ONLY. Nothing under monolith_amd/ may import, link or call this file; it is
// the checker for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
//
// Single-threaded, deliberately plain C++17: std::unordered_map + per-id loops in the same order
// the reference executes them, so that results (including float summation order) are the
// reference's. Each function cites the reference file:line it restates (paths relative to
// monolith/native_training/ of 0x1997/monolith). Pinned by tests/test_oracle_golden.py against
// the reference's own unit-test vectors (tests/golden/*.json) and against vectors generated from
// the reference's dependency-free runtime/hash_table/optimizer/avx_utils.h (oracle/_ref).
//
// Build: make -C oracle   (g++ -O2 -ffp-contract=off: no FMA contraction, so the float math is
// the "baseline" form the reference's tests pin).
//
// Unpinnable parts, restated with a build-defined hash (documented in DESIGN.md): the hash
// filter's start position uses absl::Hash (process-seeded) in the reference (hash_filter.h:191);
// here it is splitmix64's finalizer, the same function the HIP path uses.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <chrono>
#include <memory>
#include <thread>
#include <unordered_map>
#include <limits>
#include <vector>

namespace {

enum { OPT_SGD = 0, OPT_ADAGRAD = 1, OPT_FTRL = 2, OPT_MOMENTUM = 3, OPT_RMSPROP = 4,
       OPT_RMSPROP_V2 = 5, OPT_ADADELTA = 6, OPT_ADAM = 7, OPT_AMSGRAD = 8,
       OPT_MOVING_AVERAGE = 9, OPT_BATCH_SOFTMAX = 10 };
enum { COMP_FP32 = 0, COMP_FP16 = 1, COMP_FIXED_R8 = 2, COMP_ONE_BIT = 3 };
enum { ENTRY_TRAINING = 1, ENTRY_SERVING = 2 };
enum { INIT_ZEROS = 0, INIT_ONES = 1, INIT_CONSTANT = 2, INIT_UNIFORM = 3 };

struct SegCfg {  // == mono_segment_cfg
  int32_t dim;
  int32_t opt_kind;
  float init_acc;
  float weight_decay;
  float l1, l2, beta;
  int32_t init_kind;
  float init_a, init_b;
  float momentum, epsilon, averaging_ratio, learning_rate;
  int32_t use_nesterov;
  float beta1, beta2;
  int32_t comp_kind;  // serving entries: COMP_*
  float comp_a, comp_b;
  int32_t stochastic_rounding_float16;  // OptimizerConfig field 4 (optimizer.proto:213)
};

// floats of optimizer context: SizeBytes() / sizeof(float) of the reference optimizers
inline int state_floats(int kind, int dim) {
  switch (kind) {
    case OPT_SGD: case OPT_MOVING_AVERAGE: return 0;  // moving_average_optimizer.cc:29-31
    case OPT_FTRL: case OPT_ADADELTA: return 2 * dim;
    case OPT_BATCH_SOFTMAX: return 2;  // batch_softmax_optimizer.cc:31: one int64
    case OPT_ADAM: return 2 * dim + 2;     // adam_optimizer.cc:29-31: m | v | beta1^t | beta2^t
    case OPT_AMSGRAD: return 3 * dim + 2;  // amsgrad_optimizer.cc: m | v | vhat | powers
    default: return dim;
  }
}

struct TableCfg {  // == mono_table_cfg
  int32_t n_segments;
  int32_t default_expire_days;
  const SegCfg* segments;
  int64_t initial_capacity;
  int32_t n_slot_expire;
  int32_t entry_type;
  const int32_t* slot_ids;
  const int32_t* slot_expire_days;
  uint64_t seed;
};

inline uint64_t mix64(uint64_t x) {  // splitmix64 finalizer; shared definition with the HIP path
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}

// data/training_instance/cc/reader_util.h:35-37
inline int slot_id_v2(uint64_t fid) { return (fid >> 48) & ((1 << 15) - 1); }

// random_uniform has no pinned stream in the reference (thread_local mt19937,
// random_uniform_initializer.cc:30-37); both sides use this counter-based form.
inline float uniform_init(uint64_t seed, int64_t id, int elem, float lo, float hi) {
  uint64_t h = mix64(seed ^ mix64((uint64_t)id) ^ (0x9e3779b97f4a7c15ULL * (uint64_t)(elem + 1)));
  float u = (float)(h >> 40) * (1.0f / 16777216.0f);  // [0,1)
  return lo + (hi - lo) * u;
}

// ---------------------------------------------------------------- hash filter
// runtime/neo_filter/hash_filter.h:84-197 (HashFilter<uint16_t>)
struct HashFilter16 {
  static constexpr int kCountBit = 4;
  static constexpr uint32_t kMaxCount = 15;
  static constexpr int kMaxStep = 64;  // hash_filter.h:190 (map_ padding)
  uint64_t capacity = 0, total_size = 0, num_elements = 0, failure_count = 0;
  std::vector<uint16_t> map;

  HashFilter16(uint64_t cap, double fill_rate) {
    capacity = cap;
    total_size = (uint64_t)(cap * fill_rate);  // hash_filter.h:90
    map.assign(total_size + kMaxStep, 0);      // :94
  }
  static uint16_t signature(uint64_t fid) {  // :149, sign_mask = 0xFFF (:185-186)
    return (uint16_t)(((fid >> 17) | (fid << 15)) & 0xFFF);
  }
  // :110-126; returns index into map or -1
  int64_t find(uint64_t fid, int max_step) const {
    uint16_t sign = signature(fid);
    int step = 0;
    uint64_t pos = mix64(fid) % total_size;
    do {
      uint16_t v = map[pos];
      if (v == 0 || (v >> kCountBit) == sign) return (int64_t)pos;
      ++pos;
      if (pos == map.size()) pos = 0;
    } while (++step < max_step);
    return -1;
  }
  // HashFilterIterator::add, :38-61
  uint32_t add_at(int64_t pos, uint64_t fid, uint32_t add_count) {
    if (add_count > kMaxCount) add_count = kMaxCount;
    uint16_t& v = map[pos];
    if (v == 0) {
      ++num_elements;
      if (num_elements > capacity) num_elements = capacity;
      v = (uint16_t)((signature(fid) << kCountBit) + add_count);
      return 0;
    }
    uint32_t count = v & kMaxCount;
    if (count + add_count >= kMaxCount)
      v |= kMaxCount;
    else
      v = (uint16_t)(v + add_count);
    return count;
  }
  uint32_t get_at(int64_t pos) const { return pos < 0 ? kMaxCount : (map[pos] & kMaxCount); }
  bool full() const { return num_elements >= capacity - 1; }  // :128
  void clear() {
    std::fill(map.begin(), map.end(), 0);
    num_elements = 0;
    failure_count = 0;
  }
};

// runtime/neo_filter/sliding_hash_filter.{h,cc} + ops/neo_filter_tf_bridge.cc:25-40
struct Filter {
  int kind = 0;  // 0 dummy (dummy_hash_filter.h:30-51: never filters), 1 sliding
  std::vector<int> slot_threshold;  // int[32768], neo_filter_tf_bridge.cc:28-35
  // sliding state
  static constexpr int kMaxForward = 2;  // sliding_hash_filter.h:96
  static constexpr int kMaxStep = 16;    // :98
  int max_backward = 0;
  std::vector<std::unique_ptr<HashFilter16>> blocks;
  size_t head = 0;
  int head_increment = 0;
  uint64_t failure_count = 0;

  size_t prev(size_t i) const { return i == 0 ? blocks.size() - 1 : i - 1; }
  size_t next(size_t i) const { return i == blocks.size() - 1 ? 0 : i + 1; }

  // sliding_hash_filter.cc:115-127
  bool bidir_find(size_t begin, int max_look, uint64_t fid, bool exhaust, bool forward,
                  size_t* blk, int64_t* pos) const {
    size_t index = begin;
    for (int i = 0; i != max_look; ++i) {
      int64_t p = blocks[index]->find(fid, kMaxStep);
      if (p >= 0 && (!exhaust || blocks[index]->map[p] != 0)) {
        *blk = index;
        *pos = p;
        return true;
      }
      index = forward ? next(index) : prev(index);
    }
    return false;
  }

  // sliding_hash_filter.cc:57-89
  uint32_t add(uint64_t fid, uint32_t count) {
    uint32_t old_count = 0;
    size_t cb;
    int64_t cp;
    if (bidir_find(head, kMaxForward, fid, false, true, &cb, &cp)) {
      if (blocks[cb]->map[cp] != 0) return blocks[cb]->add_at(cp, fid, count);
    } else {
      failure_count += 1;
      return HashFilter16::kMaxCount;
    }
    size_t ob;
    int64_t op;
    if (bidir_find(prev(head), std::min(head_increment, max_backward), fid, true, false, &ob,
                   &op)) {
      old_count = blocks[ob]->get_at(op);
      blocks[cb]->add_at(cp, fid, old_count + count);
    } else {
      blocks[cb]->add_at(cp, fid, count);
    }
    if (blocks[head]->full()) {
      head = next(head);
      head_increment += 1;
      blocks[(head + kMaxForward - 1) % blocks.size()]->clear();
    }
    return old_count;
  }

  // sliding_hash_filter.cc:91-113
  uint32_t get(uint64_t fid) const {
    size_t cb;
    int64_t cp;
    if (bidir_find(head, kMaxForward, fid, false, true, &cb, &cp)) {
      if (blocks[cb]->map[cp] != 0) return blocks[cb]->get_at(cp);
    } else {
      return HashFilter16::kMaxCount;
    }
    size_t ob;
    int64_t op;
    if (bidir_find(prev(head), std::min(head_increment, max_backward), fid, true, false, &ob,
                   &op))
      return blocks[ob]->get_at(op);
    return 0;
  }

  int threshold(int64_t fid) const { return slot_threshold[slot_id_v2((uint64_t)fid)]; }

  // neo_filter_tf_bridge.h:36-48 -> sliding_hash_filter.h:47-54 / dummy_hash_filter.h
  bool should_be_filtered(int64_t fid, int64_t count) {
    if (kind == 0) return false;
    int t = threshold(fid);
    if (t <= 0) return false;
    return add((uint64_t)fid, (uint32_t)count) < (uint32_t)t;
  }
};

// ---------------------------------------------------------------- serving compressors
// float -> IEEE binary16 (to nearest, ties to even: what half_float::half(float) of the
// reference's third_party/half_sourceforge_net/half.hpp does with its default HALF_ROUND_STYLE 1;
// checked against that header in tests/test_serving.py) -> float.
inline float fp16_roundtrip(float v) {
  uint32_t x;
  std::memcpy(&x, &v, 4);
  const uint32_t sign = x & 0x80000000u;
  const uint32_t ax = x & 0x7FFFFFFFu;
  uint32_t out;
  if (ax >= 0x7F800000u) {                       // inf / nan
    out = ax > 0x7F800000u ? (ax | 0x00400000u) : ax;
  } else if (ax >= 0x477FF000u) {                // >= 65520: rounds to infinity
    out = 0x7F800000u;
  } else if (ax >= 0x38800000u) {                // normal half: keep 10 mantissa bits
    const uint32_t rem = ax & 0x1FFFu, base = ax & ~0x1FFFu;
    out = base + ((rem > 0x1000u || (rem == 0x1000u && (base & 0x2000u))) ? 0x2000u : 0u);
  } else {                                       // half subnormal: multiples of 2^-24
    float a;
    std::memcpy(&a, &ax, 4);
    const float scaled = a * 16777216.0f;        // exact: a < 2^-14
    float r = std::nearbyint(scaled);            // default mode: ties to even
    a = r * (1.0f / 16777216.0f);
    std::memcpy(&out, &a, 4);
  }
  out |= sign;
  float res;
  std::memcpy(&res, &out, 4);
  return res;
}

// Neighbouring binary16 values of v as floats: the smallest half >= v (up) and the largest
// half <= v (down), i.e. half.hpp's float2half<round_toward_infinity / round_toward_neg_infinity>
// followed by half2float (stochastic_rounding.h:28-31); checked against that header in
// tests/test_serving.py. Built on the to-nearest conversion: that result is one of the two
// neighbours, the other one is the next half on the far side.
inline uint16_t half_bits_of_exact(float h) {  // h is exactly representable in binary16
  uint32_t x;
  std::memcpy(&x, &h, 4);
  const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
  const uint32_t ax = x & 0x7FFFFFFFu;
  if (ax >= 0x7F800000u) return sign | 0x7C00u;
  if (ax >= 0x38800000u) return sign | (uint16_t)(((ax >> 23) - 112) << 10) | (uint16_t)((ax >> 13) & 0x3FF);
  float a;
  std::memcpy(&a, &ax, 4);
  return sign | (uint16_t)(a * 16777216.0f);
}
inline float float_of_half_bits(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31, m = h & 0x3FF;
  float r;
  if (e == 31) {
    const uint32_t x = sign | 0x7F800000u | (m << 13);
    std::memcpy(&r, &x, 4);
    return r;
  }
  if (e == 0) {
    r = (float)m * (1.0f / 16777216.0f);
    uint32_t x;
    std::memcpy(&x, &r, 4);
    x |= sign;
    std::memcpy(&r, &x, 4);
    return r;
  }
  const uint32_t x = sign | ((e + 112) << 23) | (m << 13);
  std::memcpy(&r, &x, 4);
  return r;
}
inline float half_neighbour(float v, bool up) {
  if (std::isnan(v)) return v;
  const float near = fp16_roundtrip(v);
  if (near == v) return near;
  if (up ? near > v : near < v) return near;
  const uint16_t h = half_bits_of_exact(near);
  const bool neg = (h & 0x8000u) != 0;
  uint16_t o;
  if (up) o = neg ? ((h & 0x7FFFu) ? (uint16_t)(h - 1) : (uint16_t)0x0001u) : (uint16_t)(h + 1);
  else o = neg ? (uint16_t)(h + 1) : ((h & 0x7FFFu) ? (uint16_t)(h - 1) : (uint16_t)0x8001u);
  return float_of_half_bits(o);
}
// stochastic_rounding.h:27-37
inline float stochastic_round(float vf, float p) {
  const float vf_up = half_neighbour(vf, true), vf_down = half_neighbour(vf, false);
  if (p <= (vf - vf_down) / (vf_up - vf_down)) return vf_up;
  return vf_down;
}
// The reference draws p from a thread-local multiply-with-carry sequence (stochastic_rounding.h:
// 90-107): which number meets which element depends on threads and call order, so it cannot be
// pinned. Here p is a counter-based hash of (table seed, global_step, position of the occurrence
// in the batch, element, stage), the same on the GPU; the conversion to [0, 1] is the reference's.
inline float sr_uniform(uint64_t seed, int64_t gstep, uint32_t pos, int e, int stage) {
  uint64_t x = mix64(seed ^ 0x5bd1e995u) ^ mix64((uint64_t)gstep + 0x9e3779b97f4a7c15ULL);
  x = mix64(x ^ ((uint64_t)pos << 32 | (uint32_t)e)) ^ mix64((uint64_t)stage + 77);
  const uint32_t u = (uint32_t)(mix64(x) >> 32);
  return static_cast<float>(u) / 4294967296;
}

// Decode(Encode(v)) of float_compressor.cc:63-137 for one value
inline float comp_q(const SegCfg& c, float v) {
  if (c.comp_kind == COMP_FP16) return fp16_roundtrip(v);
  if (c.comp_kind == COMP_FIXED_R8) {  // fake_quantizer.h:28-64
    const float step = c.comp_a / 128, half_step = step / 2;
    if (std::isnan(v)) return 0 * step;
    float f = v;
    if (f >= 0) f += half_step; else f -= half_step;
    const float q = f / step;
    // `int nstep = f / step_`: beyond the int range the x86 conversion the reference is built
    // for returns INT_MIN (the "integer indefinite"); stated explicitly instead of relying on it
    int nstep = (std::fabs(q) < 2147483648.f) ? (int)q : std::numeric_limits<int>::min();
    if (nstep > 127) nstep = 127; else if (nstep < -128) nstep = -128;
    return (int8_t)nstep * step;
  }
  if (c.comp_kind == COMP_ONE_BIT) {  // hash_net_quantizer.h:42-44 with scale = init_scale
    const float y = c.comp_a * std::tanh(c.comp_b * v);
    return static_cast<float>(y > 0 ? 1 : -1) * c.comp_a;
  }
  return v;
}

// ---------------------------------------------------------------- table
struct Entry {
  std::vector<float> row;  // float[dim] | opt ctx per segment (entry_accessor.cc:101-103)
  uint32_t ts = 0;         // entry_defs.h:31-33 (uint32 seconds)
};

struct Table {
  std::vector<SegCfg> segs;
  int dim = 0, row_floats = 0;
  std::vector<int> seg_w_off, seg_st_off;  // offsets into row
  int default_expire_days = 36500;
  std::unordered_map<int, int> slot_expire;
  uint64_t seed = 0;
  bool serving = false;  // entry_accessor.cc:43-98: values pass through the compressor on write
  int64_t max_update_ts = 0;
  std::unordered_map<int64_t, Entry> m;

  void compress(Entry& e) const {
    if (!serving) return;
    for (size_t s = 0; s < segs.size(); ++s)
      for (int i = 0; i < segs[s].dim; ++i)
        e.row[seg_w_off[s] + i] = comp_q(segs[s], e.row[seg_w_off[s] + i]);
  }

  // entry_accessor.cc:139-143 (Init = initializer then optimizer Init)
  void init_entry(int64_t id, Entry* e) const {
    e->row.assign(row_floats, 0.f);
    e->ts = 0;
    for (size_t s = 0; s < segs.size(); ++s) {
      const SegCfg& c = segs[s];
      float* w = e->row.data() + seg_w_off[s];
      float* st = e->row.data() + seg_st_off[s];
      for (int i = 0; i < c.dim; ++i) {
        switch (c.init_kind) {
          case INIT_ZEROS: w[i] = 0.f; break;
          case INIT_ONES: w[i] = 1.f; break;
          case INIT_CONSTANT: w[i] = c.init_a; break;
          default: w[i] = uniform_init(seed, id, seg_w_off[s] + i, c.init_a, c.init_b);
        }
      }
      if (c.opt_kind == OPT_ADAGRAD) {  // adagrad_optimizer.cc:39-44
        for (int i = 0; i < c.dim; ++i) st[i] = c.init_acc;
      } else if (c.opt_kind == OPT_FTRL) {  // ftrl_optimizer.cc:38-45: norm | zero
        for (int i = 0; i < c.dim; ++i) {
          st[i] = c.init_acc;
          st[c.dim + i] = 0.f;
        }
      } else {  // momentum / rmsprop / adadelta Init: every context float = 0
        const int nf = state_floats(c.opt_kind, c.dim);
        for (int i = 0; i < nf; ++i) st[i] = 0.f;
        if (c.opt_kind == OPT_ADAM || c.opt_kind == OPT_AMSGRAD) {  // adam_optimizer.cc:40-51
          st[nf - 2] = c.beta1;
          st[nf - 1] = c.beta2;
        }
      }
    }
  }

  Entry& upsert(int64_t id) {  // cuckoo_embedding_hash_table.cc:284-291
    auto it = m.find(id);
    if (it == m.end()) {
      it = m.emplace(id, Entry()).first;
      init_entry(id, &it->second);
    }
    return it->second;
  }

  // entry_accessor.cc:163-171 -> optimizer_combination.cc:53-63 -> per-segment Optimize
  void optimize(Entry& e, const float* grad, const float* lrs, int64_t global_step,
                uint32_t pos) const {
    for (size_t s = 0; s < segs.size(); ++s) {
      const SegCfg& c = segs[s];
      float* num = e.row.data() + seg_w_off[s];
      float* st = e.row.data() + seg_st_off[s];
      const float* g = grad + seg_w_off[s];
      const float lr = lrs[s];  // learning_rates.subspan(slice) [0]
      if (c.opt_kind == OPT_SGD) {  // sgd_optimizer.cc:36-44
        for (int i = 0; i < c.dim; ++i) num[i] -= lr * g[i];
      } else if (c.opt_kind == OPT_ADAGRAD) {  // avx_utils.h:29-38 (baseline form)
        float* norm = st;
        for (int i = 0; i < c.dim; ++i) {
          float gg = g[i] + c.weight_decay * num[i];
          norm[i] += gg * gg;
          float effective_lr = lr / std::sqrt(norm[i]);
          num[i] -= effective_lr * gg;
        }
      } else if (c.opt_kind == OPT_MOMENTUM) {  // momentum_optimizer.cc:49-72
        float* n = st;
        for (int i = 0; i < c.dim; ++i) {
          float dx = lr * (g[i] + c.weight_decay * num[i]);
          float new_n = n[i];
          float new_w = num[i];
          if (c.use_nesterov) {
            float prev_n = new_n;
            new_n = c.momentum * new_n - dx;
            new_w += -c.momentum * prev_n + (1 + c.momentum) * new_n;
          } else {
            new_n = c.momentum * new_n - dx;
            new_w += new_n;
          }
          n[i] = new_n;
          num[i] = new_w;
        }
      } else if (c.opt_kind == OPT_RMSPROP || c.opt_kind == OPT_RMSPROP_V2) {
        // rmsprop_optimizer.cc:46-63 (V1: config learning rate, (1 - momentum) dx^2) and
        // :119-137 (V2: the call's learning rate, plain dx^2); mixed float / double as written
        float* n = st;
        for (int i = 0; i < c.dim; ++i) {
          const float& cur_grad = g[i];
          float new_n = n[i];
          float new_w = num[i];
          double dx = cur_grad + static_cast<double>(c.weight_decay) * new_w;
          double eta;
          if (c.opt_kind == OPT_RMSPROP) {
            new_n = static_cast<double>(c.momentum) * new_n +
                    (1 - static_cast<double>(c.momentum)) * dx * dx;
            eta = static_cast<double>(c.learning_rate) / (std::sqrt(new_n) + 1);
          } else {
            new_n = static_cast<double>(c.momentum) * new_n + dx * dx;
            eta = static_cast<double>(lr) / (std::sqrt(new_n) + 1);
          }
          new_w -= eta * dx;
          n[i] = new_n;
          num[i] = new_w;
        }
      } else if (c.opt_kind == OPT_ADAM || c.opt_kind == OPT_AMSGRAD) {
        // adam_optimizer.cc:53-80, amsgrad_optimizer.cc:55-86. `sqrt` is written unqualified
        // there and resolves to ::sqrt(double) (checked with this toolchain: <cmath> only), so
        // the expressions around it are evaluated in double, as C++ promotes them.
        const bool ams = c.opt_kind == OPT_AMSGRAD;
        float* m = st;
        float* v = m + c.dim;
        float* vhat = v + c.dim;  // AMSGrad only
        float& beta1_power = (ams ? vhat : v)[c.dim];
        float& beta2_power = (ams ? vhat : v)[c.dim + 1];
        float lr_t = lr * sqrt(1 - beta2_power) / (1 - beta1_power);
        for (int i = 0; i < c.dim; ++i) {
          float cur_grad = g[i] + c.weight_decay * num[i];
          float new_m = m[i] + (cur_grad - m[i]) * (1 - c.beta1);
          float new_v = v[i] + (cur_grad * cur_grad - v[i]) * (1 - c.beta2);
          float den = new_v;
          if (ams) {
            float new_vhat = std::max(vhat[i], new_v);
            vhat[i] = new_vhat;
            den = new_vhat;
          }
          float new_w = num[i];
          if (c.use_nesterov) {
            new_w -= ((cur_grad * (1 - c.beta1) + c.beta1 * new_m) * lr_t) / (sqrt(den) + c.epsilon);
          } else {
            new_w -= (new_m * lr_t) / (sqrt(den) + c.epsilon);
          }
          num[i] = new_w;
          m[i] = new_m;
          v[i] = new_v;
        }
        beta1_power *= c.beta1;
        beta2_power *= c.beta2;
      } else if (c.opt_kind == OPT_BATCH_SOFTMAX) {  // batch_softmax_optimizer.cc:42-54
        float& B = num[0];
        int64_t A;
        std::memcpy(&A, st, sizeof(A));
        float alpha = lr;
        // (the rounding decorator calls Optimize without the step: its default 0 arrives here)
        const int64_t step = c.stochastic_rounding_float16 ? 0 : global_step;
        B = (1 - alpha) * B + alpha * static_cast<float>(step - A);
        A = step;
        std::memcpy(st, &A, sizeof(A));
      } else if (c.opt_kind == OPT_MOVING_AVERAGE) {  // moving_average_optimizer.cc:39-47
        for (int i = 0; i < c.dim; ++i) {
          float new_w = c.momentum * num[i] + (1 - c.momentum) * g[i];
          num[i] = new_w;
        }
      } else if (c.opt_kind == OPT_ADADELTA) {  // adadelta_optimizer.cc:50-66
        float* accum = st;
        float* accum_update = st + c.dim;
        for (int i = 0; i < c.dim; ++i) {
          float cur_grad = g[i] + c.weight_decay * num[i];
          float new_accum = accum[i] * c.averaging_ratio + cur_grad * cur_grad * (1 - c.averaging_ratio);
          float update = std::sqrt(accum_update[i] + c.epsilon) / std::sqrt(new_accum + c.epsilon) * cur_grad;
          float new_w = num[i] - update * lr;
          float new_accum_update =
              accum_update[i] * c.averaging_ratio + update * update * (1 - c.averaging_ratio);
          num[i] = new_w;
          accum[i] = new_accum;
          accum_update[i] = new_accum_update;
        }
      } else {  // ftrl_optimizer.cc:49-72
        float* norm = st;
        float* zero = st + c.dim;
        for (int i = 0; i < c.dim; ++i) {
          float norm_new = norm[i] + g[i] * g[i];
          float sigma = (std::sqrt(norm_new) - std::sqrt(norm[i])) / lr;
          zero[i] += (g[i] - sigma * num[i]);
          norm[i] = norm_new;
          // std::signbit(z) is a bool: 1 iff z < 0 (:63-67) -> for z > 0 the l1 term is 0
          num[i] = (std::abs(zero[i]) > c.l1)
                       ? lr * ((float)std::signbit(zero[i]) * c.l1 - zero[i]) /
                             (std::sqrt(norm[i]) + c.beta + c.l2 * lr)
                       : 0.0f;
        }
      }
      // StochasticRoundingFloat16OptimizerDecorator::Optimize (stochastic_rounding.h:48-56):
      // every element of the span the segment's optimizer was handed is rounded, and inside
      // CombinedOptimizer that span runs from the segment's first weight to the END of the
      // entry (optimizer_combination.cc:58-62: num.subspan(dim_size1_) keeps the tail)
      if (c.stochastic_rounding_float16)
        for (int i = seg_w_off[s]; i < dim; ++i)
          e.row[i] = stochastic_round(e.row[i], sr_uniform(seed, global_step, pos, i, (int)s));
    }
  }

  int expire_days(int64_t key) const {  // cuckoo_embedding_hash_table.cc:224-229
    auto it = slot_expire.find(slot_id_v2((uint64_t)key));
    return it == slot_expire.end() ? default_expire_days : it->second;
  }
  bool expired(int64_t key, uint32_t ts, int64_t max_update_time) const {  // :230
    return max_update_time - (int64_t)ts >= (int64_t)expire_days(key) * 86400;
  }
};

void touch_max_ts(Table* t, int64_t update_time) {  // bridge :161-162, :233-234
  t->max_update_ts = std::max(update_time, t->max_update_ts);
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------ filter
void* orc_filter_create(int kind, int64_t capacity, int split_num, int default_threshold,
                        int n_slot, const int32_t* slots, const int32_t* thresholds) {
  Filter* f = new Filter();
  f->kind = kind;
  f->slot_threshold.assign(1 << 15, default_threshold);
  for (int i = 0; i < n_slot; ++i) f->slot_threshold[slots[i]] = thresholds[i];
  if (kind == 1) {  // sliding_hash_filter.cc:28-40
    uint64_t cap = (uint64_t)capacity;
    if (cap < 300) cap = 300;
    if (split_num < 5) split_num = 5;
    f->max_backward = split_num - Filter::kMaxForward;
    uint64_t split_cap = cap / (uint64_t)(split_num - Filter::kMaxForward + 1);  // .h:33-35
    for (int i = 0; i < split_num; ++i)
      f->blocks.emplace_back(new HashFilter16(split_cap, 1.2));
  }
  return f;
}
void orc_filter_destroy(void* f) { delete (Filter*)f; }
void orc_filter_add(void* fp, const int64_t* ids, const int32_t* counts, int64_t n,
                    int32_t* old_counts) {
  Filter* f = (Filter*)fp;
  for (int64_t i = 0; i < n; ++i)
    old_counts[i] = f->kind == 0 ? 15 : (int32_t)f->add((uint64_t)ids[i], (uint32_t)counts[i]);
}
void orc_filter_get(void* fp, const int64_t* ids, int64_t n, int32_t* counts) {
  Filter* f = (Filter*)fp;
  for (int64_t i = 0; i < n; ++i) counts[i] = f->kind == 0 ? 15 : (int32_t)f->get((uint64_t)ids[i]);
}
// ops/hash_filter_intercept_gradient_op.cc:39-62
void orc_filter_intercept_gradient(void* fp, const int64_t* ids, const float* grad, int64_t n,
                                   int64_t width, float* out) {
  Filter* f = (Filter*)fp;
  for (int64_t i = 0; i < n; ++i) {
    bool filtered = f->should_be_filtered(ids[i], 1);
    for (int64_t j = 0; j < width; ++j) out[i * width + j] = filtered ? 0.f : grad[i * width + j];
  }
}
void orc_filter_state(void* fp, int64_t* out, int64_t out_len) {
  Filter* f = (Filter*)fp;
  int64_t nb = (int64_t)f->blocks.size();
  int64_t vals[5] = {(int64_t)f->head, f->head_increment, (int64_t)f->failure_count, nb,
                     nb ? (int64_t)f->blocks[0]->total_size : 0};
  for (int i = 0; i < 5 && i < out_len; ++i) out[i] = vals[i];
  for (int64_t b = 0; b < nb && 5 + b < out_len; ++b) out[5 + b] = (int64_t)f->blocks[b]->num_elements;
}

// ------------------------------------------------------------------ table
void* orc_table_create(const TableCfg* cfg) {
  Table* t = new Table();
  int w = 0;
  t->serving = cfg->entry_type == ENTRY_SERVING;
  for (int s = 0; s < cfg->n_segments; ++s) {
    t->segs.push_back(cfg->segments[s]);
    if (t->serving) {  // a serving entry is its compressor only (entry_accessor.cc:315-318)
      t->segs.back().opt_kind = OPT_SGD;
      t->segs.back().init_kind = INIT_ZEROS;
    }
    t->seg_w_off.push_back(w);
    w += cfg->segments[s].dim;
  }
  t->dim = w;
  int st = w;
  for (int s = 0; s < cfg->n_segments; ++s) {
    t->seg_st_off.push_back(st);
    const SegCfg& c = t->segs[s];
    st += state_floats(c.opt_kind, c.dim);
  }
  t->row_floats = st;
  t->default_expire_days = cfg->default_expire_days;
  for (int i = 0; i < cfg->n_slot_expire; ++i)
    t->slot_expire[cfg->slot_ids[i]] = cfg->slot_expire_days[i];
  t->seed = cfg->seed;
  return t;
}
void orc_table_destroy(void* t) { delete (Table*)t; }
// half neighbours / stochastic_round over arrays (tests compare them with the reference header)
void orc_half_up_down(const float* in, float* up, float* down, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    up[i] = half_neighbour(in[i], true);
    down[i] = half_neighbour(in[i], false);
  }
}
void orc_stochastic_round(const float* in, const float* p, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = stochastic_round(in[i], p[i]);
}
// Decode(Encode(v)) of one compressor over an array (tests compare it with the reference headers)
void orc_compress_roundtrip(int kind, float a, float b, const float* in, float* out, int64_t n) {
  SegCfg c{};
  c.comp_kind = kind;
  c.comp_a = a;
  c.comp_b = b;
  for (int64_t i = 0; i < n; ++i) out[i] = comp_q(c, in[i]);
}
int orc_table_dim(void* t) { return ((Table*)t)->dim; }
int orc_table_row_floats(void* t) { return ((Table*)t)->row_floats; }

// cuckoo_embedding_hash_table.cc:151-160
void orc_table_lookup(void* tp, const int64_t* ids, int64_t n, float* out) {
  Table* t = (Table*)tp;
  for (int64_t i = 0; i < n; ++i) {
    auto it = t->m.find(ids[i]);
    if (it == t->m.end())
      std::memset(out + i * t->dim, 0, sizeof(float) * t->dim);
    else
      std::memcpy(out + i * t->dim, it->second.row.data(), sizeof(float) * t->dim);
  }
}

// bridge :156-184: gate the whole batch against the pre-call table, then assign in order
void orc_table_assign(void* tp, void* fp, const int64_t* ids, const float* values, int64_t n,
                      int64_t req_time) {
  Table* t = (Table*)tp;
  Filter* f = (Filter*)fp;
  touch_max_ts(t, req_time);
  std::vector<int64_t> keep;
  for (int64_t i = 0; i < n; ++i) {
    if (f && !t->m.count(ids[i]) && f->should_be_filtered(ids[i], 1)) continue;
    keep.push_back(i);
  }
  for (int64_t i : keep) {  // cuckoo_embedding_hash_table.cc:174-186
    Entry& e = t->upsert(ids[i]);
    e.ts = (uint32_t)req_time;
    std::memcpy(e.row.data(), values + i * t->dim, sizeof(float) * t->dim);
    t->compress(e);
  }
}

// ops/hash_table_update_op.cc:67-71 + bridge :186-209: strictly per id, in order
void orc_table_assign_add(void* tp, void* fp, const int64_t* ids, const float* values, int64_t n,
                          int64_t req_time) {
  Table* t = (Table*)tp;
  Filter* f = (Filter*)fp;
  for (int64_t i = 0; i < n; ++i) {
    touch_max_ts(t, req_time);
    if (f && !t->m.count(ids[i]) && f->should_be_filtered(ids[i], 1)) continue;
    Entry& e = t->upsert(ids[i]);  // cuckoo_embedding_hash_table.cc:189-195
    e.ts = (uint32_t)req_time;
    for (int j = 0; j < t->dim; ++j) e.row[j] += values[i * t->dim + j];
    t->compress(e);
  }
}

// bridge :229-313
void orc_table_apply_gradients(void* tp, void* fp, const int64_t* ids, const float* grads,
                               int64_t n, const float* lrs, int64_t req_time, int64_t global_step,
                               int enable_dedup) {
  Table* t = (Table*)tp;
  Filter* f = (Filter*)fp;
  touch_max_ts(t, req_time);
  const int D = t->dim;
  std::vector<int64_t> ids_after;
  std::vector<const float*> grads_after;
  std::vector<uint32_t> pos_after;  // batch position of the (first) occurrence: rounding stream only
  std::vector<float> cache;
  if (enable_dedup) {
    std::vector<uint32_t> first_pos;
    cache.resize((size_t)n * D);
    std::unordered_map<int64_t, int64_t> pos;  // id -> unique index (first-occurrence order)
    std::vector<int64_t> uniq;
    std::vector<uint32_t> counts;
    for (int64_t i = 0; i < n; ++i) {
      auto it = pos.find(ids[i]);
      if (it == pos.end()) {
        pos.emplace(ids[i], (int64_t)uniq.size());
        std::memcpy(cache.data() + uniq.size() * D, grads + i * D, sizeof(float) * D);
        uniq.push_back(ids[i]);
        first_pos.push_back((uint32_t)i);
        counts.push_back(1);
      } else {  // ReduceSum(dest, src, dest): dest[j] = dest[j] + src[j], in input order
        float* dest = cache.data() + it->second * D;
        for (int j = 0; j < D; ++j) dest[j] = dest[j] + grads[i * D + j];
        counts[it->second]++;
      }
    }
    // The reference iterates a flat_hash_map here (unspecified order); ids are distinct, rows
    // independent, and filter cells of distinct ids only interact through aliasing, so
    // first-occurrence order is used as the canonical order.
    for (size_t u = 0; u < uniq.size(); ++u) {
      if (f && !t->m.count(uniq[u]) && f->should_be_filtered(uniq[u], counts[u])) continue;
      ids_after.push_back(uniq[u]);
      grads_after.push_back(cache.data() + u * D);
      pos_after.push_back(first_pos[u]);
    }
  } else {
    for (int64_t i = 0; i < n; ++i) {
      if (f && !t->m.count(ids[i]) && f->should_be_filtered(ids[i], 1)) continue;
      ids_after.push_back(ids[i]);
      grads_after.push_back(grads + i * D);
      pos_after.push_back((uint32_t)i);
    }
  }
  for (size_t k = 0; k < ids_after.size(); ++k) {  // cuckoo_embedding_hash_table.cc:198-216
    Entry& e = t->upsert(ids_after[k]);
    e.ts = (uint32_t)req_time;
    t->optimize(e, grads_after[k], lrs, global_step, pos_after[k]);
  }
}

int64_t orc_table_size(void* t) { return (int64_t)((Table*)t)->m.size(); }
void orc_table_contains(void* tp, const int64_t* ids, int64_t n, uint8_t* out) {
  Table* t = (Table*)tp;
  for (int64_t i = 0; i < n; ++i) out[i] = t->m.count(ids[i]) ? 1 : 0;
}
void orc_table_clear(void* t) { ((Table*)t)->m.clear(); }
int64_t orc_table_max_update_ts(void* t) { return ((Table*)t)->max_update_ts; }

// cuckoo_embedding_hash_table.cc:220-233
void orc_table_evict(void* tp, int64_t max_update_time) {
  Table* t = (Table*)tp;
  for (auto it = t->m.begin(); it != t->m.end();) {
    if (t->expired(it->first, it->second.ts, max_update_time))
      it = t->m.erase(it);
    else
      ++it;
  }
}

// Dump sorted by id (the reference's dump order is unspecified). apply_expiry restates
// ops/hash_table_save_op.cc:142-150 with max_update_ts as "now".
int64_t orc_table_export(void* tp, int64_t max_rows, int apply_expiry, int64_t* ids, float* rows,
                         uint32_t* ts) {
  Table* t = (Table*)tp;
  std::vector<int64_t> keys;
  for (auto& kv : t->m)
    if (!apply_expiry || !t->expired(kv.first, kv.second.ts, t->max_update_ts))
      keys.push_back(kv.first);
  std::sort(keys.begin(), keys.end());
  int64_t k = 0;
  for (; k < (int64_t)keys.size() && k < max_rows; ++k) {
    const Entry& e = t->m[keys[k]];
    ids[k] = keys[k];
    std::memcpy(rows + k * t->row_floats, e.row.data(), sizeof(float) * t->row_floats);
    ts[k] = e.ts;
  }
  return k;
}

// ops/hash_table_restore_op.cc:82-131 (upsert of full entries)
void orc_table_import(void* tp, const int64_t* ids, const float* rows, const uint32_t* ts,
                      int64_t n) {
  Table* t = (Table*)tp;
  for (int64_t i = 0; i < n; ++i) {
    Entry& e = t->upsert(ids[i]);
    std::memcpy(e.row.data(), rows + i * t->row_floats, sizeof(float) * t->row_floats);
    t->compress(e);
    e.ts = ts[i];
    t->max_update_ts = std::max<int64_t>(t->max_update_ts, ts[i]);
  }
}

// ops/hash_table_lookup_op.cc:154-248
void orc_tables_fused_lookup(void* const* tables, int K, const int64_t* ids,
                             const int32_t* fused_slot_size, int N, float* out_emb,
                             int32_t* emb_splits, int32_t* id_offsets, int32_t* emb_offsets,
                             int32_t* emb_sizes) {
  int output_dim = 0, prev_output_dim = 0;
  const int cnt = K * N;
  for (int i = 0; i < cnt; ++i) id_offsets[i] = emb_offsets[i] = emb_sizes[i] = 0;
  for (int shard = 0; shard < N; ++shard) {
    for (int k = 0; k < K; ++k) {
      int idx = K * shard + k;
      int seg_dim = ((Table*)tables[k])->dim * fused_slot_size[idx];
      output_dim += seg_dim;
      emb_sizes[idx] = seg_dim;
      if (idx > 0) {
        emb_offsets[idx] = emb_offsets[idx - 1] + emb_sizes[idx - 1];
        id_offsets[idx] = id_offsets[idx - 1] + fused_slot_size[idx - 1];
      }
    }
    emb_splits[shard] = output_dim - prev_output_dim;
    prev_output_dim = output_dim;
  }
  if (!out_emb) return;
  for (int shard = 0; shard < N; ++shard)
    for (int k = 0; k < K; ++k) {
      int idx = shard * K + k;
      orc_table_lookup(tables[k], ids + id_offsets[idx], fused_slot_size[idx],
                       out_emb + emb_offsets[idx]);
    }
}

// ops/hash_table_update_op.cc:183-256; canonical order = shard-major
void orc_tables_fused_apply_gradient(void* const* tables, void* const* filters, int K,
                                     const int64_t* ids, const int32_t* fused_slot_size,
                                     const float* grads, const int32_t* id_offsets,
                                     const int32_t* grad_offsets, const float* lrs,
                                     const int32_t* lr_lengths, int64_t req_time,
                                     int64_t global_step, int N, int enable_grad_accumulation) {
  std::vector<int> lr_off(K, 0);
  for (int k = 1; k < K; ++k) lr_off[k] = lr_off[k - 1] + lr_lengths[k - 1];
  for (int shard = 0; shard < N; ++shard)
    for (int k = 0; k < K; ++k) {
      int idx = shard * K + k;
      orc_table_apply_gradients(tables[k], filters ? filters[k] : nullptr, ids + id_offsets[idx],
                                grads + grad_offsets[idx], fused_slot_size[idx], lrs + lr_off[k],
                                req_time, global_step, enable_grad_accumulation);
    }
}

// ------------------------------------------------------------------ routing
// ops/fused_reorder_by_indices.cc:36-112 + distribution_ops.py:255-272
int64_t orc_fused_reorder_by_indices(const int64_t* const* inputs, const int64_t* lens, int M,
                                     int N, const int32_t* dims, int shard_rule, int64_t* out_ids,
                                     int32_t* shard_sizes, int32_t* sharded_slot_sizes,
                                     int32_t* const* out_offsets) {
  auto floormod = [](int64_t a, int64_t b) { int64_t r = a % b; return r < 0 ? r + b : r; };
  for (int i = 0; i < N; ++i) shard_sizes[i] = 0;
  for (int i = 0; i < N * M; ++i) sharded_slot_sizes[i] = 0;
  std::vector<std::unordered_map<int64_t, int>> sets(M);
  std::vector<std::vector<std::vector<int64_t>>> splits(N, std::vector<std::vector<int64_t>>(M));
  for (int m = 0; m < M; ++m)
    for (int64_t i = 0; i < lens[m]; ++i) {
      int64_t v = inputs[m][i];
      if (sets[m].emplace(v, 0).second) {
        int n = shard_rule == 1 ? (int)(1 + floormod(v, N - 1)) : (int)floormod(v, N);
        splits[n][m].push_back(v);
        sharded_slot_sizes[n * M + m]++;
      }
    }
  int64_t k = 0;
  for (int n = 0; n < N; ++n)
    for (int m = 0; m < M; ++m) {
      for (int64_t v : splits[n][m]) out_ids[k++] = v;
      shard_sizes[n] += (int32_t)splits[n][m].size();
    }
  int off = 0;
  for (int n = 0; n < N; ++n)
    for (int m = 0; m < M; ++m)
      for (int64_t v : splits[n][m]) {
        sets[m][v] = off;
        off += dims[m];
      }
  for (int m = 0; m < M; ++m)
    for (int64_t i = 0; i < lens[m]; ++i) out_offsets[m][i] = sets[m][inputs[m][i]];
  return k;
}

// ops/map_id_to_embedding_op.cc:247-287
void orc_fused_gather_embeddings_by_input(const float* fused, const int32_t* const* offsets,
                                          const int64_t* lens, const int32_t* dims, int M,
                                          float* const* outs) {
  for (int m = 0; m < M; ++m)
    for (int64_t j = 0; j < lens[m]; ++j)
      std::memcpy(outs[m] + j * dims[m], fused + offsets[m][j], sizeof(float) * dims[m]);
}
// ops/map_id_to_embedding_op.cc:302-338
void orc_fused_gather_embeddings_by_input_gradient(int64_t fused_len, const float* const* grads,
                                                   const int32_t* const* offsets,
                                                   const int64_t* lens, const int32_t* dims, int M,
                                                   float* out) {
  std::memset(out, 0, sizeof(float) * fused_len);
  for (int m = 0; m < M; ++m)
    for (int64_t j = 0; j < lens[m]; ++j) {
      float* b = out + offsets[m][j];
      const float* a = grads[m] + j * dims[m];
      for (int k = 0; k < dims[m]; ++k) b[k] = a[k] + b[k];
    }
}

// ------------------------------------------------------------------ pooling
// ops/reduce_op.cc:33-50
void orc_reduce_sum(const int64_t* row_ids, const float* v, int64_t n, int64_t dim, int64_t batch,
                    float* out) {
  std::memset(out, 0, sizeof(float) * batch * dim);
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < dim; ++j) out[row_ids[i] * dim + j] += v[i * dim + j];
}
// ops/reduce_op.cc:59-86
void orc_reduce_mean(const int64_t* row_ids, const float* v, int64_t n, int64_t dim, int64_t batch,
                     float* out) {
  std::memset(out, 0, sizeof(float) * batch * dim);
  std::vector<size_t> counter(batch, 0);
  for (int64_t i = 0; i < n; ++i) {
    for (int64_t j = 0; j < dim; ++j) out[row_ids[i] * dim + j] += v[i * dim + j];
    counter[row_ids[i]] += 1;
  }
  for (int64_t b = 0; b < batch; ++b) {
    float multiply = 1.0 / static_cast<float>(counter[b]);
    for (int64_t j = 0; j < dim; ++j) out[b * dim + j] *= multiply;
  }
}
// ops/reduce_op.cc:128-147
void orc_reduce_sum_gradient(const int64_t* row_ids, const float* grads, int64_t n, int64_t dim,
                             int64_t batch, float* out) {
  (void)batch;
  for (int64_t i = 0; i < n; ++i)
    std::memcpy(out + i * dim, grads + row_ids[i] * dim, sizeof(float) * dim);
}
// ops/reduce_op.cc:154-179 (counter sized by batch here; the reference sizes it by grad_size)
void orc_reduce_mean_gradient(const int64_t* row_ids, const float* grads, int64_t n, int64_t dim,
                              int64_t batch, float* out) {
  std::vector<size_t> counter(batch, 0);
  for (int64_t i = 0; i < n; ++i) counter[row_ids[i]] += 1;
  for (int64_t i = 0; i < n; ++i) {
    float multiply = 1.0 / static_cast<float>(counter[row_ids[i]]);
    for (int64_t j = 0; j < dim; ++j) out[i * dim + j] = grads[row_ids[i] * dim + j] * multiply;
  }
}
// ops/reduce_op.cc:239-272
void orc_fused_reduce_sum_and_split(const int64_t* row_ids, const float* v, int64_t n, int64_t dim,
                                    int64_t batch, const int32_t* split_dims, int M,
                                    float* const* outs) {
  for (int m = 0; m < M; ++m) std::memset(outs[m], 0, sizeof(float) * batch * split_dims[m]);
  for (int64_t i = 0; i < n; ++i) {
    int off = 0;
    for (int m = 0; m < M; ++m) {
      const float* a = v + i * dim + off;
      float* b = outs[m] + row_ids[i] * split_dims[m];
      for (int k = 0; k < split_dims[m]; ++k) b[k] = a[k] + b[k];
      off += split_dims[m];
    }
  }
}
// ops/reduce_op.cc:291-315
void orc_fused_reduce_sum_and_split_gradient(const int64_t* row_ids, const float* const* grads,
                                             int64_t n, int64_t batch, const int32_t* split_dims,
                                             int M, float* out) {
  (void)batch;
  int grad_dim = 0;
  for (int m = 0; m < M; ++m) grad_dim += split_dims[m];
  int off = 0;
  for (int m = 0; m < M; ++m) {
    for (int64_t j = 0; j < n; ++j)
      std::memcpy(out + j * grad_dim + off, grads[m] + row_ids[j] * split_dims[m],
                  sizeof(float) * split_dims[m]);
    off += split_dims[m];
  }
}
// ops/reduce_op.cu.cc:53-161: plain sorted segment sum (summed left to right here)
void orc_fused_sorted_segment_sum(const int32_t* const* indices, const float* const* values,
                                  const int64_t* lens, const int64_t* dims,
                                  const int64_t* out_rows, int N, float* const* outs) {
  for (int t = 0; t < N; ++t) {
    std::memset(outs[t], 0, sizeof(float) * out_rows[t] * dims[t]);
    for (int64_t i = 0; i < lens[t]; ++i)
      for (int64_t j = 0; j < dims[t]; ++j)
        outs[t][indices[t][i] * dims[t] + j] += values[t][i * dims[t] + j];
  }
}
// embedding_combiners.py:71-100 (FirstN via ragged to_tensor + pad to max_seq_length)
void orc_first_n(const int64_t* row_ids, const float* v, int64_t n, int64_t dim, int64_t batch,
                 int64_t max_seq, float* out) {
  std::memset(out, 0, sizeof(float) * batch * max_seq * dim);
  std::vector<int64_t> fill(batch, 0);
  for (int64_t i = 0; i < n; ++i) {
    int64_t b = row_ids[i];
    if (fill[b] < max_seq) {
      std::memcpy(out + (b * max_seq + fill[b]) * dim, v + i * dim, sizeof(float) * dim);
      fill[b]++;
    }
  }
}

// Standalone optimizer steps on caller-owned arrays, used to pin the math against the
// reference's optimizer unit tests and avx_utils.h-generated vectors.
void orc_adagrad(float* num, float* norm, const float* grad, int64_t len, float lr, float wd) {
  for (int64_t i = 0; i < len; ++i) {
    float g = grad[i] + wd * num[i];
    norm[i] += g * g;
    float effective_lr = lr / std::sqrt(norm[i]);
    num[i] -= effective_lr * g;
  }
}


// CPU baseline for bench.py (measurement aid, not a parity function): T threads, each owning the
// table shard of the ids with mix(id) % T == t (no locks, i.e. an upper bound for a striped-lock
// CPU table such as the reference's). Every thread scans each batch, gathers its own occurrences,
// looks them up and applies the gradients with dedup. Tables are pre-populated with exactly the
// ids of the sample (untimed). Returns the wall time of the timed part in seconds.
double orc_bench_lookup_apply_mt(const TableCfg* cfg, const int64_t* ids, const float* grads,
                                 int64_t n, int n_batches, float lr, int T, int64_t* rows_out) {
  std::vector<void*> tabs(T);
  for (int t = 0; t < T; ++t) tabs[t] = orc_table_create(cfg);
  const int D = orc_table_dim(tabs[0]);
  auto shard_of = [T](int64_t id) {
    uint64_t x = (uint64_t)id;
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 27; x *= 0x94d049bb133111ebULL; x ^= x >> 31;
    return (int)(x % (uint64_t)T);
  };
  auto run = [&](bool populate) {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        std::vector<int64_t> my_ids;
        std::vector<float> my_g, out;
        for (int b = 0; b < n_batches; ++b) {
          const int64_t* bid = ids + (int64_t)b * n;
          my_ids.clear();
          my_g.clear();
          for (int64_t i = 0; i < n; ++i)
            if (shard_of(bid[i]) == t) {
              my_ids.push_back(bid[i]);
              if (!populate) my_g.insert(my_g.end(), grads + i * D, grads + (i + 1) * D);
            }
          const int64_t m = (int64_t)my_ids.size();
          if (populate) {
            my_g.assign((size_t)m * D, 0.f);
            orc_table_assign_add(tabs[t], nullptr, my_ids.data(), my_g.data(), m, 1);
          } else {
            out.resize((size_t)m * D);
            orc_table_lookup(tabs[t], my_ids.data(), m, out.data());
            orc_table_apply_gradients(tabs[t], nullptr, my_ids.data(), my_g.data(), m, &lr, 2, b, 1);
          }
        }
      });
    for (auto& x : th) x.join();
  };
  run(true);
  auto t0 = std::chrono::steady_clock::now();
  run(false);
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  int64_t rows = 0;
  for (int t = 0; t < T; ++t) {
    rows += orc_table_size(tabs[t]);
    orc_table_destroy(tabs[t]);
  }
  if (rows_out) *rows_out = rows;
  return dt;
}

}  // extern "C"
