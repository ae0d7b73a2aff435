/*
 * monolith_hip.h — C ABI of libmonolith_hip.so, the MI355X (gfx950) replacement for the
 * sparse-embedding hot path of Monolith's TensorFlow custom ops.
 *
 * Conventions
 *   - Every data pointer is a DEVICE pointer (hipMalloc / torch.Tensor.data_ptr()) unless its
 *     name starts with `h_` (host pointer, read before the call returns).
 *   - Every call that launches work takes a `void* stream` (a hipStream_t; NULL = default
 *     stream) and is stream-ordered; nothing synchronises the host unless stated.
 *   - Return value: 0 = OK, non-zero = error; mono_last_error() gives the message of the
 *     calling thread's last failure. No C++ exception crosses this boundary.
 *   - Handles are opaque, owned by the caller, freed with the matching *_destroy.
 *   - One host thread drives one GPU; calls on one handle must not overlap.
 *
 * Each entry point cites the reference interface it replaces (paths relative to
 * monolith/native_training/ in 0x1997/monolith).
 */
#ifndef MONOLITH_HIP_H_
#define MONOLITH_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- configuration PODs (mirror runtime/hash_table/embedding_hash_table.proto:19-87,
 *      optimizer/optimizer.proto:19-72, initializer/initializer_config.proto:19-46) ---- */

enum {
  MONO_OPT_SGD = 0, MONO_OPT_ADAGRAD = 1, MONO_OPT_FTRL = 2,
  /* optimizer/{momentum,rmsprop,adadelta}_optimizer.cc */
  MONO_OPT_MOMENTUM = 3, MONO_OPT_RMSPROP = 4, MONO_OPT_RMSPROP_V2 = 5, MONO_OPT_ADADELTA = 6,
  /* optimizer/{adam,amsgrad}_optimizer.cc: m | v (| vhat) per weight + beta1^t, beta2^t per row */
  MONO_OPT_ADAM = 7, MONO_OPT_AMSGRAD = 8,
  /* optimizer/moving_average_optimizer.cc:39-47: w = momentum w + (1 - momentum) g, no state */
  MONO_OPT_MOVING_AVERAGE = 9,
  /* optimizer/batch_softmax_optimizer.cc:42-54: dim 1; w = (1 - lr) w + lr (global_step - A),
   * A = global_step; A is an int64 kept in the segment's two state floats (low | high word) */
  MONO_OPT_BATCH_SOFTMAX = 10
};
/* compressor/float_compressor.cc:42-137 (FloatCompressorConfig, float_compressor.proto:19-52) */
enum { MONO_COMP_FP32 = 0, MONO_COMP_FP16 = 1, MONO_COMP_FIXED_R8 = 2, MONO_COMP_ONE_BIT = 3 };
/* EntryConfig.EntryType (embedding_hash_table.proto:35-42) */
enum { MONO_ENTRY_TRAINING = 1, MONO_ENTRY_SERVING = 2 };
enum { MONO_INIT_ZEROS = 0, MONO_INIT_ONES = 1, MONO_INIT_CONSTANT = 2, MONO_INIT_UNIFORM = 3 };

typedef struct {
  int32_t dim;          /* EntryConfig.Segment.dim_size */
  int32_t opt_kind;     /* MONO_OPT_* */
  float init_acc;       /* initial_accumulator_value (Adagrad / FTRL), default 0.1 */
  float weight_decay;   /* Adagrad weight_decay_factor, default 0 */
  float l1;             /* FTRL l1_regularization_strength */
  float l2;             /* FTRL l2_regularization_strength */
  float beta;           /* FTRL beta */
  int32_t init_kind;    /* MONO_INIT_* */
  float init_a;         /* constant value, or uniform minval */
  float init_b;         /* uniform maxval */
  /* ABI version 2: */
  float momentum;        /* Momentum / RMSProp(V2) momentum (0.9) */
  float epsilon;         /* Adadelta epsilon (0.01) */
  float averaging_ratio; /* Adadelta averaging_ratio (0.9) */
  float learning_rate;   /* RMSProp reads the CONFIG learning rate (rmsprop_optimizer.cc:59), not
                            the per-call tensor; default 0.01 */
  int32_t use_nesterov;  /* Momentum, Adam, AMSGrad */
  /* ABI version 3: */
  float beta1;           /* Adam / AMSGrad (0.9) */
  float beta2;           /* Adam / AMSGrad (0.99) */
  /* ABI version 4 (serving entries only, entry_type MONO_ENTRY_SERVING): */
  int32_t comp_kind;     /* MONO_COMP_* */
  float comp_a;          /* FixedR8: r (1.0); OneBit: amplitude (0.1) */
  float comp_b;          /* OneBit: init_scale (1.0) */
  int32_t stochastic_rounding_float16; /* OptimizerConfig.stochastic_rounding_float16
                            (optimizer.proto:213, stochastic_rounding.h:27-56): after the segment's
                            step every weight from the segment's first to the END of the entry is
                            rounded to one of its two binary16 neighbours */
} mono_segment_cfg;

typedef struct {
  int32_t n_segments;
  int32_t default_expire_days;       /* SlotExpireTimeConfig.default_expire_time (36500) */
  const mono_segment_cfg* segments;  /* host pointer, n_segments entries */
  int64_t initial_capacity;          /* rows the table must hold before its first rehash */
  int32_t n_slot_expire;             /* SlotExpireTimeConfig.slot_expire_times */
  int32_t entry_type;                /* 0 / MONO_ENTRY_TRAINING, or MONO_ENTRY_SERVING: values are
                                        stored as the segment's compressor would return them
                                        (entry_accessor.cc:43-98); no optimizer, no save */
  const int32_t* slot_ids;           /* host pointer */
  const int32_t* slot_expire_days;   /* host pointer */
  uint64_t seed;                     /* seed of the random_uniform initializer */
  /* ABI version 5. embedding_hash_table.proto:84-86 / embedding_hash_table_tf_bridge.cc:69-100:
   * every feature_evict_every_n_hours of WALL CLOCK the table evicts with the largest req_time it
   * has seen (mono_table_evict(table, max_update_ts)). The reference does it from a thread of the
   * bridge that polls every 10 s; here the check runs at the entry of every write call (assign /
   * assign_add / apply_gradients), on that call's stream, i.e. between two calls like the thread. */
  int32_t enable_feature_eviction;
  int32_t feature_evict_every_n_hours;
} mono_table_cfg;

const char* mono_last_error(void);
/* Library/ABI version, bumped when a signature changes. */
int mono_abi_version(void);

/* ---- hash filter: runtime/neo_filter/{hash_filter.h,sliding_hash_filter.cc},
 *      ops/neo_filter_tf_bridge.cc:25-40, ops/hash_filter_op.cc:109-141 ---- */
enum { MONO_FILTER_DUMMY = 0, MONO_FILTER_SLIDING = 1 };
int mono_filter_create(int kind, int64_t capacity, int split_num, int default_threshold,
                       int n_slot_thresholds, const int32_t* h_slots,
                       const int32_t* h_thresholds, void** out_filter);
int mono_filter_destroy(void* filter);
/* Batch form of Filter::add (sliding_hash_filter.cc:57-89): old_counts[i] = count before the add. */
int mono_filter_add(void* filter, const int64_t* ids, const int32_t* counts, int64_t n,
                    int32_t* old_counts, void* stream);
/* Filter::get (sliding_hash_filter.cc:91-113). */
int mono_filter_get(void* filter, const int64_t* ids, int64_t n, int32_t* counts, void* stream);
/* ops/hash_filter_intercept_gradient_op.cc:39-62 — zero the grad rows of filtered ids
 * (every id is counted once). */
int mono_filter_intercept_gradient(void* filter, const int64_t* ids, const float* grad, int64_t n,
                                   int64_t width, float* out, void* stream);
/* Host-visible filter state: h_out = {head, head_increment, failure_count, n_blocks,
 * block_total_size, num_elements[0..n_blocks)}; synchronises the stream. */
int mono_filter_state(void* filter, int64_t* h_out, int64_t h_out_len, void* stream);

/* Filter checkpoints: ops/hash_filter_save_op.cc:73-105, ops/hash_filter_restore_op.cc:75-110,
 * runtime/neo_filter/hash_filter.cc:27-78, sliding_hash_filter.cc:157-213. One call per block
 * ("split"); h_cells is HOST memory holding the block's whole cell vector (block_total_size + 64
 * cells); num_elements as the reference's HashFilterSplitMetaDump carries it. */
int mono_filter_export_split(void* filter, int split, uint16_t* h_cells, int64_t n_cells,
                             int64_t* h_num_elements, void* stream);
int mono_filter_import_split(void* filter, int split, const uint16_t* h_cells, int64_t n_cells,
                             int64_t num_elements, void* stream);
int mono_filter_set_state(void* filter, int64_t head, int64_t head_increment, int64_t failure_count,
                          void* stream);

/* Debug aid without a counterpart in the reference: 1 if a workgroup of any fused radix-sort
 * pass of this process (csrc/sort.hip) ever gave up waiting for its predecessors, 0 otherwise, -1 on a HIP
 * error. Synchronises the device. */
int mono_debug_sort_error(void);
/* Test / measurement hook, no counterpart in the reference: stable sort of n (key, value) pairs on
 * the low end_bit key bits (vin NULL = identity values), reps times; average ms per sort in
 * *avg_ms (HIP events on the stream). Allocates and frees its own scratch. */
int mono_debug_sort_pairs(const uint32_t* kin, const uint32_t* vin, int64_t n, int end_bit,
                          uint32_t* kout, uint32_t* vout, int reps, float* avg_ms, void* stream);

/* ---- touched-key set: runtime/hopscotch/hopscotch_hash_set.{h,cc},
 *      ops/touched_key_set_{op,insert_op,steal_op}.cc, touched_key_set_ops.py ----
 * insert: the keys go in one by one, in input order; whenever an insert finds MORE than
 * `capacity` keys in the set, the whole set is dropped first (hopscotch_hash_set.cc:113-122);
 * *h_dropped = keys dropped during the call (MonolithTouchedKeySetInsert's total_dropped_num).
 * steal: GetAndClear (:168-193): *h_count keys into out_ids (device, room for out_cap), order
 * unspecified. Both synchronise the stream. */
int mono_keyset_create(int64_t capacity, void** out_set);
int mono_keyset_destroy(void* set);
int mono_keyset_size(void* set, int64_t* out_size);
int mono_keyset_insert(void* set, const int64_t* ids, int64_t n, int64_t* h_dropped, void* stream);
int mono_keyset_steal(void* set, int64_t* out_ids, int64_t out_cap, int64_t* h_count, void* stream);

/* ---- table lifetime: ops/hash_table_op.cc:38-131, ops/embedding_hash_table_tf_bridge.cc:48-104 ---- */
int mono_table_create(const mono_table_cfg* cfg, void** out_table);
int mono_table_destroy(void* table);
int mono_table_dim(void* table);         /* EmbeddingHashTableTfBridge::dim_size */
int mono_table_slice_size(void* table);  /* ::slice_size = number of segments */

/* ops/hash_table_lookup_op.cc:36-73 (MonolithHashTableLookup): out[n, dim]; miss => zeros. */
int mono_table_lookup(void* table, const int64_t* ids, int64_t n, float* out, void* stream);
/* No counterpart in the reference's single-table op set: it is what the reference's input pipeline
 * does one level up, where the next batch's ids are known while the current step is still running
 * (embedding_prefetch_capacity / enable_async_optimize in distributed_ps_sync.py, the data
 * pipeline's prefetch in native_task.py) — here without the staleness those options accept.
 * `ids` (device, ready on `stream`) are the ids of the mono_table_lookup `ahead` calls from now on
 * this table: 1 = the next lookup, 2 = the one after it, at most 3. They are read on another
 * stream some time after the call and must stay valid until the lookup they announce has been
 * issued (unlike every other argument of this library, which only has to be valid in stream order). In a training loop (lookup, update, lookup, ...) the table
 * resolves their slots and prepares the update order on streams of its own, beside the updates
 * of the batches before; the lookup they were announced for checks on the device that it got the
 * hinted ids and only gathers the rows. Purely an optimisation: results are the same with or
 * without the call, whatever ids follow (a hint that does not match, or that an eviction, a
 * clear, a rehash or an ungated key insertion has overtaken, is dropped; insertions by an update
 * gated by a hash filter keep it: the ids it recorded as absent are probed again). The table keeps a queue, oldest first, and takes a
 * hint only if it extends that queue by one (`ahead` == hints queued + 1): a caller may announce
 * ahead = 1 and ahead = 2 at every step, between the lookup and its update, and the table picks
 * what it does not have yet. Two ahead is what pays: beside an update the sort runs
 * several times slower than alone and needs the whole step. A no-op outside the training pattern. */
int mono_table_lookup_hint(void* table, const int64_t* ids, int64_t n, int ahead, void* stream);

/* ops/hash_table_update_op.cc:28-50 (MonolithHashTableAssign), bridge :156-184. filter may be NULL. */
int mono_table_assign(void* table, void* filter, const int64_t* ids, const float* values,
                      int64_t n, int64_t req_time, void* stream);
/* ops/hash_table_update_op.cc:52-75 (MonolithHashTableAssignAdd), bridge :186-209. */
int mono_table_assign_add(void* table, void* filter, const int64_t* ids, const float* values,
                          int64_t n, int64_t req_time, void* stream);
/* ops/hash_table_update_op.cc:95-164 (MonolithHashTableOptimize), bridge :229-313.
 * h_learning_rates: host float[slice_size]. */
int mono_table_apply_gradients(void* table, void* filter, const int64_t* ids, const float* grads,
                               int64_t n, const float* h_learning_rates, int n_learning_rates,
                               int64_t req_time, int64_t global_step, int enable_dedup,
                               void* stream);

/* hash_table/misc_ops.cc:38-60 (MonolithHashTableSize); synchronises the stream. */
int mono_table_size(void* table, int64_t* h_size, void* stream);
/* EmbeddingHashTableInterface::Contains (embedding_hash_table_interface.h:107): out[i] in {0,1}. */
int mono_table_contains(void* table, const int64_t* ids, int64_t n, uint8_t* out, void* stream);
/* HashTableLookupEntryOp (ops/hash_table_lookup_op.cc:79-107) ->
 * CuckooEmbeddingHashTable::LookupEntry (cuckoohash/cuckoo_embedding_hash_table.cc:162-171): the
 * whole entry of every id: rows[n, row_floats] (weights | optimizer context), ts[n], found[n]
 * (0 = not in the table: the op returns an empty string for it). Serialization of the EntryDump
 * is host work (monolith_amd/checkpoint.py). */
int mono_table_lookup_entry(void* table, const int64_t* ids, int64_t n, float* rows, uint32_t* ts,
                            uint8_t* found, void* stream);
/* EmbeddingHashTableInterface::Clear (cuckoo_embedding_hash_table.cc:270-273). */
int mono_table_clear(void* table, void* stream);
/* CuckooEmbeddingHashTable::Evict (cuckoo_embedding_hash_table.cc:220-233). */
int mono_table_evict(void* table, int64_t max_update_time, void* stream);
/* EmbeddingHashTableTfBridge::max_update_ts_sec (bridge :377-379). */
int mono_table_max_update_ts(void* table, int64_t* h_out);
/* The clock of the periodic eviction (enable_feature_eviction): now_sec >= 0 replaces the wall
 * clock (std::chrono::system_clock in the bridge, :74-78) until it is set again, -1 gives it
 * back. For tests and for replaying a log at its own pace; h_evictions (may be NULL) receives
 * the number of periodic evictions the table has run so far. */
int mono_table_set_clock(void* table, int64_t now_sec, int64_t* h_evictions);
/* Make room for `rows` live rows without a rehash inside a later call (the cuckoo map's
 * reserve(): cuckoohash_map.hpp:131); may reallocate + rehash; synchronises. */
int mono_table_reserve(void* table, int64_t rows, void* stream);
/* Dump for save / tests (hash_table/misc_ops.cc SaveAsTensor; save-time expiry
 * ops/hash_table_save_op.cc:142-150 when apply_expiry != 0): writes up to max_rows live rows
 * (ids[], rows[max_rows, dim + state], ts[]) in slot order; *h_count = rows written; synchronises. */
int mono_table_export(void* table, int64_t max_rows, int apply_expiry, int64_t* ids, float* rows,
                      uint32_t* ts, int64_t* h_count, void* stream);
/* Restore path (ops/hash_table_restore_op.cc:82-131): upsert full rows (value + optimizer state + ts). */
int mono_table_import(void* table, const int64_t* ids, const float* rows, const uint32_t* ts,
                      int64_t n, void* stream);
/* Measurement hooks (no reference counterpart; bench.py's roofline leg): when enabled, every
 * write-path call records HIP events on its own stream around its three phases;
 * mono_table_last_phase_ms waits for the last call and returns
 * h_ms[3] = {resolve slots (find/claim), group (radix sort), apply rows}. */
int mono_table_profile(void* table, int enable);
int mono_table_last_phase_ms(void* table, float* h_ms);
/* Duration of the sub-group apply kernel (k_apply) of the last write call ALONE, from HIP events
 * on the auxiliary stream it is launched on (it runs beside k_apply_long): the figure that is set
 * against rocprofv3's per-kernel average. Profiling must be on. */
int mono_table_last_apply_kernel_ms(void* table, float* h_ms);
/* Which path the calls on this table took so far (host counters, no device work; no reference
 * counterpart: the tests use it to show that a case ran on the path it is meant to cover):
 * h_counts[4] = {write calls served by the order a lookup prepared, write calls that tried such
 * an order and fell back to the full path, lookups served by a mono_table_lookup_hint record,
 * non-empty hint queues dropped}. */
int mono_table_path_counts(void* table, int64_t* h_counts);
/* Floats per stored row: dim + optimizer state. */
int mono_table_row_floats(void* table);

/* ops/hash_table_lookup_op.cc:148-251 (MonolithHashTableFusedLookup). h_fused_slot_size is a
 * host int32[K*N]; the four int32 metadata outputs are host arrays; out_emb is a device buffer
 * of at least sum(size*dim) floats (call with out_emb == NULL to get the metadata only). */
int mono_tables_fused_lookup(void* const* tables, int K, const int64_t* ids,
                             const int32_t* h_fused_slot_size, int N, float* out_emb,
                             int32_t* h_emb_splits, int32_t* h_id_offsets, int32_t* h_emb_offsets,
                             int32_t* h_emb_sizes, void* stream);
/* The lookup hint (mono_table_lookup_hint) for the fused lookup that is `ahead` calls away: same
 * ids / h_fused_slot_size / N as mono_tables_fused_lookup will be given. No counterpart in the
 * reference (its prefetch queues, distributed_ps_sync.py:218-224, run the whole lookup ahead and
 * accept stale rows; this resolves slots ahead and gathers rows in line, so nothing is stale).
 * Takes effect for K == 1; a no-op otherwise. The ids must stay valid until that lookup is issued. */
int mono_tables_fused_lookup_hint(void* const* tables, int K, const int64_t* ids,
                                  const int32_t* h_fused_slot_size, int N, int ahead, void* stream);
/* ops/hash_table_update_op.cc:174-262 (MonolithHashTableFusedOptimize). */
int mono_tables_fused_apply_gradient(void* const* tables, void* const* filters, int K,
                                     const int64_t* ids, const int32_t* h_fused_slot_size,
                                     const float* grads, const int32_t* h_id_offsets,
                                     const int32_t* h_grad_offsets, const float* h_learning_rates,
                                     const int32_t* h_lr_lengths, int64_t req_time,
                                     int64_t global_step, int N, int enable_grad_accumulation,
                                     void* stream);

/* ---- routing / exchange ops ----
 * SURVEY.md 8(b) also sketches mono_comm_init / mono_a2a_ids / mono_a2a_rows / mono_a2a_grads. They
 * are deliberately NOT part of this ABI: the four all-to-alls of a step (reference
 * distributed_ps_sync.py:170-193,230-283,369-496, hvd.alltoall / bps.alltoall) are issued by the
 * host layer through torch.distributed.all_to_all_single, whose "nccl" backend IS RCCL over xGMI;
 * this library carries no communication dependency (INTEGRATION.md 4 list the exact
 * torch calls). What the exchange needs from the device side is below: the reorder that produces
 * the send buffers and split sizes, the gathers on either side of the row exchange. */
/* ops/fused_reorder_by_indices.cc:36-112 + distribution_ops.py:255-272. inputs: M device id
 * arrays (host array of device pointers); shard_rule 0 = floormod(id, N), 1 = 1 + floormod(id, N-1).
 * out_ids has room for sum(lens) ids; *h_n_unique = ids written; out_offsets: M device int32
 * arrays shaped like the inputs; h_shard_sizes[N], h_sharded_slot_sizes[N*M] host. Returns as soon
 * as the sizes have reached the host: out_ids / out_offsets are complete in STREAM order (work
 * queued on `stream` afterwards sees them), not necessarily when the call returns. */
int mono_fused_reorder_by_indices(const int64_t* const* inputs, const int64_t* h_lens, int M, int N,
                                  const int32_t* h_dims, int shard_rule, int64_t* out_ids,
                                  int64_t* h_n_unique, int32_t* h_shard_sizes,
                                  int32_t* h_sharded_slot_sizes, int32_t* const* out_offsets,
                                  void* stream);
/* The same in two halves, for callers that compute the reorder ahead of the step that consumes it
 * (the reference does so in its input pipeline: distributed_ps_sync.py:313-331,
 * early_reorder_indicies_res_pack). begin launches on `stream` and returns; finish blocks until the
 * sizes are on the host and reports them. One reorder may be pending at a time; the gradient plan
 * token (below) is valid after finish. begin may be given a side stream: out_ids / out_offsets are
 * complete in THAT stream's order (make the consumer's stream wait for an event recorded on it). */
int mono_fused_reorder_begin(const int64_t* const* inputs, const int64_t* h_lens, int M, int N,
                             const int32_t* h_dims, int shard_rule, int64_t* out_ids,
                             int32_t* const* out_offsets, void* stream);
int mono_fused_reorder_finish(int64_t* h_n_unique, int32_t* h_shard_sizes,
                              int32_t* h_sharded_slot_sizes);
/* ShardingSparseFidsV2 on device-resident ragged FIDs: data/kernels/parse_sparse_feature.cc:173-211
 * (FillFidList: shard = uint64(fid) % ps_num), :213-288 (CreateOffsetTensor),
 * parse_sparse_feature.h:1472-1680 (FeatureParallelUniqueParse), data/parsers.py:402-456.
 * Protobuf parsing is NOT part of this entry point: fids[m] / row_splits[m] are the M configured
 * features in sorted-name order, row_splits[m] = int64[h_runs[m] + 1] relative to fids[m]
 * (h_runs[m] = batch, 1 for a feature shared by the batch, 0 for a feature absent from the input).
 * h_out_index[m * ps_num + shard] = FeatureInfo::GetFidOutputIndex (parse_sparse_feature.h:135-142),
 * all < n_out. Outputs: out_fid_list = the FIDs grouped by out index (ascending), each group in
 * first-occurrence order (unique) or input order; h_out_counts[n_out] = group sizes (host);
 * out_fid_offset[k] = (out index << 32) | index of FID k inside its group; out_feature_offset[j] =
 * first FID position of run j. Synchronises the stream. */
int mono_sharding_sparse_fids(const int64_t* const* fids, const int64_t* const* row_splits,
                              const int64_t* h_lens, const int64_t* h_runs, int M, int ps_num,
                              const int32_t* h_out_index, int n_out, int unique,
                              int64_t* out_fid_list, uint64_t* out_fid_offset,
                              int32_t* out_feature_offset, int32_t* h_out_counts, void* stream);
/* ops/map_id_to_embedding_op.cc:239-292 / map_id_to_embedding.cu.cc:30-74. */
int mono_fused_gather_embeddings_by_input(const float* fused, int64_t fused_len,
                                          const int32_t* const* offsets, const int64_t* h_lens,
                                          const int32_t* h_dims, int M, float* const* outs,
                                          void* stream);
/* ops/map_id_to_embedding_op.cc:294-343 (CPU order: deterministic, in input order). */
int mono_fused_gather_embeddings_by_input_gradient(int64_t fused_len, const float* const* grads,
                                                   const int32_t* const* offsets,
                                                   const int64_t* h_lens, const int32_t* h_dims,
                                                   int M, float* out, void* stream);
/* The same gradient with the order of summation prepared ahead of time. mono_fused_reorder_by_indices
 * sorts and classifies the offsets it has just written on a stream of its own, while the host
 * waits for its sizes; mono_fused_reorder_last_plan() returns the token of that plan (0 = none).
 * The call below uses the plan when `plan` names one of the two most recent plans AND offsets /
 * h_lens / h_dims are the very arrays that reorder wrote (same device pointers, unmodified since);
 * otherwise it does what the unplanned call does. Same result either way (same input order).
 * No counterpart in the reference: its gradient op recomputes nothing because it scatters with
 * atomics (map_id_to_embedding.cu.cc:76-119); here the sum is ordered, and the plan takes the
 * ordering off the backward pass. */
int64_t mono_fused_reorder_last_plan(void);
int mono_fused_gather_embeddings_by_input_gradient_planned(int64_t plan, int64_t fused_len,
                                                           const float* const* grads,
                                                           const int32_t* const* offsets,
                                                           const int64_t* h_lens,
                                                           const int32_t* h_dims, int M, float* out,
                                                           void* stream);

/* ---- pooling ---- */
/* ops/reduce_op.cc:29-51 / :55-87: out[batch, dim] zero-filled then summed in input order. */
int mono_reduce_sum(const int64_t* row_ids, const float* values, int64_t n, int64_t dim,
                    int64_t batch, float* out, void* stream);
int mono_reduce_mean(const int64_t* row_ids, const float* values, int64_t n, int64_t dim,
                     int64_t batch, float* out, void* stream);
/* No counterpart in the reference: mono_reduce_sum for callers that cannot allocate device memory
 * (tight serving loops, harnesses that forbid hipMalloc; NOT for use under HIP graph capture: like
 * mono_reduce_sum it reads a flag word back and synchronises the stream). Same
 * arguments, same result, same errors; its scratch (sortedness flag, (row, position) pairs, radix
 * sort temporaries) is a fixed 16 MiB __device__ array of the library, one copy per GPU, so the
 * call never calls hipMalloc / hipFree or any allocating routine. Limits: n <= 524288 row ids per
 * call (larger calls fail with a message naming mono_reduce_sum). All calls on one GPU share that
 * array: calls that overlap on one GPU (two streams, two host threads) race; issue them one after
 * the other. Like mono_reduce_sum it reads one flag word back (synchronises the stream). */
int mono_reduce_sum_noalloc(const int64_t* row_ids, const float* values, int64_t n, int64_t dim,
                            int64_t batch, float* out, void* stream);
/* ops/reduce_op.cc:124-148 / :150-180. */
int mono_reduce_sum_gradient(const int64_t* row_ids, const float* grads, int64_t n, int64_t dim,
                             int64_t batch, float* out, void* stream);
int mono_reduce_mean_gradient(const int64_t* row_ids, const float* grads, int64_t n, int64_t dim,
                              int64_t batch, float* out, void* stream);
/* ops/reduce_op.cc:232-277 / :279-321. outs / grads: M device pointers. */
int mono_fused_reduce_sum_and_split(const int64_t* row_ids, const float* values, int64_t n,
                                    int64_t dim, int64_t batch, const int32_t* h_split_dims, int M,
                                    float* const* outs, void* stream);
int mono_fused_reduce_sum_and_split_gradient(const int64_t* row_ids, const float* const* grads,
                                             int64_t n, int64_t batch, const int32_t* h_split_dims,
                                             int M, float* out, void* stream);
/* ops/reduce_op.cu.cc:53-161 (MonolithFusedSegmentSum): N sorted segment sums in one call. */
int mono_fused_sorted_segment_sum(const int32_t* const* indices, const float* const* values,
                                  const int64_t* h_lens, const int64_t* h_dims,
                                  const int64_t* h_out_rows, int N, float* const* outs,
                                  void* stream);
/* embedding_combiners.py:71-100 (FirstN): out[batch, max_seq, dim], zero padded. row_ids sorted. */
int mono_first_n(const int64_t* row_ids, const float* values, int64_t n, int64_t dim,
                 int64_t batch, int64_t max_seq, float* out, void* stream);

/* ---- checkpoint framing (host side) ----
 * The reference writes table dumps with TensorFlow's io::RecordWriter (ops/hash_table_save_op.cc:
 * 128-165; TensorFlow is a third-party dependency pinned by third_party/org_tensorflow and absent
 * from the tree): every record is  uint64 length | masked crc32c(length) | data | masked
 * crc32c(data), mask(c) = ((c >> 15) | (c << 17)) + 0xa282ead8. Pinned by the dump files the
 * reference ships (model_export/testdata/saved_model/.../assets). Plain host memory, no GPU. */
uint32_t mono_crc32c(const void* data, int64_t n);
/* frames n equal-length payloads back to back into out (n * (16 + payload_len) bytes); returns the
 * bytes written */
int64_t mono_tfrecord_frame(const uint8_t* payloads, int64_t n, int64_t payload_len, uint8_t* out);
/* walks a record file image: payload offsets / lengths of up to max_records records; returns the
 * record count, -1 on a framing or (verify != 0) crc error */
int64_t mono_tfrecord_scan(const uint8_t* buf, int64_t len, int64_t* offsets, int64_t* lengths,
                           int64_t max_records, int verify);

/* ---- layout fusion ---- */
/* One SliceConfig of one layout (idl/matrix/proto/example.proto SliceConfig / OutConfig), flattened
 * by the host exactly as MonolithEmbeddingToLayoutBase's constructor and the Layout classes do
 * (ops/fused_embedding_to_layout.cc:67-334): layouts in sorted-name order, slices in config order. */
enum { MONO_POOL_SUM = 0, MONO_POOL_MEAN = 1, MONO_POOL_FIRSTN = 2 };
typedef struct mono_layout_slice {
  int32_t feature_idx; /* index among the sorted feature names = index into nfl_offset */
  int32_t start, dim;  /* embedding columns [start, start + dim) */
  int32_t pooling;     /* MONO_POOL_* */
  int32_t max_seq;     /* FIRSTN: max_sequence_length */
  int32_t out_index;   /* which output tensor (sorted layouts; NONE layouts: one per slice) */
  int32_t out_stride;  /* floats per sample in that tensor (PtrWrapper::offset) */
  int32_t out_offset;  /* float offset of the slice inside a sample (GetSlice) */
  int32_t addn;        /* 1: the layout is ADDN - its slices accumulate in order (:553-583) */
  int32_t first;       /* 1: first slice of its layout (slice_conf_i == 0) */
} mono_layout_slice;

/* ops/fused_embedding_to_layout.cc:403-695 (MonolithEmbeddingToLayout / V2). h_emb_ptrs[i] is
 * the i-th embedding buffer a fid_offset's high word indexes (V2: already cut per feature by
 * fid_list_row_split, :438-476), row stride h_emb_dims[i], h_emb_floats[i] floats long.
 * fid_offset / feature_offset are device arrays; nfl_offset (bit 31 = shared) is a host array.
 * Outputs are zero-filled, then written. */
int mono_embedding_to_layout(const mono_layout_slice* h_slices, int n_slices,
                             const float* const* h_emb_ptrs, const int32_t* h_emb_dims,
                             const int64_t* h_emb_floats, int n_bufs, const uint64_t* fid_offset,
                             int64_t n_fids, const int32_t* feature_offset, int64_t n_features,
                             const uint32_t* h_nfl_offset, int n_nfl, int32_t batch,
                             float* const* h_outs, const int64_t* h_out_floats, int n_outs,
                             void* stream);
/* The fused form of "look up, then pool": PartitionedHashTable.lookup (distributed_ps.py:889-964)
 * calls table.lookup per (table, ps) FID list (ops/hash_table_lookup_op.cc:36-73) and hands the
 * [n_unique, dim] results to MonolithEmbeddingToLayout (ops/fused_embedding_to_layout.cc:403-695,
 * GatherEmb :624-676). Here buffer i of the fid_offset high word is named by the table and the
 * FID list instead (h_tables[i], device list h_fid_lists[i] of h_fid_counts[i] ids: V2 order, one
 * list per (table, ps, feature)); rows are gathered from the table and pooled in one pass, an id
 * the table does not hold contributes a row of zeros (lookup semantics), nothing of size
 * [n_unique, dim] is written. Same outputs, bit for bit, as mono_table_lookup per list followed by
 * mono_embedding_to_layout. */
int mono_lookup_to_layout(const mono_layout_slice* h_slices, int n_slices, void* const* h_tables,
                          const int64_t* const* h_fid_lists, const int64_t* h_fid_counts, int n_bufs,
                          const uint64_t* fid_offset, int64_t n_fids, const int32_t* feature_offset,
                          int64_t n_features, const uint32_t* h_nfl_offset, int n_nfl, int32_t batch,
                          float* const* h_outs, const int64_t* h_out_floats, int n_outs,
                          void* stream);
/* ops/fused_embedding_to_layout.cc:703-976 (MonolithEmbeddingToLayoutGrad / V2). h_emb_grads[i]
 * (shaped like embedding buffer i) are zero-filled, then every (unique fid, slice) receives the
 * layout gradients of all its uses, summed in the reference's order (layout, slice, sample, fid).
 * mean_divides = 0 reproduces the reference (MEAN scatters the undivided gradient, :929-937).
 * Copies feature_offset to the host (synchronises the stream). */
int mono_embedding_to_layout_grad(const mono_layout_slice* h_slices, int n_slices,
                                  float* const* h_emb_grads, const int32_t* h_emb_dims,
                                  const int64_t* h_emb_floats, int n_bufs,
                                  const uint64_t* fid_offset, int64_t n_fids,
                                  const int32_t* feature_offset, int64_t n_features,
                                  const uint32_t* h_nfl_offset, int n_nfl, int32_t batch,
                                  const float* const* h_out_grads, const int64_t* h_out_floats,
                                  int n_outs, int mean_divides, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MONOLITH_HIP_H_ */
