"""Scenario back-end over the HIP path (monolith_amd.hash_table_ops through the C ABI)."""
import itertools

import numpy as np
import torch

from monolith_amd import entry, hash_filter_ops, hash_table_ops

_counter = itertools.count()


def _opt(d):
  if d.get("sr16"):
    return entry.StochasticRoundingFloat16OptimizerWrapper(_opt(dict(d, sr16=False)))
  k = d.get("opt", "sgd")
  if k == "sgd":
    return entry.SgdOptimizer()
  if k == "adagrad":
    return entry.AdagradOptimizer(initial_accumulator_value=d.get("init_acc", 0.1),
                                  weight_decay_factor=d.get("weight_decay", 0.0))
  if k == "momentum":
    return entry.MomentumOptimizer(weight_decay_factor=d.get("weight_decay", 0.0),
                                   use_nesterov=d.get("use_nesterov", False),
                                   momentum=d.get("momentum", 0.9))
  if k in ("rmsprop", "rmspropv2"):
    cls = entry.RmspropOptimizer if k == "rmsprop" else entry.RmspropV2Optimizer
    return cls(learning_rate=d.get("learning_rate", 0.01),
               weight_decay_factor=d.get("weight_decay", 0.0), momentum=d.get("momentum", 0.9))
  if k in ("adam", "amsgrad"):
    cls = entry.AdamOptimizer if k == "adam" else entry.AmsgradOptimizer
    return cls(beta1=d.get("beta1", 0.9), beta2=d.get("beta2", 0.99),
               weight_decay_factor=d.get("weight_decay", 0.0),
               use_nesterov=d.get("use_nesterov", False), epsilon=d.get("epsilon", 0.01))
  if k == "moving_average":
    return entry.MovingAverageOptimizer(momentum=d.get("momentum", 0.9))
  if k == "batch_softmax":
    return entry.BatchSoftmaxOptimizer()
  if k == "adadelta":
    return entry.AdadeltaOptimizer(weight_decay_factor=d.get("weight_decay", 0.0),
                                   averaging_ratio=d.get("averaging_ratio", 0.9),
                                   epsilon=d.get("epsilon", 0.01))
  return entry.FtrlOptimizer(initial_accumulator_value=d.get("init_acc", 0.1),
                             beta=d.get("beta", 0.0), l1_regularization=d.get("l1", 0.0),
                             l2_regularization=d.get("l2", 0.0))


def _init(d):
  k = d.get("init", "zeros")
  if k == "zeros":
    return entry.ZerosInitializer()
  if k == "ones":
    return entry.OnesInitializer()
  if k == "constant":
    return entry.ConstantsInitializer(d.get("init_a", 0.0))
  return entry.RandomUniformInitializer(d.get("init_a", -0.05), d.get("init_b", 0.05))


def _comp(d):
  k = d.get("comp", "fp32")
  if k == "fp16":
    return entry.Fp16Compressor()
  if k == "fixed_r8":
    return entry.FixedR8Compressor(d.get("comp_a", 1.0))
  if k == "one_bit":
    return entry.OneBitCompressor(amplitude=d.get("comp_a", 0.05), init_scale=d.get("comp_b", 1.0))
  return entry.Fp32Compressor()


def table_config(cfg):
  tc = entry.EmbeddingHashTableConfig(
      segments=[entry.CombineAsSegment(s["dim"], _init(s), _opt(s), _comp(s))
                for s in cfg["segments"]],
      initial_capacity=cfg.get("initial_capacity", 1), seed=cfg.get("seed", 0),
      entry_type="serving" if cfg.get("serving") else "training")
  tc.slot_expire_time_config.default_expire_time = cfg.get("default_expire_days", 36500)
  tc.slot_expire_time_config.slot_expire_times = {
      int(k): v for k, v in cfg.get("slot_expire", {}).items()}
  if "evict_every_n_hours" in cfg:  # CuckooHashTableConfig(feature_evict_every_n_hours=...)
    tc.enable_feature_eviction = True
    tc.feature_evict_every_n_hours = cfg["evict_every_n_hours"]
  return tc


class HipTable:
  """numpy-in / numpy-out adapter with the oracle's method signatures."""

  def __init__(self, cfg, filt=None):
    tc = table_config(cfg)
    self.n_seg = len(tc.segments)
    self.lrs = [1.0] * self.n_seg
    inst = entry.HashTableConfigInstance(tc, [(lambda i=i: self.lrs[i])
                                              for i in range(self.n_seg)])
    self.default_filter = filt
    self.t = hash_table_ops.hash_table_from_config(inst, hash_filter=filt,
                                                   name_suffix="scn%d" % next(_counter))
    self.dim = self.t.dim_size
    self.row_floats = self.t.table.row_floats

  def _with_filter(self, filt):
    # the reference binds the filter at table creation; scenarios pass the same object
    assert filt is None or filt is self.default_filter
    return self.t

  def lookup(self, ids):
    return self.t.lookup(np.asarray(ids, np.int64)).cpu().numpy()

  def lookup_hint(self, ids, ahead=1):
    self.t.lookup_hint(np.asarray(ids, np.int64), ahead)

  def assign(self, ids, values, req_time=0, filt=None):
    self.t = self._with_filter(filt).assign(np.asarray(ids, np.int64), np.asarray(values, np.float32),
                                            req_time)

  def assign_add(self, ids, values, req_time=0, filt=None):
    self.t = self._with_filter(filt).assign_add(np.asarray(ids, np.int64),
                                                np.asarray(values, np.float32), req_time)

  def apply_gradients(self, ids, grads, lrs, req_time=0, global_step=0, enable_dedup=False,
                      filt=None):
    self.lrs = [float(x) for x in np.asarray(lrs).reshape(-1)]
    self.t = self._with_filter(filt).apply_gradients(np.asarray(ids, np.int64),
                                                     np.asarray(grads, np.float32), global_step,
                                                     enable_dedup=enable_dedup, req_time=req_time)

  def size(self):
    return self.t.size()

  def contains(self, ids):
    return self.t.contains(np.asarray(ids, np.int64)).cpu().numpy().astype(np.uint8)

  def clear(self):
    self.t.clear()

  def evict(self, t):
    self.t.evict(t)

  def max_update_ts(self):
    return self.t.max_update_ts()

  def export(self, apply_expiry=False):
    ids, rows, ts = self.t.save_as_tensor(apply_expiry)
    ids, rows, ts = ids.cpu().numpy(), rows.cpu().numpy(), ts.cpu().numpy().view(np.uint32)
    order = np.argsort(ids, kind="stable")
    return ids[order], rows[order], ts[order]


_last_filter = [None]


def make_filter(cfg):
  thr = entry.SlotOccurrenceThresholdConfig(
      {int(k): v for k, v in cfg.get("slot_thresholds", {}).items()},
      cfg.get("default_threshold", 0))
  if cfg.get("kind", "sliding") == "dummy":
    f = hash_filter_ops.create_dummy_hash_filter()
  else:
    f = hash_filter_ops.create_hash_filter(cfg.get("capacity", 300000000), cfg.get("split_num", 7),
                                           thr)
  _last_filter[0] = f
  return f


def make_table(cfg, filt=None):
  # scenarios create the filter first, then the table bound to it
  f = filt if filt is not None else _last_filter[0]
  _last_filter[0] = None
  return HipTable(cfg, f)
