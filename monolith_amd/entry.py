"""Config builders mirroring monolith/native_training/entry.py:54-588 as plain dataclasses
(no protobuf): optimizers (Sgd / Adagrad / Ftrl / Momentum / Rmsprop / RmspropV2 / Adadelta), initializers (Zeros / Constants /
RandomUniform), CombineAsSegment, CuckooHashTableConfig, HashTableConfigInstance, and the
table / slot-expire / slot-threshold messages of embedding_hash_table.proto:19-106.
Defaults are the proto defaults (optimizer.proto:19-72, initializer_config.proto:19-46)."""
import copy
import dataclasses
from typing import Any, Callable, Dict, List, Optional, Union

OPT_SGD, OPT_ADAGRAD, OPT_FTRL = 0, 1, 2
OPT_MOMENTUM, OPT_RMSPROP, OPT_RMSPROP_V2, OPT_ADADELTA = 3, 4, 5, 6
OPT_ADAM, OPT_AMSGRAD = 7, 8
OPT_MOVING_AVERAGE = 9
OPT_BATCH_SOFTMAX = 10
INIT_ZEROS, INIT_ONES, INIT_CONSTANT, INIT_UNIFORM = 0, 1, 2, 3


class Optimizer:
  kind = None


@dataclasses.dataclass
class SgdOptimizer(Optimizer):  # entry.py:54-74
  learning_rate: Optional[Any] = None
  kind = OPT_SGD


@dataclasses.dataclass
class AdagradOptimizer(Optimizer):  # entry.py:77-112
  learning_rate: Optional[Any] = None
  initial_accumulator_value: Optional[float] = None
  hessian_compression_times: int = 1
  warmup_steps: int = 0
  weight_decay_factor: float = 0.0
  kind = OPT_ADAGRAD


@dataclasses.dataclass
class FtrlOptimizer(Optimizer):  # entry.py:342-369
  learning_rate: Optional[Any] = None
  initial_accumulator_value: Optional[float] = None
  beta: Optional[float] = None
  warmup_steps: int = 0
  l1_regularization: Optional[float] = None
  l2_regularization: Optional[float] = None
  kind = OPT_FTRL


@dataclasses.dataclass
class AdadeltaOptimizer(Optimizer):  # entry.py:112-131, optimizer.proto:90-97
  learning_rate: Optional[Any] = None
  weight_decay_factor: float = 0.0
  averaging_ratio: float = 0.9
  epsilon: float = 0.01
  warmup_steps: int = 0
  kind = OPT_ADADELTA


@dataclasses.dataclass
class MomentumOptimizer(Optimizer):  # entry.py:226-244, optimizer.proto:142-149
  learning_rate: Optional[Any] = None
  weight_decay_factor: float = 0.0
  use_nesterov: bool = False
  momentum: float = 0.9
  warmup_steps: int = 0
  kind = OPT_MOMENTUM


@dataclasses.dataclass
class RmspropOptimizer(Optimizer):  # entry.py:258-269, optimizer.proto:169-174
  """The kernel divides the CONFIG learning rate (proto default 0.01 when it is not a plain
  number), not the per-call learning-rate tensor (rmsprop_optimizer.cc:59)."""
  learning_rate: Optional[Any] = None
  weight_decay_factor: float = 0.0
  momentum: float = 0.9
  kind = OPT_RMSPROP


@dataclasses.dataclass
class RmspropV2Optimizer(Optimizer):  # entry.py:272-283, optimizer.proto:180-185
  learning_rate: Optional[Any] = None
  weight_decay_factor: float = 0.0
  momentum: float = 0.9
  kind = OPT_RMSPROP_V2


@dataclasses.dataclass
class AdamOptimizer(Optimizer):  # entry.py:135-172, optimizer.proto:104-114
  learning_rate: Optional[Any] = None
  beta1: float = 0.9
  beta2: float = 0.99
  use_beta1_warmup: bool = False
  weight_decay_factor: float = 0.0
  use_nesterov: bool = False
  epsilon: float = 0.01
  warmup_steps: int = 0
  kind = OPT_ADAM


@dataclasses.dataclass
class AmsgradOptimizer(Optimizer):  # entry.py:175-222, optimizer.proto:123-132
  learning_rate: Optional[Any] = None
  beta1: float = 0.9
  beta2: float = 0.99
  weight_decay_factor: float = 0.0
  use_nesterov: bool = False
  epsilon: float = 0.01
  warmup_steps: int = 0
  kind = OPT_AMSGRAD


@dataclasses.dataclass
class MovingAverageOptimizer(Optimizer):  # entry.py:247-255, optimizer.proto:155-158
  momentum: float = 0.9
  kind = OPT_MOVING_AVERAGE


@dataclasses.dataclass
class BatchSoftmaxOptimizer(Optimizer):  # entry.py:207-223, optimizer.proto:160-167 (dim 1)
  learning_rate: Optional[Any] = None   # the kernel reads the call's learning rate, like the reference
  kind = OPT_BATCH_SOFTMAX


class StochasticRoundingFloat16OptimizerWrapper(Optimizer):  # entry.py:42-50, optimizer.proto:213
  """Marks the wrapped optimizer's segment for fp16 stochastic rounding of the weights after
  every step (StochasticRoundingFloat16OptimizerDecorator, stochastic_rounding.h:39-56)."""

  def __init__(self, optimizer: Optimizer):
    self._optimizer = optimizer

  @property
  def kind(self):
    return self._optimizer.kind

  def __repr__(self):
    return "StochasticRoundingFloat16OptimizerWrapper(%r)" % (self._optimizer,)


class _NotImplementedOptimizer(Optimizer):
  """Optimizers whose reference factory throws `optimizer is not implemented yet`
  (group_ftrl_optimizer.cc:25-29, dynamic_wd_adagrad_optimizer.cc:25-29): a table built with one
  fails the same way here (hash_table_ops raises ValueError with the reference's text)."""
  kind = None

  def __init__(self, **kwargs):
    self.__dict__.update(kwargs)


class FTRLWithGroupSparsityOptimizer(_NotImplementedOptimizer):  # entry.py:286-306
  pass


class DynamicWdAdagradOptimizer(_NotImplementedOptimizer):  # entry.py:312-338
  pass


class Initializer:
  kind = None


@dataclasses.dataclass
class ZerosInitializer(Initializer):  # entry.py:381-388
  kind = INIT_ZEROS


@dataclasses.dataclass
class OnesInitializer(Initializer):
  kind = INIT_ONES


@dataclasses.dataclass
class ConstantsInitializer(Initializer):  # entry.py:391-400
  constant: float = 0.0
  kind = INIT_CONSTANT


@dataclasses.dataclass
class RandomUniformInitializer(Initializer):  # entry.py:403-418
  minval: Optional[float] = None
  maxval: Optional[float] = None
  kind = INIT_UNIFORM


@dataclasses.dataclass
class Fp32Compressor:  # entry.py:482-488 (training entries ignore the compressor)
  kind = 0


@dataclasses.dataclass
class Fp16Compressor:  # entry.py:472-479, float_compressor.proto:27-29
  kind = 1


@dataclasses.dataclass
class FixedR8Compressor:  # entry.py:459-469, float_compressor.proto:31-34
  fixed_range: float = 1.0
  kind = 2

  @property
  def r(self):
    return self.fixed_range


@dataclasses.dataclass
class OneBitCompressor:  # entry.py:444-456, float_compressor.proto:37-43
  step_size: int = 200
  amplitude: float = 0.05
  init_scale: float = 1.0
  kind = 3


@dataclasses.dataclass
class Segment:  # EntryConfig.Segment, embedding_hash_table.proto:20-28
  dim_size: int
  init_config: Initializer
  opt_config: Optimizer
  comp_config: Any = None


def CombineAsSegment(dim_size: int, initializer: Initializer, optimizer: Optimizer,
                     compressor: Any = None) -> Segment:  # entry.py:491-513
  return Segment(dim_size, initializer, optimizer, compressor)


@dataclasses.dataclass
class SlotExpireTimeConfig:  # embedding_hash_table.proto:52-62
  slot_expire_times: Dict[int, int] = dataclasses.field(default_factory=dict)
  default_expire_time: int = 36500


@dataclasses.dataclass
class SlotOccurrenceThresholdConfig:  # embedding_hash_table.proto:89-99
  slot_occurrence_thresholds: Dict[int, int] = dataclasses.field(default_factory=dict)
  default_occurrence_threshold: int = 0


@dataclasses.dataclass
class EmbeddingHashTableConfig:  # embedding_hash_table.proto:67-87
  segments: List[Segment] = dataclasses.field(default_factory=list)
  initial_capacity: int = 1
  slot_expire_time_config: SlotExpireTimeConfig = dataclasses.field(
      default_factory=SlotExpireTimeConfig)
  enable_feature_eviction: bool = False
  feature_evict_every_n_hours: int = 240
  seed: int = 0  # build-defined: seed of the counter-based random_uniform initializer
  # EntryConfig.entry_type (embedding_hash_table.proto:35-42): "serving" = values stored as the
  # segments' compressors return them, lookup / assign / assign_add / restore only
  # (hash_table_ops.py:465-466 switches to it when exporting)
  entry_type: str = "training"

  @property
  def dim_size(self):  # hash_table_utils.py:45-50 (infer_dim_size)
    return sum(s.dim_size for s in self.segments)

  def key(self):
    return repr(self)


class CuckooHashTableConfig:  # entry.py:525-537

  def __init__(self, initial_capacity=1, feature_evict_every_n_hours=0):
    self._initial_capacity = initial_capacity
    self._feature_evict_every_n_hours = feature_evict_every_n_hours

  def mutate_table(self, table_config: EmbeddingHashTableConfig):
    table_config.initial_capacity = self._initial_capacity
    if self._feature_evict_every_n_hours > 0:
      table_config.enable_feature_eviction = True
      table_config.feature_evict_every_n_hours = self._feature_evict_every_n_hours


class HashTableConfigInstance:  # entry.py:539-588
  """The config instance for generating HashTable."""

  def __init__(self, table_config: EmbeddingHashTableConfig,
               learning_rate_fns: List[Union[float, Callable[[], float]]],
               extra_restore_names=None):
    self._table_config = table_config
    self.extra_restore_names = copy.copy(extra_restore_names) or []
    self._learning_rate_fns = learning_rate_fns
    self._learning_rate_tensor = None

  def __str__(self):  # used to check whether two slots share the same config
    return "TableConfigPB: %s, LearningRateFns: [%s]" % (self._table_config.key(), ", ".join(
        [str(fn) for fn in self._learning_rate_fns]))

  @property
  def table_config(self):
    return self._table_config

  @property
  def learning_rate_fns(self):
    return self._learning_rate_fns

  @property
  def learning_rate_tensor(self):
    return self._learning_rate_tensor

  def set_learning_rate_tensor(self, learning_rate_tensor):
    self._learning_rate_tensor = learning_rate_tensor

  def call_learning_rate_fns(self) -> List[float]:
    """Evaluates the learning-rate functions (entry.py:571-588); one float per segment."""
    if not self._learning_rate_fns:
      raise Exception("Learning_rate_fns must be not empty.")
    return [float(fn() if callable(fn) else fn) for fn in self._learning_rate_fns]
