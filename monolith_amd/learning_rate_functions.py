"""Learning-rate schedules evaluated on the host at every apply — the part of
monolith/native_training/learning_rate_functions.py the hot path's tests use.

The reference reads the TF global step from the graph; here the step is process state set by the
training loop through `set_global_step`."""
import abc

_GLOBAL_STEP = [0]


def set_global_step(step: int):
  _GLOBAL_STEP[0] = int(step)


def get_global_step() -> int:
  return _GLOBAL_STEP[0]


class LearningRateFunction(abc.ABC):  # learning_rate_functions.py:27-58

  @abc.abstractmethod
  def __call__(self) -> float:
    pass

  def __str__(self):
    return "%s(%s)" % (self.__class__.__name__, ", ".join(
        "%s=%r" % kv for kv in sorted(self.__dict__.items())))


class PolynomialDecay(LearningRateFunction):
  """learning_rate_functions.py:61-118 (tf.compat.v1.train.polynomial_decay)."""

  def __init__(self, initial_learning_rate, decay_steps, end_learning_rate=0.0001, power=1.0,
               cycle=False, name=None):
    self.initial_learning_rate = initial_learning_rate
    self.decay_steps = decay_steps
    self.end_learning_rate = end_learning_rate
    self.power = power
    self.cycle = cycle
    self.name = name

  def __call__(self) -> float:
    step = float(get_global_step())
    decay_steps = float(self.decay_steps)
    if self.cycle:
      import math
      mult = 1.0 if step == 0 else math.ceil(step / decay_steps)
      decay_steps = decay_steps * mult
    else:
      step = min(step, decay_steps)
    p = step / decay_steps
    return ((self.initial_learning_rate - self.end_learning_rate) * (1.0 - p) ** self.power +
            self.end_learning_rate)
