"""Pooling combiners — the surface of monolith/native_training/embedding_combiners.py.

A ragged key batch (tf.RaggedTensor in the reference) is carried as `Ragged(values, row_splits)`:
`values` int64[n] FIDs, `row_splits` int64[batch + 1] on the same device."""
import abc
import ctypes as C
from typing import NamedTuple

import torch

from monolith_amd import _lib, distribution_ops
from monolith_amd._lib import vp


class Ragged(NamedTuple):
  values: torch.Tensor
  row_splits: torch.Tensor

  def nrows(self) -> int:
    return int(self.row_splits.numel()) - 1

  def value_rowids(self) -> torch.Tensor:
    """ragged_utils.fused_value_rowids (ragged_utils.py:23-40): row index of every value."""
    n = int(self.values.numel())
    pos = torch.arange(n, device=self.values.device, dtype=torch.int64)
    return torch.searchsorted(self.row_splits[1:].to(torch.int64).contiguous(), pos, right=True)

  @staticmethod
  def from_lists(rows, device):
    vals = [v for r in rows for v in r]
    splits = [0]
    for r in rows:
      splits.append(splits[-1] + len(r))
    return Ragged(torch.tensor(vals, dtype=torch.int64, device=device),
                  torch.tensor(splits, dtype=torch.int64, device=device))


class Combiner(abc.ABC):  # embedding_combiners.py:24-38

  def __init__(self, max_seq_length: int):
    self._max_seq_length = max_seq_length

  @property
  def max_seq_length(self):
    return self._max_seq_length

  @abc.abstractmethod
  def combine(self, key: Ragged, embedding: torch.Tensor, name: str = None):
    pass


class ReduceSum(Combiner):  # embedding_combiners.py:41-53

  def __init__(self):
    super().__init__(0)

  def combine(self, key: Ragged, embedding: torch.Tensor, name: str = None):
    return distribution_ops.reduce_sum(key.value_rowids().unsqueeze(-1), embedding, [key.nrows()],
                                       name=name)


class ReduceMean(Combiner):  # embedding_combiners.py:56-68

  def __init__(self):
    super().__init__(0)

  def combine(self, key: Ragged, embedding: torch.Tensor, name: str = None):
    return distribution_ops.reduce_mean(key.value_rowids().unsqueeze(-1), embedding,
                                        [key.nrows()])


class _FirstN(torch.autograd.Function):

  @staticmethod
  def forward(ctx, row_ids, embedding, batch, max_seq):
    if not embedding.is_cuda:
      raise _lib.MonolithHipError("FirstN: tensors must live on the GPU (no CPU fallback)")
    e = embedding.to(torch.float32).contiguous()
    n, dim = e.shape
    out = torch.empty((batch, max_seq, dim), dtype=torch.float32, device=e.device)
    _lib.check(_lib.lib.mono_first_n(vp(row_ids), vp(e), C.c_int64(n), C.c_int64(dim),
                                     C.c_int64(batch), C.c_int64(max_seq), vp(out),
                                     _lib.stream_ptr(e.device)))
    ctx.save_for_backward(row_ids)
    ctx.max_seq = max_seq
    return out

  @staticmethod
  def backward(ctx, grad):
    # gradient of scatter_nd + slice: value i takes grad[row, pos] when pos < max_seq, else 0
    (row_ids,) = ctx.saved_tensors
    n = int(row_ids.numel())
    idx = torch.arange(n, device=row_ids.device)
    first = torch.searchsorted(row_ids, row_ids, right=False)
    pos = idx - first
    keep = pos < ctx.max_seq
    g = torch.zeros((n, grad.shape[2]), dtype=grad.dtype, device=grad.device)
    g[keep] = grad[row_ids[keep], pos[keep]]
    return None, g, None, None


class FirstN(Combiner):  # embedding_combiners.py:71-100

  def __init__(self, seq_length: int):
    assert seq_length > 0, "seq_length must be greater than 0"
    super().__init__(seq_length)

  def combine(self, key: Ragged, embedding: torch.Tensor, name: str = None):
    """out[batch, seq_length, dim]: the first seq_length embeddings of every row, zero padded."""
    return _FirstN.apply(key.value_rowids().contiguous(), embedding, key.nrows(),
                         self.max_seq_length)
