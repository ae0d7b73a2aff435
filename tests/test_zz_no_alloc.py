"""mono_reduce_sum_noalloc (include/monolith_hip.h): the entry point that allocates nothing. It has
no counterpart in the reference; its bar is the existing entry point it is a variant of,
mono_reduce_sum, bit for bit — on two calls in a row, the second with a smaller input, because
every call on a GPU shares one fixed scratch array that keeps the previous call's contents."""
import ctypes as C

import numpy as np
import pytest


def _reduce(fn, row_ids, values, batch):
  import torch
  from monolith_amd import _lib
  n, dim = values.shape
  out = torch.full((batch, dim), float("nan"), dtype=torch.float32, device=values.device)
  rc = fn(_lib.vp(row_ids), _lib.vp(values), C.c_int64(n), C.c_int64(dim), C.c_int64(batch),
          _lib.vp(out), _lib.stream_ptr(values.device))
  return rc, out


@pytest.mark.gpu
@pytest.mark.timeout(120)
def test_noalloc_matches_reduce_sum_on_two_calls_in_a_row():
  import torch
  from monolith_amd import _lib
  dev = torch.device("cuda", 0)
  rng = np.random.Generator(np.random.PCG64(20260101))
  # (n, dim, batch, sorted?) — the second call of each pair is smaller than the first
  for pair in ([(300000, 16, 4096, False), (70000, 16, 1500, False)],
               [(524288, 4, 65536, False), (1000, 4, 7, False)],
               [(200000, 8, 30000, True), (513, 3, 40, False)],
               [(8192, 32, 100, False), (0, 32, 5, False)]):
    for n, dim, batch, is_sorted in pair:
      rows = rng.integers(0, batch, n).astype(np.int64)
      if is_sorted:
        rows.sort()
      vals = rng.standard_normal((n, dim)).astype(np.float32)
      r = torch.from_numpy(rows).to(dev)
      v = torch.from_numpy(vals).to(dev).reshape(n, dim)
      rc0, want = _reduce(_lib.lib.mono_reduce_sum, r, v, batch)
      rc1, got = _reduce(_lib.lib.mono_reduce_sum_noalloc, r, v, batch)
      torch.cuda.synchronize()
      assert rc0 == 0 and rc1 == 0, _lib.lib.mono_last_error()
      assert torch.equal(want.view(torch.int32), got.view(torch.int32)), (n, dim, batch)
      # input-order sums: row 0 against numpy's sequential float32 accumulation
      acc = np.zeros(dim, np.float32)
      for i in np.nonzero(rows == 0)[0]:
        acc = acc + vals[i]
      np.testing.assert_array_equal(got[0].cpu().numpy(), acc)


@pytest.mark.gpu
@pytest.mark.timeout(120)
def test_noalloc_rejects_what_its_scratch_cannot_serve():
  import torch
  from monolith_amd import _lib
  dev = torch.device("cuda", 0)
  n = (1 << 19) + 1
  r = torch.zeros(n, dtype=torch.int64, device=dev)
  v = torch.zeros((n, 1), dtype=torch.float32, device=dev)
  rc, _ = _reduce(_lib.lib.mono_reduce_sum_noalloc, r, v, 4)
  assert rc != 0 and b"mono_reduce_sum" in _lib.lib.mono_last_error()
  bad = torch.tensor([0, 5, 2], dtype=torch.int64, device=dev)  # 5 is outside [0, 4)
  rc, _ = _reduce(_lib.lib.mono_reduce_sum_noalloc, bad, torch.ones((3, 2), device=dev), 4)
  assert rc != 0 and b"outside" in _lib.lib.mono_last_error()
  # and the entry point still works afterwards
  ok = torch.tensor([3, 0, 3], dtype=torch.int64, device=dev)
  rc, out = _reduce(_lib.lib.mono_reduce_sum_noalloc, ok, torch.ones((3, 2), device=dev), 4)
  assert rc == 0
  assert out.cpu().tolist() == [[1.0, 1.0], [0.0, 0.0], [0.0, 0.0], [2.0, 2.0]]
