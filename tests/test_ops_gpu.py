"""GPU parity of the routing / pooling / filter ops: the reference's own test vectors replayed
through libmonolith_hip.so (via the Python op surface) and seeded random inputs compared
bit-for-bit with the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import cpu
from tests.test_oracle_golden import REORDER_CASES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
  from monolith_amd import distribution_ops, embedding_combiners, hash_filter_ops
  return distribution_ops, embedding_combiners, hash_filter_ops


def dev():
  return torch.device("cuda", 0)


def T(a, dtype):
  return torch.tensor(np.asarray(a), dtype=dtype, device=dev())


def bits(x):
  if isinstance(x, torch.Tensor):
    x = x.detach().cpu().numpy()
  return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


# ----------------------------------------------------------------------------- routing
@pytest.mark.parametrize("case", REORDER_CASES, ids=lambda c: "reorder%d" % len(c[3]))
def test_fused_reorder_reference_vectors(ops, case):
  """distribution_ops_fused_test.py:52-119."""
  d_ops = ops[0]
  inputs, n, dims, out, sizes, sss, offs = case
  dims_ = dims or [1] * len(inputs)
  o, s, ss, of = d_ops.fused_reorder_by_indices([T(x, torch.int64) for x in inputs], n, dims_)
  assert o.cpu().tolist() == out and s.tolist() == sizes and ss.tolist() == sss
  if offs:
    assert [x.cpu().tolist() for x in of] == offs


@pytest.mark.parametrize("n_shards,rank0", [(2, "1"), (3, "1"), (8, "1"), (8, "0"), (5, "1")])
def test_fused_reorder_random_vs_oracle(ops, monkeypatch, n_shards, rank0):
  d_ops = ops[0]
  monkeypatch.setenv("MONOLITH_SYNC_EMPTY_RANK0_PS_SHARD", rank0)
  rule = 1 if (n_shards > 4 and rank0 == "1") else 0
  rng = np.random.Generator(np.random.PCG64(n_shards * 7 + int(rank0)))
  lens = [0, 1, 37, 5000, 20000, 3]
  inputs = [(rng.integers(-50, max(60, l // 3), l) * 1000003).astype(np.int64) for l in lens]
  # mixed dims (offsets = exclusive scan of the dims) and one dim for all (offsets = rank * dim)
  for dims in ([4, 1, 16, 8, 3, 2], [16] * 6):
    o, s, ss, of = d_ops.fused_reorder_by_indices([T(x, torch.int64) for x in inputs], n_shards, dims)
    eo, es, ess, eof = cpu.fused_reorder_by_indices(inputs, n_shards, dims, shard_rule=rule)
    assert np.array_equal(o.cpu().numpy(), eo)
    assert np.array_equal(s.numpy(), es) and np.array_equal(ss.numpy(), ess)
    for a, b in zip(of, eof):
      assert np.array_equal(a.cpu().numpy().reshape(-1), b)


@pytest.mark.parametrize("rank0", ["0", "1"])
def test_reorder_config4_routing(ops, monkeypatch, rank0):
  """BASELINE config 4's routing: one merged dim-64 slot, 1.2e5 skewed FIDs (bench.py's stream),
  8 shards, plain floormod(fid, 8) and the reference-default rank-0-empty rule
  1 + floormod(fid, 7) (distribution_ops.py:255-272): ids, shard sizes and offsets against the
  oracle, bit for bit."""
  d_ops = ops[0]
  monkeypatch.setenv("MONOLITH_SYNC_EMPTY_RANK0_PS_SHARD", rank0)
  rng = np.random.Generator(np.random.PCG64(4400 + int(rank0)))
  slot = rng.integers(0, 26, 120000).astype(np.int64)
  ids = (slot << 48) | (40000 * rng.random(120000) ** 3).astype(np.int64)
  o, s, ss, of = d_ops.fused_reorder_by_indices([T(ids, torch.int64)], 8, [64])
  eo, es, ess, eof = cpu.fused_reorder_by_indices([ids], 8, [64], shard_rule=int(rank0))
  assert np.array_equal(o.cpu().numpy(), eo)
  assert np.array_equal(s.numpy(), es) and np.array_equal(ss.numpy(), ess)
  assert np.array_equal(of[0].cpu().numpy().reshape(-1), eof[0])
  if rank0 == "1":
    assert s.numpy()[0] == 0  # rank 0 holds no rows under the reference-default rule


def test_fused_gather_and_gradient(ops):
  """distribution_ops_fused_test.py:113-119 layout + random case; the gradient accumulates
  duplicates in input order (ops/map_id_to_embedding_op.cc:294-343) -> bit-exact."""
  d_ops = ops[0]
  fused = np.array([1.1, 1.2, 1.3, 2.1, 2.2, 3.1, 3.2, 3.3, 4.1, 4.2, 4.3, 5.1, 5.2, 6.1, 6.2, 6.3,
                    7.1, 7.2, 8.1, 8.2, 9.1, 9.2], np.float32)
  offs = [[13, 0, 5, 13, 8, 13], [16, 18, 11, 11, 16, 20, 3]]
  outs = d_ops.fused_gather_embeddings_by_input(T(fused, torch.float32),
                                                [T(o, torch.int32) for o in offs], [3, 2])
  exp = cpu.fused_gather_embeddings_by_input(fused, offs, [3, 2])
  for a, b in zip(outs, exp):
    assert np.array_equal(bits(a), bits(b))
  g = d_ops.fused_gather_embeddings_by_input_gradient(
      len(fused), [torch.ones((6, 3), device=dev()), torch.ones((7, 2), device=dev())],
      [T(o, torch.int32) for o in offs], [3, 2]).cpu().numpy()
  assert g[13:16].tolist() == [3, 3, 3] and g[11:13].tolist() == [2, 2] and g[3:5].tolist() == [1, 1]

  rng = np.random.Generator(np.random.PCG64(5))
  dims = [8, 3, 16]
  rows = [300, 50, 2000]  # unique rows per slot inside the fused buffer
  base, fused_len = [], 0
  for r, d in zip(rows, dims):
    base.append(fused_len)
    fused_len += r * d
  fused = rng.normal(0, 1, fused_len).astype(np.float32)
  offs = [(b + rng.integers(0, r, n) * d).astype(np.int32)
          for b, r, d, n in zip(base, rows, dims, [5000, 7, 30000])]
  ft = T(fused, torch.float32).requires_grad_(True)
  outs = d_ops.fused_gather_embeddings_by_input(ft, [T(o, torch.int32) for o in offs], dims)
  exp = cpu.fused_gather_embeddings_by_input(fused, offs, dims)
  for a, b in zip(outs, exp):
    assert np.array_equal(bits(a), bits(b))
  grads = [rng.normal(0, 1, (len(o), d)).astype(np.float32) for o, d in zip(offs, dims)]
  torch.autograd.backward(outs, [T(g, torch.float32) for g in grads])
  eg = cpu.fused_gather_embeddings_by_input_gradient(fused_len, grads, offs, dims)
  assert np.array_equal(bits(ft.grad), bits(eg))

  # every dim a multiple of 4: the four-floats-per-thread gather; then the same through a fused
  # buffer that is not 16-byte aligned (scalar kernel again)
  dims, rows = [8, 4, 16], [300, 50, 2000]
  base, fused_len = [], 0
  for r, d in zip(rows, dims):
    base.append(fused_len)
    fused_len += r * d
  fused = rng.normal(0, 1, fused_len + 1).astype(np.float32)
  offs = [(b + rng.integers(0, r, n) * d).astype(np.int32)
          for b, r, d, n in zip(base, rows, dims, [5000, 7, 30001])]
  whole = T(fused, torch.float32)
  for shift in (0, 1):
    view = whole[shift:shift + fused_len]
    outs = d_ops.fused_gather_embeddings_by_input(view, [T(o, torch.int32) for o in offs], dims)
    exp = cpu.fused_gather_embeddings_by_input(fused[shift:shift + fused_len], offs, dims)
    for a, b in zip(outs, exp):
      assert np.array_equal(bits(a), bits(b)), shift


# ----------------------------------------------------------------------------- pooling
@pytest.mark.parametrize("dims", [[16], [64], [8, 3, 16], [4, 4, 4]])
def test_gather_gradient_planned_by_reorder(ops, dims):
  """The summation order fused_reorder_by_indices prepares for the gradient of the gather (the
  token on its offsets) gives the unplanned result, bit for bit: hot ids (thousands of
  occurrences: the long-segment kernels), several slots, mixed dims, a fused buffer longer than
  the rows the ids cover (tail zero-filled), a plan used twice, and a plan displaced by two later
  reorders (falls back to the unplanned path)."""
  d_ops = ops[0]
  rng = np.random.Generator(np.random.PCG64(1234 + sum(dims)))
  M = len(dims)
  lens = [60000, 17, 9000][:M] if M > 1 else [150000]
  inputs = []
  for m, n in enumerate(lens):
    # a Zipf-like head: the hottest id has ~2500 occurrences (the long-segment kernels; below the
    # 4096 of the chunked sum, whose reassociation DESIGN.md 4.1 documents)
    x = (200000 * rng.random(n) ** 3).astype(np.int64)
    inputs.append(x * 1000003 + m)
  ids_t = [T(x, torch.int64) for x in inputs]
  o, s, ss, offs = d_ops.fused_reorder_by_indices(ids_t, 2, dims)
  assert getattr(offs, "plan_token", 0) != 0
  covered = int(sum(int(c) * dims[i % M] for i, c in enumerate(ss.tolist())))
  grads = [rng.normal(0, 0.1, (n, d)).astype(np.float32) for n, d in zip(lens, dims)]
  g_t = [T(g, torch.float32) for g in grads]
  offs_np = [x.cpu().numpy().reshape(-1) for x in offs]
  for fused_len in (covered, covered + 40):
    exp = cpu.fused_gather_embeddings_by_input_gradient(fused_len, grads, offs_np, dims)
    planned = d_ops.fused_gather_embeddings_by_input_gradient(fused_len, g_t, offs, dims)
    again = d_ops.fused_gather_embeddings_by_input_gradient(fused_len, g_t, offs, dims)
    plain = d_ops.fused_gather_embeddings_by_input_gradient(fused_len, g_t, list(offs), dims)
    assert np.array_equal(bits(planned), bits(exp))
    assert np.array_equal(bits(again), bits(exp))
    assert np.array_equal(bits(plain), bits(exp))
  # two more reorders displace the plan: the stale token is ignored
  for _ in range(2):
    d_ops.fused_reorder_by_indices([T(inputs[0][:100], torch.int64)], 2, [dims[0]])
  stale = d_ops.fused_gather_embeddings_by_input_gradient(covered, g_t, offs, dims)
  exp = cpu.fused_gather_embeddings_by_input_gradient(covered, grads, offs_np, dims)
  assert np.array_equal(bits(stale), bits(exp))


def test_combiners_reference_vectors(ops):
  """embedding_combiners_test.py:22-35."""
  _, comb, _ = ops
  key = comb.Ragged.from_lists([[1], [2, 3]], dev())
  out = comb.ReduceSum().combine(key, T([[1.0], [2.0], [3.0]], torch.float32))
  assert out.cpu().tolist() == [[1], [5]]
  key = comb.Ragged.from_lists([[1], [2, 3], [4, 5]], dev())
  out = comb.FirstN(2).combine(key, T([[1.0], [2.0], [3.0], [4.0], [5.0]], torch.float32))
  assert out.cpu().tolist() == [[[1], [0]], [[2], [3]], [[4], [5]]]
  key = comb.Ragged.from_lists([[1, 2, 3]], dev())
  out = comb.FirstN(2).combine(key, T([[1.0], [2.0], [3.0]], torch.float32))
  assert out.cpu().tolist() == [[[1], [2]]]


@pytest.mark.parametrize("sorted_rows", [True, False])
@pytest.mark.parametrize("dim", [1, 16, 33])
def test_reduce_ops_vs_oracle(ops, sorted_rows, dim):
  """reduce_op.cc:29-180: sum / mean and their gradients, sorted (the combiner case) and
  unsorted row ids, empty rows, in-order accumulation -> bit-exact."""
  d_ops = ops[0]
  rng = np.random.Generator(np.random.PCG64(dim + int(sorted_rows)))
  batch, n = 1000, 20000
  rows = rng.integers(0, batch, n)
  rows[rows % 17 == 3] = 5  # a long row, and rows that stay empty
  if sorted_rows:
    rows = np.sort(rows)
  v = rng.normal(0, 1, (n, dim)).astype(np.float32)
  rt = T(rows, torch.int64).unsqueeze(-1)
  for name in ("reduce_sum", "reduce_mean"):
    vt = T(v, torch.float32).requires_grad_(True)
    out = getattr(d_ops, name)(rt, vt, [batch])
    exp = getattr(cpu, name)(rows, v, batch)
    assert np.array_equal(bits(out), bits(exp)), name
    g = rng.normal(0, 1, (batch, dim)).astype(np.float32)
    out.backward(T(g, torch.float32))
    eg = getattr(cpu, name + "_gradient")(rows, g)
    assert np.array_equal(bits(vt.grad), bits(eg)), name + " gradient"


def test_reduce_mean_empty_row_is_nan(ops):
  """reduce_op.cc:59-86: 0 * (1 / 0) — an empty row is NaN, as in the reference."""
  d_ops = ops[0]
  v = np.array([[1, 2], [3, 4], [5, 7]], np.float32)
  out = d_ops.reduce_mean(T([0, 0, 2], torch.int64), T(v, torch.float32), 3).cpu().numpy()
  assert out[0].tolist() == [2, 3] and np.isnan(out[1]).all() and out[2].tolist() == [5, 7]


def test_reduce_sum_and_split_and_segment_sum(ops):
  """reduce_op.cc:232-320, reduce_op.cu.cc:53-161."""
  d_ops = ops[0]
  v = np.arange(12, dtype=np.float32).reshape(4, 3)
  outs = d_ops.fused_reduce_sum_and_split(T([0, 0, 1, 1], torch.int64), T(v, torch.float32), 2,
                                          [1, 2])
  assert outs[0].cpu().tolist() == [[3], [15]] and outs[1].cpu().tolist() == [[5, 7], [17, 19]]

  rng = np.random.Generator(np.random.PCG64(3))
  batch, n, split = 512, 30000, [1, 16, 8, 3]
  rows = np.sort(rng.integers(0, batch, n))
  v = rng.normal(0, 1, (n, sum(split))).astype(np.float32)
  vt = T(v, torch.float32).requires_grad_(True)
  outs = d_ops.fused_reduce_sum_and_split(T(rows, torch.int64), vt, batch, split)
  exp = cpu.fused_reduce_sum_and_split(rows, v, batch, split)
  for a, b in zip(outs, exp):
    assert np.array_equal(bits(a), bits(b))
  gs = [rng.normal(0, 1, (batch, d)).astype(np.float32) for d in split]
  torch.autograd.backward(outs, [T(g, torch.float32) for g in gs])
  eg = cpu.fused_reduce_sum_and_split_gradient(rows, gs, split)
  assert np.array_equal(bits(vt.grad), bits(eg))

  idx = [np.sort(rng.integers(0, 100, 5000)).astype(np.int32), np.zeros(0, np.int32),
         np.sort(rng.integers(0, 7, 300)).astype(np.int32)]
  vals = [rng.normal(0, 1, (len(i), d)).astype(np.float32) for i, d in zip(idx, [4, 2, 16])]
  shapes = [[100, 4], [5, 2], [9, 16]]
  outs = d_ops.fused_sorted_segment_sum([T(i, torch.int32) for i in idx],
                                        [T(x, torch.float32) for x in vals], shapes)
  exp = cpu.fused_sorted_segment_sum(idx, vals, [s[0] for s in shapes])
  for a, b in zip(outs, exp):
    assert np.array_equal(bits(a), bits(b))


def test_first_n_vs_oracle(ops):
  _, comb, _ = ops
  rng = np.random.Generator(np.random.PCG64(9))
  lens = rng.integers(0, 9, 400)
  rows = [list(range(int(l))) for l in lens]
  key = comb.Ragged.from_lists(rows, dev())
  n = int(lens.sum())
  v = rng.normal(0, 1, (n, 6)).astype(np.float32)
  out = comb.FirstN(4).combine(key, T(v, torch.float32))
  exp = cpu.first_n(np.repeat(np.arange(400), lens), v, 400, 4)
  assert np.array_equal(bits(out), bits(exp))


# ----------------------------------------------------------------------------- filter
def test_filter_reference_sequences(ops):
  """sliding_hash_filter_test.cc:23-36,93-104 and hash_filter_ops_test.py:42-58."""
  from monolith_amd.entry import SlotOccurrenceThresholdConfig
  hf = ops[2]
  for key_num in (1, 3, 100):
    f = hf.create_hash_filter(key_num, 10)
    one = T([1], torch.int64)
    for i in range(16):
      assert int(f.add(one, T([1], torch.int32))[0]) == i
    assert int(f.add(one, T([1], torch.int32))[0]) == 15
    assert int(f.get(one)[0]) == 15

  f = hf.create_hash_filter(100, 7, SlotOccurrenceThresholdConfig(default_occurrence_threshold=3))
  ids = T([1, 3 << 17, 1], torch.int64)
  g = torch.ones((3, 2), device=dev())
  exp = [[[0, 0], [0, 0], [0, 0]], [[0, 0], [0, 0], [1, 1]], [[1, 1], [0, 0], [1, 1]],
         [[1, 1], [1, 1], [1, 1]]]
  for e in exp:
    emb = torch.zeros((3, 2), device=dev(), requires_grad=True)
    out = hf.intercept_gradient(f, ids, emb)
    out.backward(g)
    assert emb.grad.cpu().tolist() == e

  f0 = hf.create_hash_filter(1000000, 10, SlotOccurrenceThresholdConfig(default_occurrence_threshold=0))
  f1 = hf.create_hash_filter(1000000, 10, SlotOccurrenceThresholdConfig(default_occurrence_threshold=1))
  g = torch.ones((5, 1), device=dev())
  assert torch.equal(hf.intercept_gradient_gradient(f0, T(np.arange(5), torch.int64), g), g)
  assert torch.equal(hf.intercept_gradient_gradient(f1, T(np.arange(5) * 2, torch.int64), g),
                     torch.zeros_like(g))
  assert sum(f0.state()["num_elements"]) == 0 and sum(f1.state()["num_elements"]) == 5

  d = hf.create_dummy_hash_filter()
  g = torch.ones((3, 2), device=dev())
  assert torch.equal(hf.intercept_gradient_gradient(d, T([1, 2, 1], torch.int64), g), g)


def test_filter_rotation_replayed_per_id(ops):
  """hash_filter_ops_test.py:105-177 (capacity 300, split 100). The HIP filter rotates at batch
  boundaries (DESIGN.md deviation), so the reference's per-id sequence is replayed one id per
  call; counts and the final ring state then match the oracle exactly."""
  from monolith_amd.entry import SlotOccurrenceThresholdConfig
  hf = ops[2]
  f = hf.create_hash_filter(300, 100, SlotOccurrenceThresholdConfig(default_occurrence_threshold=2))
  o = cpu.OracleFilter("sliding", 300, 100, default_threshold=2)
  st = f.state()
  assert st["block_total_size"] == 3 and st["n_blocks"] == 100 and st["head"] == 0
  ids = [1, 1 << 17, 2 << 17, 1]
  for _ in range(4):
    for i in ids:
      g = torch.ones((1, 2), device=dev())
      got = hf.intercept_gradient_gradient(f, T([i], torch.int64), g).cpu().numpy()
      exp = o.intercept_gradient(np.array([i], np.int64), np.ones((1, 2), np.float32))
      assert np.array_equal(got, exp)
  so, sh = o.state(), f.state()
  assert sh["head"] == so["head"] and sh["num_elements"] == so["num_elements"]


def test_filter_rotates_past_every_block_a_batch_filled(ops):
  """One batch far above a block's capacity (sliding filter of 10 blocks x ~333 ids, 700 new
  ids at once): the head block and the one after it both fill inside the batch, and the
  batch-end rotation is repeated until the head is a block with room — the next batch must not
  start on a full head, and adds without failures. (Inside the oversized batch the adds are
  spread over the two blocks filter_add can reach and the overflow fails, which failure_count
  records; the reference rotates in mid-stream instead: DESIGN.md deviation 2.)"""
  from monolith_amd.entry import SlotOccurrenceThresholdConfig
  hf = ops[2]
  f = hf.create_hash_filter(3000, 10, SlotOccurrenceThresholdConfig(default_occurrence_threshold=2))
  rng = np.random.Generator(np.random.PCG64(31))
  big = (rng.permutation(1 << 20)[:700].astype(np.int64) << 17) + 5
  hf.intercept_gradient_gradient(f, T(big, torch.int64), torch.ones((700, 2), device=dev()))
  st1 = f.state()
  assert st1["head_increment"] >= 1, st1
  fails = st1["failure_count"]
  small = (rng.permutation(1 << 20)[:40].astype(np.int64) << 17) + 9
  hf.intercept_gradient_gradient(f, T(small, torch.int64), torch.ones((40, 2), device=dev()))
  st2 = f.state()
  assert st2["failure_count"] == fails, (st1, st2)
  assert sum(st2["num_elements"]) == sum(st1["num_elements"]) + 40, (st1, st2)


def test_filter_random_counts_vs_oracle(ops):
  """Batched add / get on ids with distinct signatures: old counts equal the sequential oracle."""
  hf = ops[2]
  rng = np.random.Generator(np.random.PCG64(21))
  f = hf.create_hash_filter(200000, 7)
  o = cpu.OracleFilter("sliding", 200000, 7)
  pool = (np.arange(1, 3001, dtype=np.int64) << 17) | 5
  for step in range(4):
    ids = rng.choice(pool, 5000)
    cnt = rng.integers(1, 4, 5000).astype(np.int32)
    got = f.add(T(ids, torch.int64), T(cnt, torch.int32)).cpu().numpy()
    exp = o.add(ids, cnt)
    assert np.array_equal(got, exp), step
    q = rng.choice(pool, 500)
    assert np.array_equal(f.get(T(q, torch.int64)).cpu().numpy(), o.get(q))


def test_sort_paths_small_fused_and_large_three_kernel():
  """sort.hip has two paths: one fused kernel per pass up to 512 tiles (4.19 M pairs) and
  hist / scan / scatter kernels beyond. fused_reorder_by_indices goes through the sort with
  n pairs: both sides of the threshold must give the reference's first-occurrence order."""
  import ctypes as C
  import torch
  from monolith_amd import _lib
  from monolith_amd import distribution_ops as dops
  for n in (512 * 8192 - 3, 512 * 8192, 512 * 8192 + 1, 5_000_000):
    g = torch.Generator(device="cuda")
    g.manual_seed(n)
    ids = torch.randint(0, 3_000_000, (n,), device="cuda", generator=g)
    out, shard_sizes, sss, offsets = dops.fused_reorder_by_indices([ids], 4, [1])
    out = out.cpu().numpy()
    ref = ids.cpu().numpy()
    uniq, first = np.unique(ref, return_index=True)
    order = uniq[np.argsort(first, kind="stable")]           # first-occurrence order
    exp = np.concatenate([order[order % 4 == s] for s in range(4)])
    assert np.array_equal(out, exp)
    assert np.asarray(shard_sizes).tolist() == [int((order % 4 == s).sum()) for s in range(4)]
    # offsets (dim 1) index the unique list
    assert np.array_equal(exp[offsets[0].cpu().numpy()], ref)
  fn = _lib.lib.mono_debug_sort_error
  fn.restype = C.c_int
  assert fn() == 0


def _sort_pairs(keys, bits, reps=1):
  import ctypes as C
  import torch
  from monolith_amd import _lib
  fn = _lib.lib.mono_debug_sort_pairs
  fn.restype = C.c_int
  n = keys.numel()
  ko = torch.empty(max(n, 1), dtype=torch.int32, device="cuda")
  vo = torch.empty(max(n, 1), dtype=torch.int32, device="cuda")
  ms = C.c_float(0)
  _lib.check(fn(_lib.vp(keys), None, C.c_int64(n), C.c_int(bits), _lib.vp(ko), _lib.vp(vo),
                C.c_int(reps), C.byref(ms), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
  return ko[:n].cpu().numpy().view(np.uint32), vo[:n].cpu().numpy().view(np.uint32), ms.value


def test_sort_pairs_two_level_lookback():
  """Up to 512 tiles the fused passes sum the tiles before a tile in two levels (groups of 16
  tiles), beyond that three kernels per pass: sizes on both sides of every boundary, skewed and
  constant keys, against numpy's stable argsort."""
  import ctypes as C
  import torch
  from monolith_amd import _lib
  rng = np.random.Generator(np.random.PCG64(5))
  T8 = 8192
  for n, bits in ((1, 9), (T8 + 1, 16), (15 * T8, 28), (16 * T8, 28), (16 * T8 + 1, 28),
                  (17 * T8 - 5, 20), (33 * T8 + 7, 29), (208 * T8, 28), (256 * T8, 7), (256 * T8 + 1, 23),
                  (511 * T8 + 9, 32), (512 * T8, 17), (513 * T8 + 5, 26)):
    k = rng.integers(0, 1 << bits, n, dtype=np.uint64).astype(np.uint32)
    if n > 100000:
      k[rng.integers(0, n, n // 3)] = k[0]               # one hot key
      k[n // 2: n // 2 + 30000] = np.uint32((1 << bits) - 1)
    ko, vo, _ = _sort_pairs(torch.from_numpy(k.view(np.int32)).cuda(), bits)
    order = np.argsort(k, kind="stable").astype(np.uint32)
    assert np.array_equal(vo, order), (n, bits)
    assert np.array_equal(ko, k[order]), (n, bits)
  k = np.full(40 * T8, 7, dtype=np.uint32)
  ko, vo, _ = _sort_pairs(torch.from_numpy(k.view(np.int32)).cuda(), 28)
  assert np.array_equal(vo, np.arange(k.size, dtype=np.uint32))
  fn = _lib.lib.mono_debug_sort_error
  fn.restype = C.c_int
  assert fn() == 0
