"""GPU: fused_embedding_to_layout (+ gradient) through the C ABI vs the oracle, bit-exact, on the
reference-test-shaped configuration (5 PS, 11 slots, ADDN / CONCAT / STACK / NONE / FIRSTN, shared
features, empty runs) and on a config-3-shaped pooling workload."""
import numpy as np
import pytest
import torch

from oracle import cpu
from tests import layout_cases as lc

pytestmark = pytest.mark.gpu

CASES = [("ref_shaped", lc.make_configs, dict(batch=10, ps_num=5, seed=2)),
         ("ref_shaped_b1", lc.make_configs, dict(batch=1, ps_num=2, seed=3)),
         ("no_shared", lc.make_configs, dict(batch=64, ps_num=3, seed=4, shared_every=0)),
         ("pooling", lc.make_pooling_configs, dict(batch=257, ps_num=4, seed=5, max_fids=80,
                                                   vocab=300, shared_every=0)),
         ("pooling_hot", lc.make_pooling_configs, dict(batch=512, ps_num=2, seed=6, max_fids=30,
                                                       vocab=5, shared_every=3)),
         # rows with ~15 000 uses each: the gradient's in-order chains run over several LDS tiles
         ("pooling_chain", lc.make_pooling_configs, dict(batch=8192, ps_num=1, seed=8, max_fids=12,
                                                         vocab=3, shared_every=0, empty_rate=0.05))]


def dev():
  return torch.device("cuda", 0)


def exact_order():
  """The default: every sum runs in the reference's order. MONO_CHUNK=<rows> (read once per process
  by the library) is the switch that sums longer destinations in chunks."""
  import os
  return os.environ.get("MONO_CHUNK", "0") in ("", "0")


def assert_gradient_rows(got, exp, uses, what):
  """got / exp: [rows, D] gradient rows, uses[r] = addends of row r. Every row is bit-exact. Under
  MONO_CHUNK=4096 (the optional switch) a row with more addends is summed as near-equal chunks of
  at most 4096, each in input order: it then differs from the sequential sum by reassociation
  only, bounded here by 2e-5 of the row's magnitude."""
  same = (got.view(np.uint32) == exp.view(np.uint32)).all(axis=1)
  if exact_order():
    assert same.all(), what
    return
  hot = uses > 4096
  assert same[~hot].all(), what
  if hot.any():
    mag = np.abs(exp[hot]).max(axis=1, keepdims=True)
    assert (np.abs(got[hot] - exp[hot]) <= 2e-5 * mag + 1e-12).all(), what


@pytest.mark.parametrize("name,mk,kw", CASES, ids=[c[0] for c in CASES])
def test_layout_forward_and_gradient_bit_exact(name, mk, kw):
  from monolith_amd import distribution_ops, feature_configs as fc
  cfgs = mk()
  inp = lc.make_inputs(cfgs, **kw)
  plan = fc.build_layout_plan(cfgs)
  embs = [torch.tensor(e, device=dev(), requires_grad=True) for e in inp["embeddings_list"]]
  splits = [torch.tensor(s) for s in inp["fid_list_row_split"]]
  fid_offset = torch.tensor(inp["fid_offset"].view(np.int64), device=dev())
  feature_offset = torch.tensor(inp["feature_offset"], device=dev())
  outs = distribution_ops.fused_embedding_to_layout(
      embs, splits, fid_offset, feature_offset, inp["nfl_offset"], inp["batch"], "example_batch",
      cfgs, inp["ps_num"])
  bufs = lc.oracle_bufs(plan, inp)
  exp = cpu.embedding_to_layout(plan.slices, plan.out_shapes, bufs, inp["fid_offset"],
                                inp["feature_offset"], inp["nfl_offset"], inp["batch"])
  assert len(outs) == len(exp)
  for i, (o, e) in enumerate(zip(outs, exp)):
    assert tuple(o.shape) == e.shape
    assert np.array_equal(o.detach().cpu().numpy().view(np.uint32), e.view(np.uint32)), i

  rng = np.random.Generator(np.random.PCG64(99))
  gouts = [rng.normal(0, 1, e.shape).astype(np.float32) for e in exp]
  torch.autograd.backward(outs, [torch.tensor(g, device=dev()) for g in gouts])
  egrads = cpu.embedding_to_layout_grad(plan.slices, plan.slice_idx_of, plan.max_slice_num,
                                        [(len(a), d) for a, d in bufs], inp["fid_offset"],
                                        inp["feature_offset"], inp["nfl_offset"], inp["batch"],
                                        gouts)
  # the oracle's per-feature buffers are consecutive cuts of embeddings_list
  cut = 0
  for t, e in enumerate(embs):
    got = e.grad.detach().cpu().numpy().reshape(-1)
    pos = 0
    while pos < got.size or (cut < len(bufs) and len(bufs[cut][0]) == 0 and pos == got.size):
      n = len(bufs[cut][0])
      d = bufs[cut][1]
      if n:  # (pooling_chain: every row has ~15 000 addends; the other cases stay below 4096)
        uses = np.full(n // d, 15000 if name == "pooling_chain" else 0)
        assert_gradient_rows(got[pos:pos + n].reshape(-1, d), egrads[cut].reshape(-1, d), uses,
                             (t, cut))
      pos += n
      cut += 1
      if cut >= len(bufs):
        break
  assert cut == len(bufs)


def test_layout_v1_and_validation():
  """Version 1 (embeddings_list = one [rows, dim] tensor per buffer) and argument checks."""
  from monolith_amd import _lib, distribution_ops, feature_configs as fc
  cfgs = fc.FeatureConfigs()
  cfgs.feature_configs["a"] = fc.FeatureConfig("t", [2, 2], fc.PoolingType.SUM, 0)
  oc = fc.OutConfig([fc.SliceConfig("a", 0, 2), fc.SliceConfig("a", 2, 4)])
  fc.infer_shape(oc, fc.OutType.CONCAT)
  cfgs.out_configs["o"] = oc
  emb = torch.tensor([[1., 2., 3., 4.], [10., 20., 30., 40.]], device=dev())
  fid_offset = torch.tensor([0, 1, 1], dtype=torch.int64, device=dev())
  feature_offset = torch.tensor([0, 2], dtype=torch.int32, device=dev())
  (out,) = distribution_ops.fused_embedding_to_layout([emb], [], fid_offset, feature_offset, [0], 2,
                                                      "instance", cfgs, 1, version=1)
  assert out.cpu().tolist() == [[11., 22., 33., 44.], [10., 20., 30., 40.]]
  with pytest.raises(_lib.MonolithHipError):  # two runs for a batch of three
    distribution_ops.fused_embedding_to_layout([emb], [], fid_offset, feature_offset, [0], 3,
                                               "instance", cfgs, 1, version=1)


@pytest.mark.parametrize("dim,opt,batch,ps_num,max_fids,vocab", [
    (16, "sgd", 257, 4, 80, 300), (16, "adagrad", 512, 2, 30, 5), (6, "adagrad", 33, 3, 12, 40),
    (64, "adagrad", 64, 2, 20, 200)])
def test_fused_lookup_to_layout_equals_lookup_then_layout(dim, opt, batch, ps_num, max_fids, vocab):
  """mono_lookup_to_layout (rows gathered from the tables and pooled in one pass) against the
  two-step form it fuses: table.lookup per (table, ps) FID list, then fused_embedding_to_layout
  — bit for bit, including FIDs the tables do not hold (rows of zeros) and, through the two-step
  form, against the oracle."""
  from monolith_amd import distribution_ops, feature_configs as fc
  from tests import hip_backend
  cfgs = lc.make_pooling_configs(dim=dim)
  inp = lc.make_inputs(cfgs, batch=batch, ps_num=ps_num, seed=dim + batch, max_fids=max_fids,
                       vocab=vocab, shared_every=3)
  plan = fc.build_layout_plan(cfgs)
  rng = np.random.Generator(np.random.PCG64(7))
  tables, fid_lists, embs_np = [], [], []
  for ti, t in enumerate(inp["tables"]):
    for ps in range(ps_num):
      fids = [f for n in inp["table_feats"][t] for f in inp["rows"][(n, ps)]]
      fl = np.array(fids, np.int64)
      rows = inp["embeddings_list"][ti * ps_num + ps].reshape(-1, dim).copy()
      assert len(rows) == len(fl)
      tab = hip_backend.make_table({"segments": [{"dim": dim, "opt": opt}]})
      keep = rng.random(len(fl)) > 0.1          # a tenth of the FIDs never reach the table
      if keep.any():
        tab.assign(fl[keep], rows[keep])
      rows[~keep] = 0.0
      tables.append(tab.t)
      fid_lists.append(torch.tensor(fl, device=dev()))
      embs_np.append(rows)
  splits = [torch.tensor(s) for s in inp["fid_list_row_split"]]
  fid_offset = torch.tensor(inp["fid_offset"].view(np.int64), device=dev())
  feature_offset = torch.tensor(inp["feature_offset"], device=dev())
  looked_up = [t.lookup(f) for t, f in zip(tables, fid_lists)]
  for e, r in zip(looked_up, embs_np):
    assert np.array_equal(e.cpu().numpy().reshape(-1, dim).view(np.uint32), r.view(np.uint32))
  two_step = distribution_ops.fused_embedding_to_layout(
      [e.reshape(-1) for e in looked_up], splits, fid_offset, feature_offset, inp["nfl_offset"],
      inp["batch"], "example_batch", cfgs, ps_num)
  fused = distribution_ops.fused_lookup_to_layout(
      tables, fid_lists, splits, fid_offset, feature_offset, inp["nfl_offset"], inp["batch"],
      "example_batch", cfgs, ps_num)
  bufs = lc.oracle_bufs(plan, dict(inp, embeddings_list=[r.reshape(-1) for r in embs_np]))
  exp = cpu.embedding_to_layout(plan.slices, plan.out_shapes, bufs, inp["fid_offset"],
                                inp["feature_offset"], inp["nfl_offset"], inp["batch"])
  assert len(fused) == len(two_step) == len(exp)
  for i, (a, b, e) in enumerate(zip(fused, two_step, exp)):
    assert np.array_equal(a.cpu().numpy().view(np.uint32), b.detach().cpu().numpy().view(np.uint32)), i
    assert np.array_equal(a.cpu().numpy().view(np.uint32), e.view(np.uint32)), i


def test_layout_gradient_chunked_switch_subprocess():
  """MONO_CHUNK=4096 (the optional switch; read once per process): destinations with more than
  4096 addends are summed in ordered chunks — the long-chain layout cases and config 3's pipeline
  at its own bag lengths stay within 2e-5 of the row's magnitude of the oracle, every shorter row
  bit-exact. (The default, tested by every other test of this file, is the reference's order.)"""
  import os
  import subprocess
  import sys
  env = dict(os.environ, MONO_CHUNK="4096")
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu",
                      os.path.join(root, "tests", "test_layout_gpu.py"),
                      os.path.join(root, "tests", "test_configs_gpu.py"), "-k",
                      "pooling_chain or pooling_hot or b2048_bag50", "--timeout", "120",
                      "--timeout-method", "thread"], env=env, cwd=root, capture_output=True,
                     text=True, timeout=600)
  assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
  assert "3 passed" in r.stdout, r.stdout[-500:]


def test_layout_gradient_pointer_form_subprocess():
  """The gradient's fallback for (slice, sample, seq) descriptors that do not fit 32 bits — per-use
  source pointers — is the same code path MONO_LAYOUT_GRAD_PTRS=1 forces; the switch is read once
  per process, so the bit-exact layout cases run again in a child process with it set."""
  import os
  import subprocess
  import sys
  env = dict(os.environ, MONO_LAYOUT_GRAD_PTRS="1")
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu",
                      os.path.join(root, "tests", "test_layout_gpu.py"), "-k",
                      "test_layout_forward_and_gradient_bit_exact", "--timeout", "120",
                      "--timeout-method", "thread"], env=env, cwd=root, capture_output=True,
                     text=True, timeout=600)
  assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
