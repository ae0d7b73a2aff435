"""CPU: the layout oracle (oracle/layout_oracle.cc) and the config flattening
(monolith_amd.feature_configs.build_layout_plan) pinned against the reference test's own truth
algorithm (fused_embedding_to_layout_test.py:90-115,435-481) and its gradient assertion
("grad == number of uses" under all-ones layout gradients, :682-694, incl. the MEAN quirk)."""
import numpy as np
import pytest

from monolith_amd import feature_configs as fc
from oracle import cpu
from tests import layout_cases as lc


def _use_counts(cfgs, plan, inp):
  """Expected gradient under all-ones layout gradients: every (fid, slice columns) gets +1 per
  use (layout slice x sample x occurrence; FIRSTN only for occurrences below max_seq)."""
  fcs = cfgs.feature_configs
  exp = {}
  for lname in sorted(cfgs.out_configs):
    for sc in cfgs.out_configs[lname].slice_configs:
      f = fcs[sc.feature_name]
      shared, runs = inp["sample_fids"][sc.feature_name]
      for b in range(inp["batch"]):
        fids = runs[0] if shared else runs[b]
        for q, fid in enumerate(fids):
          if f.pooling_type == fc.PoolingType.FIRSTN and q >= f.max_sequence_length:
            continue
          g = exp.setdefault((sc.feature_name, fid), np.zeros(sum(f.slice_dims), np.float32))
          g[sc.start:sc.end] += 1.0
  return exp


CASES = [("ref_shaped", lc.make_configs, dict(batch=10, ps_num=5, seed=2)),
         ("ref_shaped_b1", lc.make_configs, dict(batch=1, ps_num=2, seed=3)),
         ("no_shared", lc.make_configs, dict(batch=7, ps_num=3, seed=4, shared_every=0)),
         ("pooling", lc.make_pooling_configs, dict(batch=33, ps_num=4, seed=5, max_fids=40,
                                                   vocab=60, shared_every=0))]


@pytest.mark.parametrize("name,mk,kw", CASES, ids=[c[0] for c in CASES])
def test_layout_oracle_matches_reference_truth(name, mk, kw):
  cfgs = mk()
  inp = lc.make_inputs(cfgs, **kw)
  plan = fc.build_layout_plan(cfgs)
  outs = cpu.embedding_to_layout(plan.slices, plan.out_shapes, lc.oracle_bufs(plan, inp),
                                 inp["fid_offset"], inp["feature_offset"], inp["nfl_offset"],
                                 inp["batch"])
  exp = lc.truth(cfgs, inp)
  assert len(outs) == len(exp)
  for o, e in zip(outs, exp):
    assert o.shape == e.shape
    np.testing.assert_allclose(o, e, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name,mk,kw", CASES, ids=[c[0] for c in CASES])
def test_layout_grad_oracle_is_use_count(name, mk, kw):
  cfgs = mk()
  inp = lc.make_inputs(cfgs, **kw)
  plan = fc.build_layout_plan(cfgs)
  bufs = lc.oracle_bufs(plan, inp)
  ones = [np.ones([inp["batch"] if d == -1 else d for d in s], np.float32) for s in plan.out_shapes]
  grads = cpu.embedding_to_layout_grad(plan.slices, plan.slice_idx_of, plan.max_slice_num,
                                       [(len(a), d) for a, d in bufs], inp["fid_offset"],
                                       inp["feature_offset"], inp["nfl_offset"], inp["batch"], ones)
  exp = _use_counts(cfgs, plan, inp)
  # walk fid_offset to find every fid's (buffer, row)
  seen = {}
  pos = 0
  for name_ in inp["names"]:
    shared, runs = inp["sample_fids"][name_]
    for fids in runs:
      for fid in fids:
        fo = int(inp["fid_offset"][pos]); pos += 1
        seen[(name_, fid)] = (fo >> 32, fo & 0xFFFFFFFF)
  for key, (i1, i2) in seen.items():
    dim = bufs[i1][1]
    got = grads[i1][i2 * dim:(i2 + 1) * dim]
    assert np.array_equal(got, exp.get(key, np.zeros(dim, np.float32))), key


def test_plan_matches_reference_docstring_layout():
  """Layout order / offsets the reference's Layout classes produce (:92-242)."""
  cfgs = lc.make_configs()
  plan = fc.build_layout_plan(cfgs)
  # layouts sorted: bias, ffm1, ffm2 (NONE: one tensor per slice), firstN (NONE), vec
  n_ffm2 = len(cfgs.out_configs["ffm2"].slice_configs)
  n_first = len(cfgs.out_configs["firstN"].slice_configs)
  assert len(plan.out_shapes) == 1 + 1 + n_ffm2 + n_first + 1
  assert plan.out_shapes[0] == [-1, 1] and plan.out_shapes[1] == [-1, 4, 8]
  assert plan.out_shapes[-1] == [-1, 4 * 7]
  bias = [s for s in plan.slices if s.out_index == 0]
  assert all(s.addn and s.out_offset == 0 and s.out_stride == 1 for s in bias)
  assert [s.first for s in bias] == [1] + [0] * (len(bias) - 1)
  stack = [s for s in plan.slices if s.out_index == 1]
  assert [s.out_offset for s in stack] == [0, 8, 16, 24] and all(s.out_stride == 32 for s in stack)
  vec = [s for s in plan.slices if s.out_index == len(plan.out_shapes) - 1]
  assert [s.out_offset for s in vec] == [4 * i for i in range(7)]
  assert plan.feature_names[:3] == ["fc_slot_1", "fc_slot_10", "fc_slot_11"]
