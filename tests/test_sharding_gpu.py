"""monolith_amd.parsers.sharding_sparse_fids (HIP, through the C-ABI) against the oracle,
bit-exact, plus the hand-off to the fused layout op and a config-3 sized property check."""
import numpy as np
import pytest
import torch

from monolith_amd import feature_configs as fc
from monolith_amd import parsers
from monolith_amd.embedding_combiners import Ragged
from oracle import cpu
from tests import sharding_cases as sc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfgs(feature_tables):
  cfgs = fc.FeatureConfigs()
  for n, t in feature_tables.items():
    cfgs.feature_configs[n] = fc.FeatureConfig(t, [4], fc.PoolingType.SUM, 0)
  return cfgs


def _to_device(features):
  out = {}
  for n, (kind, data) in features.items():
    if kind == "shared":
      out[n] = parsers.SharedFeature(torch.tensor(data, dtype=torch.int64, device=DEV))
    else:
      out[n] = Ragged.from_lists(data, DEV)
  return out


def _check(ft, feats, ps_num, unique):
  exp = cpu.sharding_sparse_fids(ft, feats, ps_num, unique)
  got = parsers.sharding_sparse_fids(_to_device(feats), ps_num, _cfgs(ft), unique,
                                     fid_list_ret_list=True)
  fid_list, fid_offset, feature_offset, nfl_offset, batch, row_splits = got
  assert len(fid_list) == len(exp[0])
  for a, e in zip(fid_list, exp[0]):
    assert np.array_equal(a.cpu().numpy(), e)
  for a, e in zip(row_splits, exp[1]):
    assert np.array_equal(a.numpy(), e)
  assert np.array_equal(fid_offset.cpu().numpy().view(np.uint64), exp[2])
  assert np.array_equal(feature_offset.cpu().numpy(), exp[3])
  assert np.array_equal(nfl_offset, exp[4])
  return got


@pytest.mark.parametrize("unique", [False, True])
@pytest.mark.parametrize("ps_num", [1, 3, 5])
def test_matches_oracle(ps_num, unique):
  for seed in (1, 2, 3):
    ft, feats = sc.make_case(seed)
    got = _check(ft, feats, ps_num, unique)
    assert got[4] == 9


def test_edge_cases():
  # every run empty; a single feature; a feature whose only input is shared; -1 as an id
  ft = {"a": "t", "b": "t"}
  _check(ft, {"a": ("rows", [[], [], []]), "b": ("rows", [[], [], []])}, 2, True)
  _check({"a": "t"}, {"a": ("rows", [[3, 3, 3], [3]])}, 1, True)
  _check({"a": "t"}, {"a": ("rows", [[3, 3, 3], [3]])}, 1, False)
  _check(ft, {"b": ("shared", [5, 6, 5])}, 3, True)
  _check(ft, {"a": ("rows", [[-1, -1, 2], [-1]]), "b": ("rows", [[-1], []])}, 3, True)
  # the name dict decides the order, not insertion order
  dict_out = parsers.sharding_sparse_fids(
      _to_device({"b": ("rows", [[1], [2]]), "a": ("rows", [[3], [4]])}), 2, _cfgs(ft), True)
  assert list(dict_out[0].keys()) == ["t:0", "t:1"]
  assert dict_out[0]["t:0"].cpu().tolist() == [4, 2] and dict_out[0]["t:1"].cpu().tolist() == [3, 1]
  with pytest.raises(ValueError):
    parsers.sharding_sparse_fids(_to_device({"zz": ("rows", [[1]])}), 2, _cfgs(ft), True)


def test_feeds_fused_layout():
  """sharding_sparse_fids -> per-buffer 'embeddings' -> fused_embedding_to_layout: with every
  embedding row = [fid-derived value] the pooled output equals the per-sample sum computed
  directly from the ragged input."""
  from monolith_amd import distribution_ops
  B, ps_num = 16, 3
  rng = np.random.Generator(np.random.PCG64(4))
  names = ["slot_a", "slot_b", "slot_c"]
  cfgs = fc.FeatureConfigs()
  oc = fc.OutConfig()
  for i, n in enumerate(names):
    cfgs.feature_configs[n] = fc.FeatureConfig("t%d" % (i % 2), [2], fc.PoolingType.SUM, 0)
    oc.slice_configs.append(fc.SliceConfig(n, 0, 2))
  fc.infer_shape(oc, fc.OutType.CONCAT)
  cfgs.out_configs = {"concat": oc}
  rows = {n: [[int(x) for x in ((i + 1) << 48) | rng.integers(0, 9, int(rng.integers(0, 5)))]
              for _ in range(B)] for i, n in enumerate(names)}
  feats = {n: Ragged.from_lists(r, DEV) for n, r in rows.items()}
  fid_list, fid_offset, feature_offset, nfl_offset, batch, row_splits = \
      parsers.sharding_sparse_fids(feats, ps_num, cfgs, True, fid_list_ret_list=True)
  assert batch == B

  def emb_of(fids):  # deterministic 2-float row per fid
    f = (fids & 0xFFFF).to(torch.float32)
    return torch.stack([f, f * 0.5 + 1.0], dim=1)

  embs = [emb_of(f).reshape(-1) for f in fid_list]
  outs = distribution_ops.fused_embedding_to_layout(
      embs, row_splits, fid_offset, feature_offset, nfl_offset, B, "example_batch", cfgs, ps_num)
  out = outs[0].cpu().numpy()
  for b in range(B):
    for i, n in enumerate(names):
      v = np.array([x & 0xFFFF for x in rows[n][b]], np.float32)
      exp = np.array([v.sum(), (v * 0.5 + 1.0).sum()], np.float32) if len(v) else np.zeros(2)
      assert np.allclose(out[b, 2 * i:2 * i + 2], exp, rtol=1e-6)


def test_config3_size_properties():
  """B = 8192, 8 multi-hot slots, Poisson(50) bag lengths, Zipf(1.1) ids (3.3 M FIDs): every
  occurrence's fid_offset points at its own FID; unique lists hold no duplicates; non-unique
  lists are a permutation of the input."""
  B, S, ps_num = 8192, 8, 4
  rng = np.random.Generator(np.random.PCG64(8))
  cfgs = fc.FeatureConfigs()
  feats, names = {}, ["slot_%02d" % s for s in range(S)]
  for s, n in enumerate(names):
    cfgs.feature_configs[n] = fc.FeatureConfig("t%d" % (s % 2), [16], fc.PoolingType.SUM, 0)
    lens = np.clip(rng.poisson(50, B), 1, 200)
    vals = ((s + 1) << 48) | (rng.zipf(1.1, int(lens.sum())).astype(np.int64) % 10_000_000)
    feats[n] = Ragged(torch.tensor(vals, device=DEV),
                      torch.tensor(np.concatenate([[0], np.cumsum(lens)]), device=DEV))
  plan = parsers.build_sharding_plan(cfgs, ps_num)
  all_in = torch.cat([feats[n].values for n in plan.feature_names])
  for unique in (True, False):
    fid_list, fid_offset, feature_offset, nfl_offset, batch, row_splits = \
        parsers.sharding_sparse_fids(feats, ps_num, cfgs, unique, fid_list_ret_list=True)
    assert batch == B and fid_offset.numel() == all_in.numel()
    # start of every out-index group inside the concatenation of fid_list
    sizes = torch.tensor([f.numel() for f in fid_list])
    buf_base = torch.cumsum(sizes, 0) - sizes
    starts = torch.zeros(plan.n_out, dtype=torch.int64)
    for ti, tf in enumerate(plan.table_features):
      for ps in range(ps_num):
        rs = row_splits[ti * ps_num + ps]
        for fi, n in enumerate(tf):
          starts[plan.out_index[plan.feature_names.index(n), ps]] = \
              buf_base[ti * ps_num + ps] + rs[fi]
    cat = torch.cat(fid_list)
    oi, idx = fid_offset >> 32, fid_offset & 0xFFFFFFFF
    assert torch.equal(cat[starts.to(DEV)[oi] + idx], all_in)
    shard = (all_in % ps_num)  # ids are non-negative here
    m_of = torch.repeat_interleave(torch.arange(S, device=DEV),
                                   torch.tensor([feats[n].values.numel() for n in plan.feature_names],
                                                device=DEV))
    assert torch.equal(oi, torch.tensor(plan.out_index, device=DEV, dtype=torch.int64)[m_of, shard])
    if unique:
      assert cat.numel() == torch.unique(cat).numel() == torch.unique(all_in).numel()
    else:
      assert torch.equal(torch.sort(cat).values, torch.sort(all_in).values)
    fo = feature_offset.cpu().numpy()
    assert np.all(np.diff(fo) >= 0) and fo[0] == 0 and len(fo) == S * B
