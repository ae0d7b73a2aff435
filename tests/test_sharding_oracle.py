"""oracle.cpu.sharding_sparse_fids (restatement of ShardingSparseFidsV2) against the truth
algorithm of the reference's test (data/data_ops_v2_test.py:84-202). The reference test's input
file (training_instance/examplebatch.data) is not part of the checkout, so the inputs are seeded
synthetic ragged FIDs with the same structure: several tables, shared features, features absent
from the input, empty runs, FIDs with the top bit set."""
import numpy as np
import pytest

from oracle import cpu
from tests import sharding_cases as sc


@pytest.mark.parametrize("ps_num", [1, 3, 5])
@pytest.mark.parametrize("seed", [1, 2])
def test_oracle_matches_reference_test_truth(seed, ps_num):
  ft, feats = sc.make_case(seed)
  t = sc.truth(ft, feats, ps_num)
  for unique, tag in ((False, ""), (True, "_unique")):
    fid_list, row_splits, fid_offset, feature_offset, nfl_offset = cpu.sharding_sparse_fids(
        ft, feats, ps_num, unique)
    assert nfl_offset.tolist() == t["nfl_offset"]
    assert feature_offset.tolist() == t["feature_offset"]
    assert fid_offset.tolist() == t["fid_offset" + tag]
    assert len(fid_list) == len(t["keys"])
    for k, fl, rs in zip(t["keys"], fid_list, row_splits):
      assert fl.tolist() == t["fid_map" + tag][k]
      assert rs.tolist() == t["row_split" + tag][k]


def test_hand_case():
  """Two tables, ps_num 2: the output index enumerates (table, ps, feature in table)."""
  ft = {"a": "t0", "b": "t1", "c": "t0"}
  feats = {"a": ("rows", [[4, 5, 4], []]), "b": ("shared", [7, 8]), "c": ("rows", [[6], [6, 9]])}
  fid_list, row_splits, fid_offset, feature_offset, nfl_offset = cpu.sharding_sparse_fids(
      ft, feats, 2, True)
  # t0: features a, c -> out index = ps * 2 + {a: 0, c: 1}; t1 starts at 4
  assert [x.tolist() for x in fid_list] == [[4, 6], [5, 9], [8], [7]]
  assert [x.tolist() for x in row_splits] == [[0, 1, 2], [0, 1, 2], [0, 1], [0, 1]]
  assert nfl_offset.tolist() == [0, 2 | (1 << 31), 3]
  assert feature_offset.tolist() == [0, 3, 3, 5, 6]
  assert fid_offset.tolist() == [(0 << 32) | 0, (2 << 32) | 0, (0 << 32) | 0,       # a: 4 5 4
                                 (5 << 32) | 0, (4 << 32) | 0,                      # b: 7 8
                                 (1 << 32) | 0, (1 << 32) | 0, (3 << 32) | 0]       # c: 6 | 6 9
