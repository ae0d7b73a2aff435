"""Seeded ragged inputs for sharding_sparse_fids + the truth algorithm of the reference's own
test (data/data_ops_v2_test.py:84-202: get_feature_cfg / handle_feature / get_offset_result),
restated over plain python lists."""
from collections import defaultdict

import numpy as np


def make_case(seed, n_features=7, n_tables=3, batch=9, max_len=6, vocab=25, shared=(2,), absent=(4,),
              negative=True):
  rng = np.random.Generator(np.random.PCG64(seed))
  names = ["f_%s" % "abcdefghijklmnopqrstuvwxyz"[(i * 7) % 26] + str(i) for i in range(n_features)]
  feature_tables = {n: "table_%d" % (i % n_tables) for i, n in enumerate(names)}
  features = {}
  for i, n in enumerate(names):
    if i in absent:
      continue

    def draw(k):
      v = ((i + 1) << 48) | rng.integers(0, vocab, k).astype(np.int64)
      if negative and i % 3 == 0:
        v = v | np.int64(-2**63)  # top bit set: shard must come from the UNSIGNED value
      return [int(x) for x in v]

    if i in shared:
      features[n] = ("shared", draw(int(rng.integers(1, max_len + 1))))
    else:
      features[n] = ("rows", [draw(int(rng.integers(0, max_len + 1))) for _ in range(batch)])
  return feature_tables, features


def truth(feature_tables, features, ps_num):
  """Returns dict with the non-unique and unique expectations."""
  feature_cfg, table_cfg = {}, {}
  for name, table in feature_tables.items():
    feature_cfg[name] = {"feature_name": name, "table_name": table}
    table_cfg.setdefault(table, {"table_name": table, "feature_list": []})
  table_name_sort = sorted(table_cfg)
  feature_name_sort = sorted(feature_cfg)
  for name in feature_name_sort:
    f, t = feature_cfg[name], table_cfg[feature_cfg[name]["table_name"]]
    f["feature_in_table_index"] = len(t["feature_list"])
    t["feature_list"].append(name)
  pre_index = 0
  for name in table_name_sort:
    t = table_cfg[name]
    for fn in t["feature_list"]:
      feature_cfg[fn]["pre_output_index"] = pre_index
      feature_cfg[fn]["table_feature_count"] = len(t["feature_list"])
    pre_index += max(len(t["feature_list"]), 1) * ps_num

  fid_map, fid_map_unique, unique_map = defaultdict(list), defaultdict(list), defaultdict(dict)
  off_map, off_map_unique = defaultdict(list), defaultdict(list)
  for name in feature_name_sort:
    if name not in features:
      continue
    f = feature_cfg[name]
    kind, data = features[name]
    for run in ([data] if kind == "shared" else data):
      o1, o2 = [], []
      for value in run:
        shard = (value & 0xFFFFFFFFFFFFFFFF) % ps_num
        key = name + ":" + str(shard)
        base = (f["pre_output_index"] + shard * f["table_feature_count"] +
                f["feature_in_table_index"]) << 32
        o1.append(base | len(fid_map[key]))
        fid_map[key].append(value)
        if value not in unique_map[key]:
          unique_map[key][value] = len(unique_map[key])
          fid_map_unique[key].append(value)
        o2.append(base | unique_map[key][value])
      off_map[name].append(o1)
      off_map_unique[name].append(o2)
  feature_offset, nfl_offset, fid_offset, fid_offset_unique = [], [], [], []
  for name in feature_name_sort:
    shared = name in features and features[name][0] == "shared"
    nfl_offset.append(len(feature_offset) | (int(shared) << 31))
    for o1, o2 in zip(off_map.get(name, []), off_map_unique.get(name, [])):
      feature_offset.append(len(fid_offset))
      fid_offset.extend(o1)
      fid_offset_unique.extend(o2)
  out = {"nfl_offset": nfl_offset, "feature_offset": feature_offset, "fid_offset": fid_offset,
         "fid_offset_unique": fid_offset_unique}
  for tag, src in (("", fid_map), ("_unique", fid_map_unique)):
    lists, splits = {}, {}
    for t in table_name_sort:
      for ps in range(ps_num):
        k = t + ":" + str(ps)
        lists[k], splits[k] = [], [0]
        for fn in table_cfg[t]["feature_list"]:
          lists[k].extend(src[fn + ":" + str(ps)])
          splits[k].append(len(lists[k]))
    out["fid_map" + tag], out["row_split" + tag] = lists, splits
  out["keys"] = [t + ":" + str(ps) for t in table_name_sort for ps in range(ps_num)]
  return out
