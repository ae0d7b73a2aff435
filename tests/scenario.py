"""Data-driven replay of the reference's unit-test vectors (tests/golden/table_scenarios.json).

One runner, two back-ends: the CPU oracle (not-gpu tests) and the HIP path through the C-ABI
(gpu tests). A back-end is a module-like object with make_table(cfg) / make_filter(cfg); tables
expose lookup / assign / assign_add / apply_gradients / size / clear / evict returning numpy.
"""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_scenarios(name="table_scenarios.json"):
  with open(os.path.join(GOLDEN, name)) as f:
    return json.load(f)


def run_scenario(backend, sc):
  filt = backend.make_filter(sc["filter"]) if sc.get("filter") else None
  table = backend.make_table(sc["table"])
  dim = sum(s["dim"] for s in sc["table"]["segments"])
  for i, st in enumerate(sc["steps"]):
    op = st["op"]
    where = "%s step %d (%s) [%s]" % (sc["name"], i, op, sc["cite"])
    if op in ("assign", "assign_add"):
      ids = np.array(st["ids"], np.int64)
      vals = np.array(st["values"], np.float32).reshape(len(ids), dim)
      getattr(table, op)(ids, vals, req_time=st.get("req_time", 0), filt=filt)
    elif op == "apply_gradients":
      ids = np.array(st["ids"], np.int64)
      if "grads_fill" in st:
        g = np.full((len(ids), dim), st["grads_fill"], np.float32)
      else:
        g = np.array(st["grads"], np.float32).reshape(len(ids), dim)
      table.apply_gradients(ids, g, np.array(st["lrs"], np.float32),
                            req_time=st.get("req_time", 0), global_step=0,
                            enable_dedup=st.get("dedup", False), filt=filt)
    elif op == "lookup":
      ids = np.array(st["ids"], np.int64)
      got = np.asarray(table.lookup(ids))
      if "expect_fill" in st:
        exp = np.repeat(np.array(st["expect_fill"], np.float32)[:, None], dim, axis=1)
      else:
        exp = np.array(st["expect"], np.float32).reshape(len(ids), dim)
      tol = st.get("tol", 0.0)
      if tol == 0.0:
        assert np.array_equal(got, exp), "%s: %r != %r" % (where, got, exp)
      else:
        assert np.allclose(got, exp, rtol=tol, atol=tol), "%s: %r != %r" % (where, got, exp)
    elif op == "size":
      assert table.size() == st["expect"], where
    elif op == "clear":
      table.clear()
    elif op == "evict":
      table.evict(st["time"])
    else:
      raise ValueError(op)
  return table
