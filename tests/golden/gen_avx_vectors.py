"""Generates tests/golden/avx_vectors.json from the reference's own avx_utils.h, compiled by
oracle/Makefile into oracle/_ref/libavx_ref.so (build container only: needs /root/reference).

  python tests/golden/gen_avx_vectors.py

Inputs are seeded; the outputs are what BaselineAdagradOptimize / the AVX AdagradOptimize /
ReduceSum of the reference produce on this machine. Floats are stored as uint32 bit patterns
so the fixture is exact.
"""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# production flags (-D_ENABLE_AVX -mavx -mfma, optimizer/BUILD:94-109): AVX body + FMA-contracted tail
ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libavx_ref.so"))
# same header without AVX/FMA flags and -ffp-contract=off: the plain "baseline" arithmetic
base = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libbase_ref.so"))


def p(a):
  return C.c_void_p(a.ctypes.data)


def bits(a):
  return np.asarray(a, np.float32).view(np.uint32).tolist()


rng = np.random.Generator(np.random.PCG64(20250117))
cases = []
for dim in (1, 2, 7, 8, 16, 64):
  for wd in (0.0, 0.1):
    num = rng.uniform(-0.05, 0.05, dim).astype(np.float32)
    norm = np.full(dim, 0.1, np.float32)
    num_b, norm_b = num.copy(), norm.copy()
    num_a, norm_a = num.copy(), norm.copy()
    steps = []
    for step in range(3):
      g = rng.normal(0, 0.1, dim).astype(np.float32)
      base.ref_baseline_adagrad(p(num_b), p(norm_b), p(g), C.c_size_t(dim), C.c_float(0.01),
                               C.c_float(wd))
      ref.ref_adagrad(p(num_a), p(norm_a), p(g), C.c_size_t(dim), C.c_float(0.01), C.c_float(wd))
      steps.append({"grad": bits(g), "baseline_num": bits(num_b), "baseline_norm": bits(norm_b),
                    "avx_num": bits(num_a), "avx_norm": bits(norm_a)})
    cases.append({"dim": dim, "lr": 0.01, "wd": wd, "num0": bits(num), "norm0": bits(norm),
                  "steps": steps})

# the two known answers of adagrad_optimizer_test.cc:32-73 through the reference header
num = np.zeros(2, np.float32); norm = np.ones(2, np.float32); g = np.array([1, 2], np.float32)
base.ref_baseline_adagrad(p(num), p(norm), p(g), C.c_size_t(2), C.c_float(0.1), C.c_float(0.1))
first = num.tolist()
base.ref_baseline_adagrad(p(num), p(norm), p(g), C.c_size_t(2), C.c_float(0.1), C.c_float(0.1))
known = {"first": first, "second": num.tolist()}

rs = []
for dim in (1, 8, 10, 16, 33):
  a = rng.normal(0, 1, dim).astype(np.float32)
  b = rng.normal(0, 1, dim).astype(np.float32)
  out = np.empty(dim, np.float32)
  ref.ref_reduce_sum(p(a), p(b), p(out), C.c_size_t(dim))
  rs.append({"a": bits(a), "b": bits(b), "out": bits(out)})

with open(os.path.join(HERE, "avx_vectors.json"), "w") as f:
  json.dump({"source": "monolith/native_training/runtime/hash_table/optimizer/avx_utils.h "
                       "(baseline_*: g++ -O2 -ffp-contract=off; avx_*: g++ -O2 -D_ENABLE_AVX -mavx -mfma)",
             "adagrad": cases, "adagrad_test_known": known, "reduce_sum": rs}, f)
print("wrote avx_vectors.json", known)
