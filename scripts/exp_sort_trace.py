"""Phase stamps (wall_clock64, 100 MHz) of the last fused sort pass, from the instrumented build
scripts/bin/libmonolith_hip_trace.so (built from a scratch copy of sort.hip; not part of the product)."""
import ctypes as C, sys
import numpy as np
import torch
lib = C.CDLL(sys.argv[1] if len(sys.argv) > 1 else "scripts/bin/libmonolith_hip_trace.so")
n, bits = 1703936, 28
rng = np.random.Generator(np.random.PCG64(1))
k = rng.integers(0, 1 << bits, n, dtype=np.uint64).astype(np.uint32)
kd = torch.from_numpy(k.view(np.int32)).cuda()
ko = torch.empty(n, dtype=torch.int32, device="cuda"); vo = torch.empty_like(ko)
ms = C.c_float(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
  rc = lib.mono_debug_sort_pairs(C.c_void_p(kd.data_ptr()), None, C.c_int64(n), bits, C.c_void_p(ko.data_ptr()),
                                 C.c_void_p(vo.data_ptr()), 1, C.byref(ms), st)
  assert rc == 0
tr = np.zeros(512 * 16, dtype=np.uint64)
assert lib.mono_debug_sort_trace(C.c_void_p(tr.ctypes.data)) == 0
tr = tr.reshape(512, 16)[:208].astype(np.int64)
t0 = tr[:, 0].min()
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 9
names = ["s%d" % i for i in range(NS)]
rel = (tr[:, :NS] - t0) / 100.0   # us
order = np.argsort(tr[:, 15])
rel = rel[order]
print("us since first start; mean / min / max over 208 tiles")
for i, nm in enumerate(names):
  print("%-10s %6.2f %6.2f %6.2f" % (nm, rel[:, i].mean(), rel[:, i].min(), rel[:, i].max()))
print("per-phase mean duration:", np.round(np.diff(rel, axis=1).mean(axis=0), 2))
for t in (0, 15, 16, 100, 207):
  print("tile", t, np.round(rel[t], 2))
print("sort ms", ms.value)
