"""Ceiling probe: random row gather / scatter bandwidth of the GPU with library kernels
(torch.index_select / index_copy_), for row widths 64 B and 128 B. Not part of the product."""
import torch, time
dev = torch.device("cuda", 0)
for width in (16, 32, 64):
  rows = (1 << 33) // (width * 4)   # 8 GiB table
  t = torch.empty((rows, width), dtype=torch.float32, device=dev).normal_()
  for n in (1 << 20, 1 << 22, 1 << 24):
    idx = torch.randint(0, rows, (n,), device=dev)
    out = torch.empty((n, width), dtype=torch.float32, device=dev)
    for name, fn in (("gather", lambda: torch.index_select(t, 0, idx, out=out)),
                     ("scatter", lambda: t.index_copy_(0, idx, out))):
      fn(); torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(5): fn()
      e1.record(); torch.cuda.synchronize()
      ms = e0.elapsed_time(e1) / 5
      gb = n * width * 4 * 2 / 1e9   # random side + streaming side
      print("width %3d B n %9d %-7s %.3f ms  %.0f GB/s (random side %.0f GB/s)" %
            (width * 4, n, name, ms, gb / ms * 1e3, gb / 2 / ms * 1e3))
  del t
