#!/bin/bash
# A/B of the summation-order switch on the GPU box: scripts/exact_order_ab.sh <tag>
# config 2 (default line) and config 3 with chunked hot keys (default) and MONO_EXACT_ORDER=1.
set -e -o pipefail
T=$1
O=gpurun_out; mkdir -p $O
timeout -k 10 200 python3 bench.py --steps 50 --warmup 10 --no-cpu > $O/${T}_c2_chunk.json 2> $O/${T}_c2_chunk.err
MONO_EXACT_ORDER=1 timeout -k 10 200 python3 bench.py --steps 50 --warmup 10 --no-cpu > $O/${T}_c2_exact.json 2> $O/${T}_c2_exact.err
timeout -k 10 200 python3 bench.py --config 3 --steps 20 --warmup 5 --no-cpu > $O/${T}_c3_chunk.json 2> $O/${T}_c3_chunk.err
MONO_EXACT_ORDER=1 timeout -k 10 200 python3 bench.py --config 3 --steps 20 --warmup 5 --no-cpu > $O/${T}_c3_exact.json 2> $O/${T}_c3_exact.err
python3 - <<P
import json
for f in ("c2_chunk", "c2_exact", "c3_chunk", "c3_exact"):
  d = json.loads(open("$O/${T}_%s.json" % f).read().strip().splitlines()[-1])
  r = d["roofline"]
  print(f, "%.2f G" % (d["value"] / 1e9), "%.1f us" % (d["ms_per_step"] * 1e3), r.get("kernel_ms") or r.get("phase_ms"))
P
