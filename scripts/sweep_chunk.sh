#!/bin/bash
# MONO_CHUNK sweep of the config-2 step (hot-key chunk size); one JSON line per setting
for c in 512 1024 2048 4096; do
  echo "MONO_CHUNK=$c"
  MONO_CHUNK=$c timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu || exit 1
done
