#!/bin/bash
# MONO_CHUNK sweep of the config-2 step (hot-key chunk size); one JSON line per setting
for c in 1024 4096 16384 1048576; do
  echo "MONO_CHUNK=$c"
  MONO_CHUNK=$c timeout -k 10 120 python bench.py --steps 50 --warmup 10 --no-cpu || exit 1
done
