#!/bin/bash
for cfg in "2 256" "2 128" "1 256" "1 512" "2 192"; do set -- $cfg
  echo "MONO_GATHER_ILP=$1 MONO_GATHER_BLOCKS=$2"
  MONO_GATHER_ILP=$1 MONO_GATHER_BLOCKS=$2 timeout -k 10 120 python bench.py --steps 50 --warmup 10 --no-cpu || exit 1
done
