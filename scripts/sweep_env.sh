#!/bin/bash
# usage: sweep_env.sh VAR v1 v2 ... : bench.py (no CPU leg) once per value of an experiment switch
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout -k 10 120 python3 bench.py --steps 100 --warmup 10 --no-cpu > gpurun_out/sweep.json || exit 1
  python3 -c "
import json; d=json.load(open('gpurun_out/sweep.json')); print('$VAR=$v', round(d['value']/1e9,3), round(d['ms_per_step']*1e3,1), {k: round(x*1e3,1) for k,x in d['roofline']['kernel_ms'].items()})"
done
