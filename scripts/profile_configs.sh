#!/bin/bash
# Bench line + rocprofv3 kernel stats of configs 3, 4 and 5 (BASELINE.json), and the sharded
# pipeline on the config-2 shape at N=1:  scripts/profile_configs.sh r02_r
set -e -o pipefail
T=$1
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
for c in 3 4 5; do
  timeout -k 10 280 python3 bench.py --config $c > $O/${T}_c${c}_bench.json 2> $O/${T}_c${c}_bench.err
  scripts/prof_config.sh ${T}_c${c}_prof --config $c --steps 10 --warmup 3 > $O/${T}_c${c}_top.txt
done
scripts/prof_sharded.sh ${T}_sharded_n1 --sharded > $O/${T}_sharded_n1_top.txt
scripts/prof_sharded.sh ${T}_c4_n1 --config 4 > $O/${T}_c4_n1_top.txt
