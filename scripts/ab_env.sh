#!/bin/bash
# A/B of environment switches on the GPU box: scripts/ab_env.sh <tag> "<bench args>" VAR=val ... ("-" = no variable)
# one bench.py run per variant, results under gpurun_out/<tag>_<i>.json, summary on stdout
set -o pipefail
T=$1; shift
ARGS=$1; shift
O=gpurun_out; mkdir -p $O
i=0
for v in "$@"; do
  i=$((i+1))
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  echo "== ${e:-default}"
  env $e timeout -k 10 300 python3 bench.py $ARGS --no-cpu > $O/${T}_$i.json 2> $O/${T}_$i.err || { echo "FAILED rc=$?"; tail -5 $O/${T}_$i.err; exit 1; }
  python3 scripts/show_bench.py $O/${T}_$i.json
done
