#!/bin/bash
# Kernel trace of one bench configuration: scripts/prof_cfg.sh <tag> <anchor kernel> <bench args...>
# -> gpurun_out/<tag>_kernel_stats_all.csv, <tag>_timeline.txt, <tag>_bench.json
set -e -o pipefail
T=$1; A=$2; shift 2
R=$(pwd); O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $O/prof_$T -o p -- python3 $R/bench.py "$@" \
  > $O/${T}_bench.json 2> $O/${T}_bench.err
cd $R
DB=$(find $O/prof_$T -name '*.db' | head -1)
python3 scripts/db_to_stats.py $DB $O/${T}_kernel_stats_all.csv
python3 scripts/timeline.py $DB $A ${TIMELINE_BACK:-0} > $O/${T}_timeline.txt
rm -rf $O/prof_$T
