#!/bin/bash
# kernel timeline of a TIMED step (not bench.py's trailing phase-timing loop) under rocprofv3:
#   scripts/prof_tl.sh <tag> <anchor kernel substring> <steps back from the end> [VAR=val ...]
set -e -o pipefail
T=$1; A=$2; B=$3; shift 3
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
for v in "$@"; do export "$v"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $O/prof_$T -o p -- python3 $R/bench.py --steps 20 --no-cpu \
  > $O/${T}_bench.json 2> $O/${T}_bench.err
cd $R
DB=$(find $O/prof_$T -name '*.db' | head -1)
python3 scripts/timeline.py $DB $A $B > $O/${T}_step_timeline.txt
python3 scripts/timeline.py $DB $A $((B+1)) >> $O/${T}_step_timeline.txt
rm -rf $O/prof_$T
cat $O/${T}_step_timeline.txt
