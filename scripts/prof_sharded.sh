#!/bin/bash
# timeline of one step of the sharded pipeline at N = 1:  scripts/prof_sharded.sh <tag> <bench args...>
set -e -o pipefail
T=$1; shift
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout -k 10 280 python3 bench.py "$@" --steps 30 --warmup 10 --no-cpu > $O/${T}_bench.json 2> $O/${T}_bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $O/prof_$T -o p -- python3 $R/bench.py "$@" --steps 10 --warmup 5 --no-cpu \
  > $O/${T}_prof_bench.json 2> $O/${T}_prof_bench.err
cd $R
DB=$(find $O/prof_$T -name '*.db' | head -1)
python3 scripts/db_to_stats.py $DB $O/${T}_kernel_stats.csv k_reorder_keys
python3 scripts/timeline.py $DB k_reorder_keys > $O/${T}_timeline.txt
rm -rf $O/prof_$T
cat $O/${T}_timeline.txt
python3 -c "
import json
d=json.loads(open('$O/${T}_bench.json').read().strip().splitlines()[-1])
print('%.2f G  %.1f us/step'%(d['value']/1e9,d['ms_per_step']*1e3))
"
