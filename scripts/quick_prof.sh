#!/bin/bash
# quick look: plain bench line + kernel timeline of one step (rocprofv3 kernel trace)
#   scripts/quick_prof.sh <tag> [anchor kernel substring]
set -e -o pipefail
T=$1
A=${2:-k_probe}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
timeout -k 10 200 python3 bench.py --steps 50 --warmup 10 --no-cpu > $O/${T}_bench_plain.json 2> $O/${T}_bench_plain.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $O/prof_$T -o p -- python3 $R/bench.py --steps 20 --no-cpu \
  > $O/${T}_bench.json 2> $O/${T}_bench.err
cd $R
DB=$(find $O/prof_$T -name '*.db' | head -1)
python3 scripts/db_to_stats.py $DB $O/${T}_kernel_stats_steps.csv $A
python3 scripts/timeline.py $DB $A > $O/${T}_step_timeline.txt
rm -rf $O/prof_$T
cat $O/${T}_step_timeline.txt
python3 -c "
import json,sys
d=json.loads(open('$O/${T}_bench_plain.json').read().strip().splitlines()[-1])
print('%.2f G  %.1f us/step  step_frac %.3f'%(d['value']/1e9,d['ms_per_step']*1e3,d['roofline']['step_frac']), d['roofline']['kernel_ms'])
"
