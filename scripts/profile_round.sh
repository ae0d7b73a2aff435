#!/bin/bash
# One round's measurement set on the GPU box (run through gpurun from the repo root):
#   scripts/profile_round.sh r02_r [anchor kernel of a step, default k_probe]
# bench line, rocprofv3 --kernel-trace --stats of the same command, FETCH_SIZE / WRITE_SIZE passes
# (each its own run), the sharded pipeline at N=1. Summaries land in gpurun_out/<tag>_*; the
# rocprof databases are deleted on the box (gpurun_out is capped at 64 MiB).
set -e -o pipefail
T=$1
A=${2:-k_probe}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $O/prof_$T -o p -- python3 $R/bench.py --steps 20 \
  > $O/${T}_bench.json 2> $O/${T}_bench.err
cd $R
DB=$(find $O/prof_$T -name '*.db' | head -1)
python3 scripts/db_to_stats.py $DB $O/${T}_kernel_stats_steps.csv $A
python3 scripts/db_to_stats.py $DB $O/${T}_kernel_stats_all.csv
python3 scripts/timeline.py $DB $A ${TIMELINE_BACK:-0} > $O/${T}_step_timeline.txt
rm -rf $O/prof_$T
if [ -z "$SKIP_PMC" ]; then
cd /tmp
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python3 $R/bench.py \
  --steps 3 --warmup 1 --no-cpu > $O/pmc_f.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python3 $R/bench.py \
  --steps 3 --warmup 1 --no-cpu > $O/pmc_w.log 2>&1
cd $R
cp $(find $O/pmc_f -name '*counter_collection.csv' | head -1) $O/${T}_pmc_fetch_size.csv
cp $(find $O/pmc_w -name '*counter_collection.csv' | head -1) $O/${T}_pmc_write_size.csv
python3 scripts/pmc_summary.py $O/${T}_pmc_summary.json FETCH_SIZE=$O/${T}_pmc_fetch_size.csv \
  WRITE_SIZE=$O/${T}_pmc_write_size.csv
rm -rf $O/pmc_f $O/pmc_w
fi
timeout -k 10 200 python3 bench.py --steps 50 --warmup 10 > $O/${T}_bench_plain.json 2> $O/${T}_bench_plain.err
tail -c 600 $O/${T}_bench_plain.json
