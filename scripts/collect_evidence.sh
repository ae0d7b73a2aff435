#!/bin/bash
# Copy one evidence set from gpurun_out/ (scratch) into profiles/ (tracked):
#   scripts/collect_evidence.sh r02_r      (after profile_round.sh and profile_configs.sh of that tag)
set -e
T=$1; O=gpurun_out; P=profiles
cp_if() { [ -s "$1" ] && cp "$1" "$2" || echo "missing: $1"; }
for f in bench.json kernel_stats_all.csv kernel_stats_steps.csv step_timeline.txt pmc_fetch_size.csv \
         pmc_write_size.csv pmc_summary.json; do cp_if $O/${T}_$f $P/${T}_$f; done
cp_if $O/${T}_bench_plain.json $P/${T}_bench_noprof.json
for c in 3 4 5; do
  cp_if $O/${T}_c${c}_bench.json $P/${T}_c${c}_bench.json
  cp_if $O/${T}_c${c}_prof_kernel_stats_all.csv $P/${T}_c${c}_kernel_stats_all.csv
  cp_if $O/${T}_c${c}_top.txt $P/${T}_c${c}_kernel_top.txt
done
for s in sharded_n1 c4_n1; do
  for f in bench.json kernel_stats.csv timeline.txt; do cp_if $O/${T}_${s}_$f $P/${T}_${s}_$f; done
done
