#!/bin/bash
# kernel stats of one bench configuration under rocprofv3:  scripts/prof_config.sh <tag> <bench args...>
set -e -o pipefail
T=$1; shift
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $O/prof_$T -o p -- python3 $R/bench.py "$@" --no-cpu \
  > $O/${T}_bench.json 2> $O/${T}_bench.err
cd $R
DB=$(find $O/prof_$T -name '*.db' | head -1)
python3 scripts/db_to_stats.py $DB $O/${T}_kernel_stats_all.csv
rm -rf $O/prof_$T
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/${T}_kernel_stats_all.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:22]:
    print("%-70s calls %5s avg %10.1f us total %9.1f ms"%(r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
