#!/bin/bash
# rocprofv3 kernel statistics of a python tool: tools/prof_run.sh <out name> <script> [args...]  -> gpurun_out/<out name>_kernel_stats.csv
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1; shift
SCRIPT=$R/$1; shift
O=$R/gpurun_out/prof_$NAME
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o p -- python3 $SCRIPT "$@" > $O/run.log 2>&1 || { tail -20 $O/run.log; exit 1; }
cd $R
DB=$(find $O -name "*_results.db" | head -1)
python tools/rocpd_stats.py $DB $R/gpurun_out/${NAME}_kernel_stats.csv
