#!/bin/bash
# kernel-by-kernel timeline of ONE ProveBatch call of N requests: bash scripts/debug/bn_prof.sh <alg> <N> <window ms>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
alg=${1:-0}; n=${2:-32}; win=${3:-12}; extra=${4:-}
O=gpurun_out/bnprof_${alg}_${n}; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 scripts/debug/bn_trace.py $alg $n $extra > $O/out.txt 2> $O/log.txt
grep ProveBatch $O/out.txt
F=$(find $O -name "*kernel_trace.csv" | head -1)
python3 scripts/debug/timeline_call.py "$F" 0.05 $win > $O/timeline.txt
python3 scripts/debug/level_times.py "$F" k_solve_level $win > $O/level_times.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
