#!/bin/bash
# same-box A/B of one environment switch on the single-call latencies: bash scripts/debug/ab_env.sh "<VAR=value>" <alg> <sizes...>
cd $GRAFT_REPO_ROOT
sw=$1; alg=$2; shift 2
O=gpurun_out/ab_env_$(echo $sw | tr -c 'A-Za-z0-9\n' '_')_$alg.txt; : > $O
for rep in 1 2 3; do
  echo "== base rep $rep" >> $O; python3 scripts/debug/small_batch_latency.py $alg "$@" --tables 2>&1 | grep " n " >> $O
  echo "== $sw rep $rep" >> $O; env $sw python3 scripts/debug/small_batch_latency.py $alg "$@" --tables 2>&1 | grep " n " >> $O
done
cat $O
