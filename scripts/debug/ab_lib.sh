#!/bin/bash
# same-box A/B of the product library against another build of it (zk-symmetric-crypto_amd/libzkprove_ab.so, linked with the test entries):
#   bash scripts/debug/ab_lib.sh <tag> [reps]     single-call latencies of both circuits, then the bench line, variants alternating
cd $GRAFT_REPO_ROOT
tag=$1; reps=${2:-2}
AB=$GRAFT_REPO_ROOT/zk-symmetric-crypto_amd/libzkprove_ab.so
O=gpurun_out/${tag}_latency.txt; : > $O
for rep in $(seq 1 $reps); do
  for alg in 2 0; do
    echo "== product alg $alg rep $rep" >> $O; python3 scripts/debug/small_batch_latency.py $alg 1 8 16 64 128 --tables 2>&1 | grep " n " >> $O
    echo "== other build alg $alg rep $rep" >> $O; ZKPROVE_LIB=$AB python3 scripts/debug/small_batch_latency.py $alg 1 8 16 64 128 --tables 2>&1 | grep " n " >> $O
  done
done
cat $O
for rep in $(seq 1 $reps); do
  python3 bench.py --cpu-proofs 0 --abi-requests 0 > gpurun_out/${tag}_bench_product_$rep.json 2>/dev/null
  ZKPROVE_LIB=$AB python3 bench.py --cpu-proofs 0 --abi-requests 0 > gpurun_out/${tag}_bench_other_$rep.json 2>/dev/null
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${tag}_bench_*.json")):
    l=[json.loads(x) for x in open(f) if x.startswith("{")]
    if l: print(f, round(l[0]["value"]), round(l[0].get("aes256_ctr",{}).get("value",0)), l[0]["ms_per_step"])
PY
