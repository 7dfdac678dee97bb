cd $GRAFT_REPO_ROOT
AB=$GRAFT_REPO_ROOT/zk-symmetric-crypto_amd/libzkprove_ab.so
O=gpurun_out/ab_fold_prio.txt; : > $O
for rep in 1 2; do
 for alg in 2 0; do
  echo "== base alg $alg rep $rep" >> $O; python3 scripts/debug/small_batch_latency.py $alg 1 8 16 32 64 128 --tables >> $O 2>&1
  echo "== prio3 alg $alg rep $rep" >> $O; ZKPROVE_LIB=$AB python3 scripts/debug/small_batch_latency.py $alg 1 8 16 32 64 128 --tables >> $O 2>&1
 done
done
cat $O
