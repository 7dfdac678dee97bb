cd $GRAFT_REPO_ROOT
EXP=$GRAFT_REPO_ROOT/zk-symmetric-crypto_amd/libzkprove_exp.so
for q in 0 128 256; do
  for alg in 2 0; do
    echo "== Q=$q alg $alg"; if [ $q = 0 ]; then ZKPROVE_LIB=$EXP python3 scripts/debug/small_batch_latency.py $alg 16 64 128 --tables 2>&1 | grep " n "; else ZKPROVE_LIB=$EXP ZKPROVE_COMB_Q=$q python3 scripts/debug/small_batch_latency.py $alg 16 64 128 --tables 2>&1 | grep " n "; fi
  done
done
