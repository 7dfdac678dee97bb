#!/bin/bash
# single-Prove latency: level launches (ZKPROVE_SOLVER_LANES=0) against the lane program, experiments build
cd $GRAFT_REPO_ROOT
E=zk-symmetric-crypto_amd/libzkprove_exp.so
for alg in 0 2; do
  for cfg in "ZKPROVE_SOLVER_LANES=0" "ZKPROVE_SOLVER_LANES_ALL=1"; do
    echo "alg $alg $cfg: $(env ZKPROVE_LIB=$E $cfg python3 scripts/debug/b1_trace.py $alg | grep single)"
  done
done
