#!/bin/bash
# kernel-by-kernel timeline of ONE synchronous Prove() call (the last of 12) for ChaCha20 (0) and AES-256 (2)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for alg in 0 2; do
  O=gpurun_out/b1prof_$alg; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 scripts/debug/b1_trace.py $alg > $O/out.txt 2> $O/log.txt
  cat $O/out.txt | grep single
  F=$(find $O -name "*kernel_trace.csv" | head -1)
  python3 scripts/debug/timeline_call.py "$F" 0.03 $( [ $alg = 0 ] && echo 7 || echo 40 ) > $O/timeline.txt
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
done
