#!/bin/bash
# per-kernel durations of the transform stage alone (zkp_bench_to_coset), fused middle pass on / off
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in 1 0; do
  export ZKPROVE_NTT_FUSE=$f
  O=gpurun_out/nttprof_$f; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 scripts/debug/ntt_bench.py libzkprove_test.so > $O/out.txt 2> $O/log.txt
  echo "fuse=$f"; cat $O/out.txt
  F=$(find $O -name "*kernel_stats.csv" | head -1)
  python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("%-100s calls %5s  avg %9.3f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
done
