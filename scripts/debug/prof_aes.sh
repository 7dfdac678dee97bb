TAG=${1:-q}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_aes -o aes -- python3 bench.py --cpu-proofs 0 --alg 2 --abi-requests 0 > $O/bench_aes_under_rocprof.json 2> $O/stats_aes.log
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
python3 - "$O/stats_aes/aes_kernel_stats.csv" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:34]:
    print("%-100s %6s %10.2f ms avg %9.3f ms  %5s%%" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6, r["Percentage"]))
PY
