# usage (GPU box): bash scripts/debug/prof_quick.sh <tag>  -> rocprofv3 kernel stats of the ChaCha20 and AES-256 bench legs under gpurun_out/<tag>/
TAG=${1:-q}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cc -o cc -- python3 bench.py --cpu-proofs 0 --no-aes256 > $O/bench_chacha_under_rocprof.json 2> $O/stats_cc.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_aes -o aes -- python3 bench.py --cpu-proofs 0 --alg 2 > $O/bench_aes_under_rocprof.json 2> $O/stats_aes.log
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
for f in $O/stats_cc/cc_kernel_stats.csv $O/stats_aes/aes_kernel_stats.csv; do echo == $f; python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print("%-90s %6s %10.2f ms avg %9.3f ms  %5s%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6, r["Percentage"]))
PY
done
