# usage (on the GPU box): bash scripts/profile_round.sh <tag>      -> raw rocprofv3 output + bench lines under gpurun_out/<tag>/
set -x
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --alg 1 --cpu-proofs 0 --steps 5 --abi-requests 0 > $O/bench_aes128.json 2> /dev/null
python bench.py --no-pipeline --cpu-proofs 0 --steps 5 --abi-requests 0 > $O/bench_single_stream.json 2> /dev/null
ZKPROVE_COMB=0 python bench.py --cpu-proofs 0 --steps 5 --abi-requests 0 > $O/bench_bucket_pipeline.json 2> /dev/null
python scripts/gpu_microbench.py $O/microbench.json > $O/microbench.txt 2>&1
# The per-circuit profiling runs below load ONE circuit; alone it would take a larger table (the budget looks at what is loaded and
# fills the device).  They are pinned to the table shapes of the two-circuit bench line above -- clump size k and the number of leading
# clumps of k + 1 bases, read back from its roofline.quotient_msm -- so that the kernel durations and the PMC traffic describe the
# launches `roofline` is about.
# (set in the environment of the rocprofv3 command itself: nothing but the program may stand behind `--`)
read K0 N10 K2 N12 <<< $(python3 - $O/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
def shape(r, n):
    q = r["quotient_msm"]; k, c = q["k_or_c"], q["clumps_or_windows"]
    return k, max(0, n - c * k)
a = shape(d["roofline"], 65536); b = shape(d["aes256_ctr"]["roofline"], 131072)
print(a[0], a[1], b[0], b[1])
PY
)
echo "table shapes of the bench line: chacha20 k=$K0 n1=$N10, aes-256 k=$K2 n1=$N12" | tee $O/shapes.txt
ZKPROVE_COMB_K=$K0 ZKPROVE_COMB_N1=$N10 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cc -o cc -- python3 bench.py --cpu-proofs 0 --no-aes256 --abi-requests 0 > $O/bench_chacha_under_rocprof.json 2> $O/stats_cc.log
ZKPROVE_COMB_K=$K2 ZKPROVE_COMB_N1=$N12 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_aes -o aes -- python3 bench.py --cpu-proofs 0 --alg 2 --abi-requests 0 > $O/bench_aes_under_rocprof.json 2> $O/stats_aes.log
ZKPROVE_COMB_K=$K0 ZKPROVE_COMB_N1=$N10 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 1 --cpu-proofs 0 --no-pipeline --no-aes256 --abi-requests 0 > /dev/null 2> $O/pmc_fetch.log
ZKPROVE_COMB_K=$K0 ZKPROVE_COMB_N1=$N10 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 1 --warmup 1 --cpu-proofs 0 --no-pipeline --no-aes256 --abi-requests 0 > /dev/null 2> $O/pmc_write.log
# the same two passes for the AES-256 leg (its quotient walk is the dominant kernel of that half of the metric)
ZKPROVE_COMB_K=$K2 ZKPROVE_COMB_N1=$N12 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_aes -o f -- python3 bench.py --steps 1 --warmup 1 --cpu-proofs 0 --no-pipeline --alg 2 --abi-requests 0 > /dev/null 2> $O/pmc_fetch_aes.log
ZKPROVE_COMB_K=$K2 ZKPROVE_COMB_N1=$N12 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_aes -o w -- python3 bench.py --steps 1 --warmup 1 --cpu-proofs 0 --no-pipeline --alg 2 --abi-requests 0 > /dev/null 2> $O/pmc_write_aes.log
python scripts/host_feed_bench.py > $O/host_feed.jsonl 2> /dev/null
# one steady-state step of each circuit as a kernel timeline (which stream ran what when): scripts/debug/timeline_step.py
for A in 0 2; do
  ZKPROVE_COMB_K=$([ $A = 0 ] && echo $K0 || echo $K2) ZKPROVE_COMB_N1=$([ $A = 0 ] && echo $N10 || echo $N12) rocprofv3 --kernel-trace --output-format csv -d $O/tr$A -o t -- python3 bench.py --cpu-proofs 0 --alg $A --no-aes256 --abi-requests 0 > /dev/null 2> $O/tr$A.log
  python3 scripts/debug/timeline_step.py $(find $O/tr$A -name "*kernel_trace.csv" | head -1) 0.3 > $O/timeline_alg$A.txt
  rm -rf $O/tr$A
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
ls -R $O | head -40
