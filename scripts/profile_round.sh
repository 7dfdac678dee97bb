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
# The per-circuit profiling runs below load ONE circuit; alone it would take the next clump size (the table budget looks at what is
# loaded: k = 20 / 19).  They are pinned to the clump sizes of the two-circuit bench line above (k = 19 ChaCha20, 18 AES-256), so that
# the kernel durations and the PMC traffic describe the launches `roofline` is about.
# (set in the environment of the rocprofv3 command itself: nothing but the program may stand behind `--`)
ZKPROVE_COMB_K=19 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cc -o cc -- python3 bench.py --cpu-proofs 0 --no-aes256 --abi-requests 0 > $O/bench_chacha_under_rocprof.json 2> $O/stats_cc.log
ZKPROVE_COMB_K=18 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_aes -o aes -- python3 bench.py --cpu-proofs 0 --alg 2 --abi-requests 0 > $O/bench_aes_under_rocprof.json 2> $O/stats_aes.log
ZKPROVE_COMB_K=19 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 1 --cpu-proofs 0 --no-pipeline --no-aes256 --abi-requests 0 > /dev/null 2> $O/pmc_fetch.log
ZKPROVE_COMB_K=19 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 1 --warmup 1 --cpu-proofs 0 --no-pipeline --no-aes256 --abi-requests 0 > /dev/null 2> $O/pmc_write.log
# the same two passes for the AES-256 leg (its quotient walk is the dominant kernel of that half of the metric)
ZKPROVE_COMB_K=18 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_aes -o f -- python3 bench.py --steps 1 --warmup 1 --cpu-proofs 0 --no-pipeline --alg 2 --abi-requests 0 > /dev/null 2> $O/pmc_fetch_aes.log
ZKPROVE_COMB_K=18 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_aes -o w -- python3 bench.py --steps 1 --warmup 1 --cpu-proofs 0 --no-pipeline --alg 2 --abi-requests 0 > /dev/null 2> $O/pmc_write_aes.log
python scripts/host_feed_bench.py > $O/host_feed.jsonl 2> /dev/null
# one steady-state step of each circuit as a kernel timeline (which stream ran what when): scripts/debug/timeline_step.py
for A in 0 2; do
  ZKPROVE_COMB_K=$([ $A = 0 ] && echo 19 || echo 18) rocprofv3 --kernel-trace --output-format csv -d $O/tr$A -o t -- python3 bench.py --cpu-proofs 0 --alg $A --no-aes256 --abi-requests 0 > /dev/null 2> $O/tr$A.log
  python3 scripts/debug/timeline_step.py $(find $O/tr$A -name "*kernel_trace.csv" | head -1) 0.3 > $O/timeline_alg$A.txt
  rm -rf $O/tr$A
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
ls -R $O | head -40
