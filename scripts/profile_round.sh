set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01h; rm -rf $O; mkdir -p $O
python bench.py --alg 0 > $O/bench_chacha.json 2> $O/bench_chacha.err
python bench.py --alg 2 --cpu-proofs 48 > $O/bench_aes256.json 2> $O/bench_aes256.err
python bench.py --alg 1 --cpu-proofs 0 > $O/bench_aes128.json 2> $O/bench_aes128.err
python bench.py --alg 0 --cpu-proofs 0 --no-pipeline > $O/bench_chacha_single.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cc -o cc -- python3 bench.py --cpu-proofs 0 > $O/bench_chacha_under_rocprof.json 2> $O/stats_cc.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_aes -o aes -- python3 bench.py --cpu-proofs 0 --alg 2 > $O/bench_aes_under_rocprof.json 2> $O/stats_aes.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 1 --cpu-proofs 0 --no-pipeline > /dev/null 2> $O/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 1 --warmup 1 --cpu-proofs 0 --no-pipeline > /dev/null 2> $O/pmc_write.log
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
ls -R $O | head -40
