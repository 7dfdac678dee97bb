cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e; rm -rf $O; mkdir -p $O
export ZKPROVE_MSM_TREE=1
python bench.py --no-aes256 --cpu-proofs 0 --steps 5 > $O/bench_tree.json 2> $O/bench_tree.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --no-aes256 --cpu-proofs 0 --steps 3 --warmup 1 > /dev/null 2> $O/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --no-aes256 --steps 1 --warmup 1 --cpu-proofs 0 --no-pipeline > /dev/null 2> $O/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 bench.py --no-aes256 --steps 1 --warmup 1 --cpu-proofs 0 --no-pipeline > /dev/null 2> $O/pmc_write.log
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
ls -R $O | head -30
