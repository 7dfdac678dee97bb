# usage (GPU box): bash scripts/debug/prof_timeline.sh <tag> <alg>  -> per-step kernel timeline (scripts/debug/timeline_step.py) of one bench leg
TAG=${1:-tl}; ALG=${2:-0}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python3 bench.py --cpu-proofs 0 --alg $ALG --no-aes256 --abi-requests 0 > $O/bench.json 2> $O/log.txt
python3 scripts/debug/timeline_step.py $(find $O/tr -name "*kernel_trace.csv" | head -1) 0.3 > $O/timeline_alg$ALG.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
