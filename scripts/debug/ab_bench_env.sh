#!/bin/bash
# same-box A/B of one environment switch on the bench line: bash scripts/debug/ab_bench_env.sh "<VAR=value>" <tag> [reps]
cd $GRAFT_REPO_ROOT
sw=$1; tag=$2; reps=${3:-2}
for rep in $(seq 1 $reps); do
  python3 bench.py --cpu-proofs 0 --abi-requests 0 > gpurun_out/${tag}_base_$rep.json 2>/dev/null
  env $sw python3 bench.py --cpu-proofs 0 --abi-requests 0 > gpurun_out/${tag}_sw_$rep.json 2>/dev/null
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${tag}_*.json")):
    l=[json.loads(x) for x in open(f) if x.startswith("{")]
    if l: print(f, round(l[0]["value"]), round(l[0].get("aes256_ctr",{}).get("value",0)), l[0]["ms_per_step"])
PY
