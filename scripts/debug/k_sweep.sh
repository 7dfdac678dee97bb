# usage (GPU box): bash scripts/debug/k_sweep.sh   -> one circuit per process, clump size forced: does a table twice as large still walk at the same rate?
run() {  # alg k
  ZKPROVE_COMB_K=$2 python bench.py --alg $1 --no-aes256 --steps 10 --warmup 2 --cpu-proofs 0 --abi-requests 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); q=d['roofline'].get('quotient_msm',{})
print('alg $1 k $2: %8.1f proofs/s (%.1f ms, walk %.1f) table %s MiB clumps %s' % (d['value'], d['ms_per_step'], d['phases_ms']['z_segments'], q.get('table_MiB'), q.get('clumps_or_windows')))"
}
run 0 19; run 0 20; run 2 18; run 2 19
