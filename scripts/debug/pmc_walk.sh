#!/bin/bash
# SQ counters of the dominant kernel k_comb_walk (quotient MSM, csrc/msm_comb.hip): one rocprofv3 pass per counter set over
# `bench.py --steps 1 --warmup 1 --no-pipeline` (ChaCha20, then AES-256); summary -> gpurun_out/pmc_walk/summary.json
# usage (GPU box): bash scripts/debug/pmc_walk.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_walk; rm -rf $O; mkdir -p $O
for ALG in 0 2; do
  if [ "$ALG" = "0" ]; then EXTRA="--no-aes256"; else EXTRA="--alg 2"; fi
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/a${ALG}_s$i -o p -- python3 bench.py --steps 1 --warmup 1 --cpu-proofs 0 --no-pipeline --abi-requests 0 $EXTRA > /dev/null 2> $O/a${ALG}_s$i.log
  done
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for alg in (0, 2):
    agg = collections.defaultdict(list)
    for f in glob.glob('gpurun_out/pmc_walk/a%d_s*/*counter_collection.csv' % alg) + glob.glob('gpurun_out/pmc_walk/a%d_s*/*/*counter_collection.csv' % alg):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            # the quotient walk: the k_comb_walk<Fp, *> launch with the largest grid (wire combs launch the same kernel on small tables)
            if 'k_comb_walk' in k and 'Fp2' not in k:
                agg[r['Counter_Name']].append((float(r.get('Grid_Size', 0) or 0), float(r['Counter_Value'])))
    if not agg:
        continue
    row = {}
    for c, v in agg.items():
        gmax = max(g for g, _ in v)
        vals = [x for g, x in v if g == gmax]
        row[c] = max(vals)
        row['launch_threads'] = gmax
    out[str(alg)] = row
json.dump(out, open('gpurun_out/pmc_walk/summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
