# SQ counters of the dominant kernels (one pass per counter set), bench --steps 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_acc; rm -rf $O; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' > $O/sq_counters.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM SQ_IFETCH" "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/s$i -o p -- python3 bench.py --steps 1 --warmup 1 --cpu-proofs 0 --no-pipeline --no-aes256 > /dev/null 2> $O/s$i.log
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_acc/s*/*counter_collection.csv') + glob.glob('gpurun_out/pmc_acc/s*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        if any(x in k for x in ('accumulate9', 'ntt_dif_lds', 'bucket_reduce9', 'msm_ones9')):
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print('   %-28s max %.4g  n=%d' % (c, max(v), len(v)))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
