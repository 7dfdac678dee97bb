TAG=${1:-qv}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o v -- python3 -m pytest tests/test_gpu_configs.py -q -x -k "verify_batch and 0-1024" > $O/log.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
python3 - "$O/stats/v_kernel_stats.csv" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(x in r["Name"] for x in ("k_vb_","g2_validate","decompress","k_bsb22","k_msm_ones9","k_msm_sort","k_sum_groups")):
        print("%-80s %5s  total %9.2f ms  avg %8.3f ms" % (r["Name"][:80], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6))
PY
