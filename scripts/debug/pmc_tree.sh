cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; rm -rf $O; mkdir -p $O
python bench.py --no-aes256 --cpu-proofs 0 --steps 5 > $O/bench_tree.json 2> $O/bench_tree.err
ZKPROVE_TREE_DEBUG=32 python bench.py --no-aes256 --cpu-proofs 0 --steps 5 > $O/bench_tree_fwdonly.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --no-aes256 --cpu-proofs 0 --steps 3 --warmup 1 > /dev/null 2> $O/stats.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc1 -o p -- python3 bench.py --no-aes256 --cpu-proofs 0 --steps 1 --warmup 1 --no-pipeline > /dev/null 2> $O/pmc1.log
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/pmc2 -o p -- python3 bench.py --no-aes256 --cpu-proofs 0 --steps 1 --warmup 1 --no-pipeline > /dev/null 2> $O/pmc2.log
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
python3 - <<'PY'
import csv, glob, json
O="gpurun_out/r02d"
for f in ("bench_tree","bench_tree_fwdonly"):
    d=json.load(open(O+"/%s.json"%f)); print(f, round(d["value"]), d["phases_ms"]["z_segments"], d["phases_ms"]["msm_g1"], d["gather_spot_check_verified"])
rows=list(csv.DictReader(open(glob.glob(O+"/stats/*/*kernel_stats.csv")[0])))
for r in rows[:14]: print(r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e6, r["Percentage"])
for d in ("pmc1","pmc2"):
    agg={}
    for r in csv.DictReader(open(glob.glob(O+"/%s/*/*counter_collection.csv"%d)[0])):
        if "k_msm_tree" in r["Kernel_Name"] or "k_msm_accumulate" in r["Kernel_Name"]:
            agg.setdefault((r["Kernel_Name"][:40], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()): print(d, k, max(v), len(v))
PY
