# per-kernel time of a ChaCha20 (default) or AES-256 (arg "2") bench run: rocprofv3 kernel stats, top kernels by total time
ALG=${1:-0}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/kstats_$ALG; rm -rf $O; mkdir -p $O
if [ "$ALG" = "0" ]; then EXTRA="--no-aes256"; else EXTRA="--alg $ALG"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 bench.py --cpu-proofs 0 --steps 6 --warmup 2 $EXTRA > $O/bench.json 2> $O/log.txt
F=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%-90s calls %5s  avg %9.3f ms  total %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
