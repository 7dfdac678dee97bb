cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ntt_trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 scripts/debug/ntt_bench.py libzkprove.so > $O/out.txt 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/ntt_trace/t_kernel_trace.csv")))
rows=[r for r in rows if "k_ntt_d" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# print the last 24 dispatches of each logn group (one to_coset = 12 launches)
def show(rs):
    for r in rs: print(r["Kernel_Name"][9:32], "grid", r["Grid_Size_X"], r["Grid_Size_Y"], "%.3f ms" % ((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
n=len(rows); half=n//2
print("logn16 last to_coset:"); show(rows[half-12:half])
print("logn17 last to_coset:"); show(rows[n-12:n])
PY
