cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/timeline; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 bench.py --cpu-proofs 0 --steps 4 --warmup 2 --no-aes256 > $O/bench.json 2> $O/log.txt
python3 - <<'PY'
import csv, collections
rows=list(csv.DictReader(open("gpurun_out/timeline/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# find the last 3 accumulate<Fp,true> launches: a step = from one accumulate start to the next
acc=[r for r in rows if "k_msm_accumulate<zk::Field<zk::FpParams>, true>" in r["Kernel_Name"]]
a0,a1=acc[-3],acc[-2]
t0,t1=int(a0["Start_Timestamp"]),int(a1["Start_Timestamp"])
print("step ms", (t1-t0)/1e6)
win=[r for r in rows if int(r["Start_Timestamp"])>=t0 and int(r["Start_Timestamp"])<t1]
byq=collections.defaultdict(list)
for r in win: byq[r["Queue_Id"]].append(r)
for q,rs in byq.items():
    busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rs)/1e6
    print("queue",q,"kernels",len(rs),"busy ms %.1f"%busy)
# main queue = the one holding the accumulate: list its kernels with gaps
mq=a0["Queue_Id"]
prev_end=None
for r in byq[mq]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    gap=(s-prev_end)/1e6 if prev_end else 0
    nm=r["Kernel_Name"].replace("zk::","").replace("Field<FpParams>","Fp").replace("Field<FrParams>","Fr")[:50]
    if (e-s)/1e6>0.3 or gap>0.2: print("%-52s dur %7.3f gap %6.3f"%(nm,(e-s)/1e6,gap))
    prev_end=e
PY
