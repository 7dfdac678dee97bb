# usage (GPU box): bash scripts/debug/abi_timeline.sh <alg> <n>  -> ProveBatch of n requests: seconds per call + the host-side event timeline of the last call
ALG=${1:-2}; N=${2:-4096}
rm -f /tmp/tl.jsonl
ZKPROVE_TIMELINE=/tmp/tl.jsonl python scripts/abi_throughput.py $ALG $N 3 2>/dev/null | tail -3
python3 - <<'PY'
import json
ev=[json.loads(l) for l in open('/tmp/tl.jsonl')]
# keep the last call: events restart at small us
cut=0
for i in range(1,len(ev)):
    if ev[i]['us']<ev[i-1]['us']-100000: cut=i
ev=ev[cut:]
ev.sort(key=lambda e:e['us'])
for e in ev: print("%9.1f ms  %-10s chunk %2d who %d %s"%(e['us']/1e3,e['stage'],e['chunk'],e['who'],e['edge']))
PY
