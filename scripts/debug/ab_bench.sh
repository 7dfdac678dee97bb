# usage (GPU box): bash scripts/debug/ab_bench.sh "<ENV=val ...>" "<ENV=val ...>" ...   -> proofs/s of both circuits per environment, same box
for cfg in "$@"; do
  env $cfg python bench.py --steps 10 --warmup 2 --cpu-proofs 0 --abi-requests 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); a=d.get('aes256_ctr',{})
print('%-50s chacha %8.1f (%.1f ms, walk %.1f)   aes256 %8.1f (%.1f ms, walk %.1f)' % ('$cfg', d['value'], d['ms_per_step'], d['phases_ms']['z_segments'], a.get('value',0), a.get('ms_per_step',0), a.get('phases_ms',{}).get('z_segments',0)))"
done
