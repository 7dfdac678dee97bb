# sweep_env.sh VAR v1 v2 ...  -> one bench line summary per value (same box)
cd /root/repo
VAR=$1; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
  python bench.py --cpu-proofs 0 --steps 6 2>/dev/null > /tmp/sw.json
  python -c "
import json
d=json.load(open('/tmp/sw.json')); a=d['aes256_ctr']
print('$VAR=$v', round(d['value']), round(d['ms_per_step'],1), 'ntt', round(d['phases_ms']['ntt'],1), 'z', round(d['phases_ms']['z_segments'],1), 'g1', round(d['phases_ms']['msm_g1'],1), d['gather_spot_check_verified'], '| aes', round(a['value']), round(a['ms_per_step'],1), 'z', round(a['phases_ms']['z_segments'],1), 'g1', round(a['phases_ms']['msm_g1'],1), a['gather_spot_check_verified'])
"
done
