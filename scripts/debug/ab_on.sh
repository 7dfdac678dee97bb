# same-box A/B of an opt-in switch: ab_on.sh VAR  -> bench with VAR unset, then VAR=1
cd /root/repo
for v in "" 1; do
  if [ -n "$v" ]; then export $1=$v; else unset $1; fi
  python bench.py --cpu-proofs 0 --steps 8 2>/dev/null > /tmp/ab_$v.json
  python -c "
import json
d=json.load(open('/tmp/ab_$v.json')); a=d['aes256_ctr']
print('$1=$v', round(d['value']), round(d['ms_per_step'],1), 'ntt', round(d['phases_ms']['ntt'],1), 'z', round(d['phases_ms']['z_segments'],1), 'g1', round(d['phases_ms']['msm_g1'],1), d['gather_spot_check_verified'], '| aes', round(a['value']), round(a['ms_per_step'],1), 'ntt', round(a['phases_ms']['ntt'],1), 'g1', round(a['phases_ms']['msm_g1'],1), a['gather_spot_check_verified'])
"
done
