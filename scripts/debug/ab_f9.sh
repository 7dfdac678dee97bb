# same-box A/B of the 9-limb G1 accumulation: ZKPROVE_MSM_F9=0 (8 x 32 walk) first, then the default
cd /root/repo
for v in 0 1; do
  export ZKPROVE_MSM_F9=$v
  python bench.py --cpu-proofs 0 --steps 8 2>/dev/null > /tmp/ab_$v.json
  python -c "
import json
d=json.load(open('/tmp/ab_$v.json')); a=d['aes256_ctr']
print('F9=$v', round(d['value']), round(d['ms_per_step'],1), 'z', round(d['phases_ms']['z_segments'],1), 'g1', round(d['phases_ms']['msm_g1'],1), d['gather_spot_check_verified'], '| aes', round(a['value']), round(a['ms_per_step'],1), 'g1', round(a['phases_ms']['msm_g1'],1), a['gather_spot_check_verified'])
"
done
