for p in 1024 1536 3072 4096; do
  ZKPROVE_MSM_PIECE=$p python bench.py --cpu-proofs 0 --steps 6 --no-aes256 2>/dev/null > /tmp/ps_$p.json
  python -c "
import json
d=json.load(open('/tmp/ps_$p.json')); print('piece', $p, round(d['value']), d['phases_ms']['z_segments'], d['phases_ms']['msm_g1'])
"
done
