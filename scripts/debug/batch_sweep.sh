# proofs/s by batch size, comb (tables forced from batch 65 on) vs bucket pipeline; same box
for B in 96 128 256 512; do
  for cfg in "ZKPROVE_COMB_MIN_BATCH=65" "ZKPROVE_COMB=0"; do
    env $cfg python bench.py --batch $B --steps 10 --warmup 3 --cpu-proofs 0 --abi-requests 0 --no-aes256 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('B=%4d %-28s chacha %8.1f proofs/s (%.1f ms/step)' % ($B, '$cfg', d['value'], d['ms_per_step']))"
  done
done
