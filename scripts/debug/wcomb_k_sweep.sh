for k in 16 17 18; do ZKPROVE_WCOMB_K=$k python bench.py --alg 2 --no-aes256 --steps 10 --warmup 2 --cpu-proofs 0 --abi-requests 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d['roofline'].get('wire_msms',{})
print('wcomb k $k: %8.1f proofs/s (%.1f ms, walk %.1f) wire tables %s MiB' % (d['value'], d['ms_per_step'], d['phases_ms']['z_segments'], w.get('table_MiB')))"; done
