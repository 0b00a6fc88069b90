for src in stream buffer; do for e in 195313 97656 48828 24414; do
ORBIT_BENCH_FORCE_DIST=1 timeout 300 python bench.py --meshlet-source $src --entities $e --cpu-sample-entities 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['per_rank']
print('$src', $e, 'step', round(d['ms_per_step'],4), 'cull', round(p['cull_ms_max'],4), 'xchg', p['exchange_ms_max'] and round(p['exchange_ms_max'],4), 'expand', p['expand_ms_max'] and round(p['expand_ms_max'],4), 'eval', round(d['roofline']['avg_launch_ms'],4))"
done; done
