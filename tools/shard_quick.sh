# the stream path's shard culls into the record list, twice each (quick form of shard_pieces.sh); GPU box, repo root
for e in 195313 97656 48828 24414; do for k in 1 2; do
ORBIT_BENCH_FORCE_DIST=1 timeout 300 python bench.py --meshlet-source stream --entities $e --cpu-sample-entities 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['per_rank']
print('stream records', $e, 'step', round(d['ms_per_step'],4), 'cull', round(p['cull_ms_max'],4), 'xchg', round(p['exchange_ms_max'],4), 'eval', round(d['roofline']['avg_launch_ms'],4))"
done; done
