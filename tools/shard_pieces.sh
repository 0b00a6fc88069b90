# The sharded engine's stages alone on one device, for the whole scene and for 1/2, 1/4, 1/8 of it (a rank's shard at
# N = 2, 4, 8), from both meshlet sources and for both compact list formats (12-B records, 8-B items): cull, exchange
# of the list (a world of one: the device-side cost without a wire), list bytes, and — with --expand all — the
# expansion of the list.   usage (GPU box, repo root): bash tools/shard_pieces.sh
for src in stream buffer; do for fmt in records visible; do for e in 195313 97656 48828 24414; do for x in none all; do
ORBIT_BENCH_FORCE_DIST=1 timeout 300 python bench.py --meshlet-source $src --exchange $fmt --entities $e --expand $x --cpu-sample-entities 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['per_rank']
print('$src', '$fmt', $e, 'expand=$x', 'step', round(d['ms_per_step'],4), 'cull', round(p['cull_ms_max'],4), 'xchg', p['exchange_ms_max'] and round(p['exchange_ms_max'],4), 'expand', p['expand_ms_max'] and round(p['expand_ms_max'],4), 'eval', round(d['roofline']['avg_launch_ms'],4), 'list_MB', round(d['config'].get('list_bytes_per_rank_max', 0) / 1e6, 2))"
done; done; done; done
