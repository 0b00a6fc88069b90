# The sharded engine's stages alone on one device, for the whole scene and for 1/2, 1/4, 1/8 of it (a rank's shard at
# N = 2, 4, 8): a world of one with the real exchange code (device-signalled orbit_exchange_list through IPC-mapped
# buffers — the device-side cost without a wire), for the three things a step can end in (--expand own / none),
# with the default two frames in flight and with strictly serial frames (F=1).
# Per row: wall / K, the median and the maximum interval between consecutive steps' exchange completions inside the timed
# region (config.step_intervals), what the host spent enqueueing a step, the stages alone (cull, exchange, expansion of the
# whole list), the evaluation launch (with F=2: overlapped by the other frame's launches), the list's bytes.
# usage (GPU box, repo root): bash tools/shard_pieces.sh [sources] [frames-in-flight values]
for src in ${1:-stream buffer}; do for f in ${2:-2 1}; do for e in 195313 97656 48828 24414; do for x in own none; do
ORBIT_BENCH_FORCE_DIST=1 timeout 300 python bench.py --meshlet-source $src --entities $e --expand $x --frames-in-flight $f --steps 30 --cpu-sample-entities 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['per_rank']; c=d['config']; si=c['step_intervals'] or {}
r=lambda v: v if v is None else round(v,4)
print('$src', 'records', $e, 'expand=$x', 'F=$f', c['signalling'], 'step', r(d['ms_per_step']), 'interval median', r(si.get('median_ms')), 'max', r(si.get('max_ms')), 'host', r(c.get('host_enqueue_ms_per_step')), 'cull', r(p['cull_ms_max']), 'xchg', r(p['exchange_ms_max']), 'alt', (d.get('transport_alt') or {}).get('name'), r((d.get('transport_alt') or {}).get('ms_per_step')), 'eval', r(d['roofline']['avg_launch_ms']), 'list_MB', round((c.get('list_bytes_per_rank_max') or 0) / 1e6, 2))"
done; done; done; done
