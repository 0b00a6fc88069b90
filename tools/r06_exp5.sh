#!/bin/bash
O=gpurun_out/r06_exp5; mkdir -p $O
run() { name=$1; shift; timeout 300 python bench.py --cpu-sample-entities 0 "$@" > $O/$name.json 2> $O/$name.err; python3 -c "
import json,sys
d=json.load(open('$O/$name.json')); c=d['config']
print('$name', 'ms_per_step', round(d['ms_per_step'],4), 'warmup_effective', c['warmup_effective'], 'from_idle', round(c['ms_per_step_without_settle'] or 0,4), 'p8d', round(d['protocol_8d']['ms_per_step_median'],4), 'eval', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],3))
" || tail -3 $O/$name.err; }
for i in 1 2 3; do
run stream_k20_$i --steps 20 --warmup 5
run frames_k20_$i --steps 20 --warmup 5 --settle-with frames
done
run stream_k30 
run stream_k100 --steps 100
run stream_k300 --steps 300
python -m pytest tests/test_frame_late_gpu.py -x -q 2>&1 | tail -2
timeout 600 python tools/bench_configs.py > $O/configs.json 2> $O/configs.err; tail -2 $O/configs.err
python3 -c "
import json
d=json.load(open('$O/configs.json'))
print(json.dumps(d['frame_configs_3_and_4']['ms'])); print(json.dumps(d['frame_configs_3_and_4']['graph_replay_ms'])); print(d['config4']['gpu'])
"
