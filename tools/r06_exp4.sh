#!/bin/bash
O=gpurun_out/r06_exp4; mkdir -p $O
run() { name=$1; shift; timeout 300 python bench.py --cpu-sample-entities 0 "$@" > $O/$name.json 2> $O/$name.err; python3 -c "
import json,sys
d=json.load(open('$O/$name.json')); c=d['config']
print('$name', 'ms_per_step', round(d['ms_per_step'],4), 'warmup_effective', c['warmup_effective'], 'cold', round(c['ms_per_step_without_settle'] or 0,4), 'p8d', round(d['protocol_8d']['ms_per_step_median'],4))
"; }
run base_k100_w0 --steps 100 --runtime-warm-frames 0
run prepay3000_k100_w0 --steps 100 --runtime-warm-frames 0 --prepay-launches 3000
run prepay6000_k100_w0 --steps 100 --runtime-warm-frames 0 --prepay-launches 6000
run base_k100_w1024 --steps 100
run base_k20_w0 --steps 20 --warmup 5 --runtime-warm-frames 0
run prepay3000_k20_w0 --steps 20 --warmup 5 --runtime-warm-frames 0 --prepay-launches 3000
run default --steps 20 --warmup 5
