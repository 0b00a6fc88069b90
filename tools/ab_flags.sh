#!/bin/bash
# same-box alternating A/B of ORBIT_SP_DEBUG flag sets on the headline bench (serial frames): tools/ab_flags.sh 0 64 [reps]
A=${1:-0}; B=${2:-64}; R=${3:-3}
for w in 1 2; do python bench.py --cpu-sample-entities 0 --frames-in-flight 1 >/dev/null 2>&1; done
for r in $(seq $R); do for f in $A $B; do
  ORBIT_SP_DEBUG=$f python bench.py --cpu-sample-entities 0 --frames-in-flight 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('flags $f', 'step', round(d['ms_per_step'],4), 'eval', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4), 'op', round(d['roofline']['op']['avg_ms'],4))"
done; done
