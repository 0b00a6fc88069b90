# how often does a world-of-one sharded run (real exchange code, RCCL communicator of one) contain a stalled step?
for k in 1 2 3 4 5 6 7 8; do
ORBIT_BENCH_FORCE_DIST=1 timeout 300 python bench.py --entities 48828 --expand ${X:-all} --cpu-sample-entities 0 --steps ${K:-30} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('step', round(d['ms_per_step'],4), 'cold', round(d['config']['ms_per_step_without_settle'],4))"
done
