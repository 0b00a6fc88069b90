# How often does a world-of-one sharded run (real exchange code, RCCL communicator of one) contain a stalled step?
# Round 3: with --expand all (three streams) about a quarter of the 30-step timed regions hold ONE step of 35-70 ms
# (never the first 35 steps of a process, never N = 1, not the library's timing events: --time-every 1000 changes
# nothing; over 2000 steps the average moves by 4 %); with --expand none (the default) 1 mild outlier in 24 runs.
# usage: [X=none|all] [K=steps] [EXTRA='...'] bash tools/stall_probe.sh
for k in 1 2 3 4 5 6 7 8; do
ORBIT_BENCH_FORCE_DIST=1 timeout 300 python bench.py --entities 48828 --expand ${X:-own} --cpu-sample-entities 0 --steps ${K:-30} ${EXTRA:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('step', round(d['ms_per_step'],4), 'cold', round(d['config']['ms_per_step_without_settle'],4))"
done
