#!/bin/bash
# round 6, experiment batch 1 (GPU box): frame_late + log2 guard tests, arena probe A/B in fresh processes, the one long
# stall against pre-launches, the configs tool
O=gpurun_out/r06_exp1; mkdir -p $O
python -m pytest tests/test_frame_late_gpu.py tests/test_compute_clusters_gpu.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2 3 4 5 6; do timeout 300 python tools/ab_arena.py > $O/arena_$i.log 2>&1; tail -1 $O/arena_$i.log; done
for pre in 0 4000 20000; do PRE_LAUNCHES=$pre N=900 timeout 300 python tools/stall_find.py > $O/stall_pre$pre.log 2>&1; done
PRE_LAUNCHES=0 N=900 E=24000 timeout 300 python tools/stall_find.py > $O/stall_small_pre0.log 2>&1
PRE_LAUNCHES=20000 N=900 E=24000 timeout 300 python tools/stall_find.py > $O/stall_small_pre20000.log 2>&1
timeout 600 python tools/bench_configs.py > $O/configs.json 2> $O/configs.err; tail -2 $O/configs.err
