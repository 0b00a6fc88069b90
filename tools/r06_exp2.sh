#!/bin/bash
O=gpurun_out/r06_exp2; mkdir -p $O
N=700 E=24000 EVENTS_PER_FRAME=0 timeout 200 python tools/stall_find.py > $O/stall_noevents.log 2>&1
N=700 E=24000 EVENTS_PER_FRAME=2 timeout 200 python tools/stall_find.py > $O/stall_2events.log 2>&1
N=700 E=24000 REUSE_EVENT=1 timeout 200 python tools/stall_find.py > $O/stall_reuse.log 2>&1
N=700 E=24000 PRE_EVENTS=1000 timeout 200 python tools/stall_find.py > $O/stall_pre_events.log 2>&1
N=700 E=24000 SOURCE=buffer timeout 200 python tools/stall_find.py > $O/stall_buffer.log 2>&1
for i in 1 2 3; do timeout 300 python tools/ab_alloc.py > $O/alloc_$i.log 2>&1; tail -1 $O/alloc_$i.log | cut -c1-600; done
grep -v amdgpu $O/stall_*.log
