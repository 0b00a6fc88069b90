#!/bin/bash
O=gpurun_out/r06_exp3; mkdir -p $O
AMD_LOG_LEVEL=4 N=450 E=24000 EVENTS_PER_FRAME=0 timeout 300 python tools/stall_find.py > $O/stall.out 2> /tmp/amdlog.txt
grep -v amdgpu $O/stall.out
wc -l /tmp/amdlog.txt
python3 - <<'PY'
import re
rows=[]
for l in open('/tmp/amdlog.txt', errors='replace'):
    m=re.search(r'(\d+) us:', l)
    if m: rows.append((int(m.group(1)), l.rstrip()[:260]))
gaps=sorted(((rows[i+1][0]-rows[i][0], i) for i in range(len(rows)-1)), reverse=True)[:6]
out=open('gpurun_out/r06_exp3/around_gaps.txt','w')
for g,i in sorted(gaps, key=lambda t:t[1]):
    out.write(f"==== gap {g} us after line {i} of {len(rows)}\n")
    for t,l in rows[max(0,i-14):i+14]:
        out.write(l+"\n")
out.close()
PY
head -c 20000 gpurun_out/r06_exp3/around_gaps.txt | tail -c 9000
