# perf triage of the headline bench under the debug switches of the library (ORBIT_SP_DEBUG bit mask)
for f in 0 8; do echo -n "flags=$f "; ORBIT_SP_DEBUG=$f timeout 300 python bench.py --steps 20 --warmup 3 --cpu-sample-entities 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],4), round(d['ms_per_step'],4))"; done
