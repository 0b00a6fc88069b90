# usage (GPU box, repo root): bash tools/ab3.sh <rounds> <variant> [<variant> ...]
# Like ab2.sh but prints the whole meshlet_cull op (eval + scan + emit) beside the eval kernel and the step.
rounds=$1; shift
cp tools/variants/$1.so orbit_amd/liborbit_cull.so
for i in 1 2 3; do timeout 300 python bench.py --steps 40 --warmup 5 --cpu-sample-entities 0 > /dev/null 2>&1; done
for r in $(seq $rounds); do
  for v in "$@"; do
    cp tools/variants/$v.so orbit_amd/liborbit_cull.so
    timeout 300 python bench.py --steps 40 --warmup 5 --cpu-sample-entities 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', 'eval_ms', round(r['avg_launch_ms'],4), 'op_ms', round(r['op']['avg_ms'],4), 'rest_of_op_us', round(1e3*(r['op']['avg_ms']-r['avg_launch_ms']),1), 'step_ms', round(d['ms_per_step'],4))"
  done
done
