# usage (GPU box, repo root): bash tools/ab.sh <variant> [<variant> ...]
# Runs the meshlet parity tests and the headline bench once per library variant.
for v in "$@"; do
  cp tools/variants/$v.so orbit_amd/liborbit_cull.so
  echo "== $v"
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
  for i in 1 2 3; do
    timeout 300 python bench.py --steps 40 --warmup 5 --cpu-sample-entities 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('eval_ms', round(d['roofline']['avg_launch_ms'],4), 'step_ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3))"
  done
done
