# usage (GPU box, repo root): bash tools/ab2.sh <rounds> <variant> [<variant> ...]
# Alternates the variants <rounds> times after a warm-up run (the chip slows down ~8 % over the first back-to-back
# processes, see profiles/r01_notes.md "Measurement hygiene"); prints eval / step ms per run.
rounds=$1; shift
cp tools/variants/$1.so orbit_amd/liborbit_cull.so
for i in 1 2 3; do timeout 300 python bench.py --steps 40 --warmup 5 --cpu-sample-entities 0 > /dev/null 2>&1; done
for r in $(seq $rounds); do
  for v in "$@"; do
    cp tools/variants/$v.so orbit_amd/liborbit_cull.so
    timeout 300 python bench.py --steps 40 --warmup 5 --cpu-sample-entities 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'eval_ms', round(d['roofline']['avg_launch_ms'],4), 'step_ms', round(d['ms_per_step'],4))"
  done
done
