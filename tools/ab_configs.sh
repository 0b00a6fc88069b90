# usage (GPU box, repo root): bash tools/ab_configs.sh <rounds> <variant> [<variant> ...]
# Alternates library variants over tools/bench_configs.py (secondary configurations); prints the config-4 timings.
rounds=$1; shift
for r in $(seq $rounds); do
  for v in "$@"; do
    cp tools/variants/$v.so orbit_amd/liborbit_cull.so
    timeout 300 python tools/bench_configs.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config4']; print('$v', 'mark', round(c['mark_ms'],4), 'compact', round(c['compact_ms'],4), 'assign', round(c['assign_ms'],4))"
  done
done
