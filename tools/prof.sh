# usage: bash tools/prof.sh <name> [bench args]  -> kernel stats of the orbit kernels + bench line
name=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$name
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 "$@" > $out.log 2>&1
grep '^{' $out.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], d['cpu_baseline'])"
f=$(find $out -name "*kernel_stats.csv" | head -1); grep orbit $f | sed 's/orbit::(anonymous namespace):://; s/(orbit::[A-Za-z]*)//' | cut -c1-120
find $out -name "*kernel_trace.csv" -delete  # per-launch rows: MBs; the stats file is what is kept
