# usage: bash tools/prof_shapes.sh <tag>   (GPU box, repo root) -> kernel stats + bench line per scene shape and meshlet source
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in scattered coherent; do
  for src in stream buffer; do
    out=$R/gpurun_out/${tag}_${shape}_${src}
    rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-sample-entities 0 --scene-shape $shape --meshlet-source $src > $out.log 2>&1
    grep '^{' $out.log | tail -1 > $out.json
    python3 - "$out.json" "$shape" "$src" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], 'ms/step %.4f' % d['ms_per_step'], 'eval_ms %.4f' % d['roofline']['avg_launch_ms'], 'survivors', d['config']['survivors'])
PY
    find $out -name "*kernel_trace.csv" -delete
    f=$(find $out -name "*kernel_stats.csv" | head -1); grep orbit $f | sed 's/orbit::(anonymous namespace):://; s/(orbit::[A-Za-z]*)//' | cut -d, -f1-4 | head -6
  done
done
