# kernel stats of the secondary configurations (tools/bench_configs.py) -> stdout (GPU box: bash tools/prof_configs.sh)
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_configs
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py > $out.json 2> $out.err
f=$(find $out -name "*kernel_stats.csv" | head -1)
grep -i "orbit\|Name" $f | sed 's/orbit::(anonymous namespace):://g; s/(orbit::[A-Za-z]*)//g; s/orbit:://g' | cut -d, -f1-8 | cut -c1-150
find $out -name "*kernel_trace.csv" -delete
