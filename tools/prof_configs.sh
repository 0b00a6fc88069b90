# kernel stats of the secondary configurations (tools/bench_configs.py) -> stdout
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_configs
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py > $out.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1); grep orbit $f | sed 's/orbit::(anonymous namespace):://; s/(orbit::[A-Za-z]*)//' | cut -d, -f1-4 | cut -c1-110
