# kernel stats of a small shard (1/8 and 1/4 of config 5) through the whole sharded step, a world of one
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in 24414 48828; do
ORBIT_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_e8 -- python3 $R/bench.py --entities $e --cpu-sample-entities 0 --steps 30 > $R/gpurun_out/prof_e8.log 2>&1
f=$(find $R/gpurun_out/prof_e8 -name "*kernel_stats.csv" | head -1); echo "== $e"; grep orbit $f | sed 's/orbit::(anonymous namespace):://; s/(orbit::[A-Za-z]*)//' | cut -d, -f1-4 | cut -c1-110 | head -12
rm -rf $R/gpurun_out/prof_e8
done
