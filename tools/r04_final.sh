# Round-4 measurement set (GPU box, repo root): bash tools/r04_final.sh  -> gpurun_out/r04_final/*
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_line.json 2> $O/bench.err
python3 $R/bench.py --meshlet-source buffer > $O/bench_line_meshlet_buffer.json 2>> $O/bench.err
for src in stream buffer; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$src -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-sample-entities 0 --meshlet-source $src > $O/prof_$src.log 2>&1
  grep '^{' $O/prof_$src.log | tail -1 > $O/bench_line_under_rocprof_$src.json
  f=$(find $O/prof_$src -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$src.csv; rm -rf $O/prof_$src $O/prof_$src.log
done
ORBIT_BENCH_ONE_DEVICE=1 ORBIT_BENCH_BACKEND=gloo python3 $R/bench.py --gpus 2 --cpu-sample-entities 0 > $O/bench_line_two_ranks_one_device.json 2>> $O/bench.err
ORBIT_BENCH_ONE_DEVICE=1 ORBIT_BENCH_BACKEND=gloo python3 $R/bench.py --gpus 2 --signalling collectives --cpu-sample-entities 0 > $O/bench_line_two_ranks_one_device_collectives.json 2>> $O/bench.err
cd $R
python3 tools/bench_c5_two_pass.py --check > $O/c5_two_pass.json 2>> $O/bench.err
python3 tools/bench_configs.py > $O/configs.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_configs -- python3 tools/bench_configs.py > /dev/null 2>> $O/bench.err
f=$(find $O/prof_configs -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_configs.csv; rm -rf $O/prof_configs
python3 tools/bench_views.py > $O/views.json 2>> $O/bench.err
bash tools/shard_pieces.sh stream > $O/shard_pieces.txt 2>> $O/bench.err
bash tools/traffic.sh stream > $O/traffic_stream.log 2>&1; bash tools/traffic.sh buffer > $O/traffic_buffer.log 2>&1
cp $R/gpurun_out/traffic_stream.json $R/gpurun_out/traffic.json $O/ 2>/dev/null
# the default line again, now that the PMC passes of THIS library exist (roofline.traffic_current)
cp $O/traffic_stream.json $O/traffic.json $R/profiles/ && python3 $R/bench.py > $O/bench_line.json 2>> $O/bench.err
ls -la $O
