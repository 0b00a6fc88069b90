# Round-6 measurement set (GPU box, repo root): bash tools/r06_final.sh  -> gpurun_out/r06_final/*
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_line.json 2> $O/bench.err
python3 $R/bench.py --meshlet-source buffer > $O/bench_line_meshlet_buffer.json 2>> $O/bench.err
python3 $R/bench.py --frames-in-flight 2 --cpu-sample-entities 0 > $O/bench_line_two_frames_in_flight.json 2>> $O/bench.err
for src in stream buffer; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$src -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-sample-entities 0 --meshlet-source $src > $O/prof_$src.log 2>&1
  grep '^{' $O/prof_$src.log | tail -1 > $O/bench_line_under_rocprof_$src.json
  f=$(find $O/prof_$src -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$src.csv; rm -rf $O/prof_$src $O/prof_$src.log
done
# the driver's own command under rocprofv3 (CPU baseline and all)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/prof_default.log 2>&1
grep '^{' $O/prof_default.log | tail -1 > $O/bench_line_under_rocprof_default_command.json
f=$(find $O/prof_default -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_default_command.csv; rm -rf $O/prof_default $O/prof_default.log
# a rank's shard at N = 8 through the whole sharded step (world of one), strictly serial frames: kernel stats
ORBIT_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_shard -- python3 $R/bench.py --entities 24414 --frames-in-flight 1 --cpu-sample-entities 0 --steps 30 > $O/prof_shard.log 2>&1
f=$(find $O/prof_shard -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_shard_e8.csv; rm -rf $O/prof_shard $O/prof_shard.log
ORBIT_BENCH_ONE_DEVICE=1 ORBIT_BENCH_BACKEND=gloo python3 $R/bench.py --gpus 2 --cpu-sample-entities 0 > $O/bench_line_two_ranks_one_device.json 2>> $O/bench.err
ORBIT_BENCH_ONE_DEVICE=1 ORBIT_BENCH_BACKEND=gloo python3 $R/bench.py --gpus 8 --cpu-sample-entities 0 --verify-single-rank --steps 10 > $O/bench_line_eight_ranks_one_device.json 2>> $O/bench.err
ORBIT_BENCH_FORCE_DIST=1 python3 $R/bench.py --transport rccl --cpu-sample-entities 0 > $O/bench_line_world_of_one_rccl_allgather.json 2>> $O/bench.err
cd $R
python3 tools/bench_c5_two_pass.py --check > $O/c5_two_pass.json 2>> $O/bench.err
python3 tools/bench_configs.py > $O/configs.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_configs -- python3 tools/bench_configs.py > /dev/null 2>> $O/bench.err
# (the largest stats file: the tool itself, not its hop micro-benchmark child process)
f=$(ls -S $(find $O/prof_configs -name "*kernel_stats.csv") | head -1); cp $f $O/kernel_stats_configs.csv; rm -rf $O/prof_configs
python3 tools/bench_views.py > $O/views.json 2>> $O/bench.err
python3 tools/cluster_time.py > $O/cluster_time.json 2>> $O/bench.err
tools/ubench/hop.bin > $O/hops.json 2>> $O/bench.err
bash tools/traffic.sh stream > $O/traffic_stream.log 2>&1; bash tools/traffic.sh buffer > $O/traffic_buffer.log 2>&1
cp $R/gpurun_out/traffic_stream.json $R/gpurun_out/traffic.json $O/ 2>/dev/null
# the default line again, now that the PMC passes of THIS library exist (roofline.traffic_current)
cp $O/traffic_stream.json $O/traffic.json $R/profiles/ && python3 $R/bench.py > $O/bench_line.json 2>> $O/bench.err
ls -la $O
