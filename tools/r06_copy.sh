# copies gpurun_out/r06_final/* (tools/r06_final.sh on the GPU box) to profiles/ under the committed names
O=gpurun_out/r06_final; P=profiles
cp $O/bench_line.json $P/r06_bench_line.json; cp $O/bench_line_meshlet_buffer.json $P/r06_bench_line_meshlet_buffer.json
cp $O/bench_line_two_ranks_one_device.json $P/r06_bench_line_two_ranks_one_device.json
cp $O/bench_line_eight_ranks_one_device.json $P/r06_bench_line_eight_ranks_one_device.json
cp $O/bench_line_under_rocprof_stream.json $P/r06_bench_line_under_rocprof.json
cp $O/bench_line_under_rocprof_buffer.json $P/r06_bench_line_under_rocprof_meshlet_buffer.json
cp $O/bench_line_under_rocprof_default_command.json $P/r06_bench_line_under_rocprof_default_command.json
cp $O/bench_line_world_of_one_rccl_allgather.json $P/r06_bench_line_world_of_one_rccl_allgather.json
cp $O/c5_two_pass.json $P/r06_c5_two_pass.json; cp $O/configs.json $P/r06_configs.json; cp $O/hops.json $P/r06_hops.json
cp $O/kernel_stats_stream.csv $P/r06_kernel_stats.csv; cp $O/kernel_stats_buffer.csv $P/r06_kernel_stats_meshlet_buffer.csv
cp $O/kernel_stats_default_command.csv $P/r06_kernel_stats_default_command.csv; cp $O/kernel_stats_configs.csv $P/r06_kernel_stats_configs.csv
cp $O/kernel_stats_shard_e8.csv $P/r06_kernel_stats_shard_e8.csv
cp $O/views.json $P/r06_views.json; cp $O/cluster_time.json $P/r06_cluster_time.json
cp $O/traffic_stream.json $P/traffic_stream.json; cp $O/traffic.json $P/traffic.json
cp $O/bench_line_two_frames_in_flight.json $P/r06_bench_line_two_frames_in_flight.json
