# How the kernels' durations move under sustained back-to-back frames: kernel trace of a long bench run per meshlet
# source, and the clocks rocm-smi reports while a longer one runs.  usage: bash tools/sustained.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for src in stream buffer; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/sus_$src -- python3 $R/bench.py --steps 400 --warmup 3 --cpu-sample-entities 0 --meshlet-source $src > $R/gpurun_out/sus_$src.log 2>&1
done
for src in stream buffer; do
  (for i in $(seq 1 40); do rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done) > $R/gpurun_out/clocks_$src.jsonl &
  smi=$!
  python3 $R/bench.py --steps 15000 --warmup 3 --cpu-sample-entities 0 --meshlet-source $src > $R/gpurun_out/long_$src.json 2>/dev/null
  wait $smi
done
