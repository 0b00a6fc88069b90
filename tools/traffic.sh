# Two separate PMC passes (TCC slots: FETCH_SIZE costs 3, WRITE_SIZE 2 -> cannot share a pass)
# usage: bash tools/traffic.sh [stream|buffer]      (default: stream, bench.py's default meshlet source)
src=${1:-stream}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$src -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-sample-entities 0 --meshlet-source $src > $R/gpurun_out/pmc_fetch_$src.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$src -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-sample-entities 0 --meshlet-source $src > $R/gpurun_out/pmc_write_$src.log 2>&1
f=$(find $R/gpurun_out/pmc_fetch_$src -name "*counter_collection.csv" | head -1)
w=$(find $R/gpurun_out/pmc_write_$src -name "*counter_collection.csv" | head -1)
python3 $R/tools/traffic.py $f $w 195313 256 $src
rm -rf $R/gpurun_out/pmc_fetch_$src $R/gpurun_out/pmc_write_$src  # raw counter CSVs: tens of MB
