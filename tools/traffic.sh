# Two separate PMC passes (TCC slots: FETCH_SIZE costs 3, WRITE_SIZE 2 -> cannot share a pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-sample-entities 0 > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-sample-entities 0 > $R/gpurun_out/pmc_write.log 2>&1
f=$(find $R/gpurun_out/pmc_fetch -name "*counter_collection.csv" | head -1)
w=$(find $R/gpurun_out/pmc_write -name "*counter_collection.csv" | head -1)
python3 $R/tools/traffic.py $f $w 195313 256
