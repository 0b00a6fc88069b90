# kernel stats of a small shard (1/8 and 1/4 of config 5) through the whole sharded step, a world of one
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in 24414 48828; do
ORBIT_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_e8 -- python3 $R/bench.py --entities $e --cpu-sample-entities 0 --steps 30 > $R/gpurun_out/prof_e8.log 2>&1
f=$(find $R/gpurun_out/prof_e8 -name "*kernel_stats.csv" | head -1); echo "== $e"; python3 - "$f" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "orbit" in r["Name"]:
        n = re.sub(r"orbit::\(anonymous namespace\)::|void ", "", r["Name"])
        print(f"  {n[:60]:60s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:8.2f} us  min {float(r['MinNs']) / 1e3:8.2f}")
PY
rm -rf $R/gpurun_out/prof_e8
done
