# usage: bash tools/pmc.sh <outdir> "<counters>"   (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$1
rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-sample-entities 0 > $out.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "orbit" not in k: continue
    m = re.search(r"(\w+)(<[^>]*>)?\(orbit::", k) or re.search(r"(\w+)(<[^>]*>)?\(", k)
    agg[m.group(1) + (m.group(2) or "") if m else k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
rm -rf $out  # the raw counter CSVs are tens of MB: only the summary above is kept (gpurun_out/ is capped at 64 MiB)
