# usage: bash tools/pmc_cmd.sh <outdir> "<counters>" <script.py> [args...]   (GPU box, repo root)
# One rocprofv3 PMC pass over an arbitrary python tool of this repo; prints per-kernel counter means.
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$1; ctr=$2; shift 2
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/"$@" > $out.log 2>&1
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
