#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" <bench args...>   -> gpurun_out/<tag>_pmc.csv (this library's kernels only)
tag=$1; shift; ctrs=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $root/gpurun_out /tmp/pmc_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_$tag -- python3 $root/tools/pmc_driver.py "$@" > $root/gpurun_out/${tag}_bench.json 2> $root/gpurun_out/${tag}_err.log
f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
cd $root
python3 - "$f" "$tag" <<'PY'
import csv, sys, collections
f, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if "k_" not in name or "anonymous" not in name: continue
    short = name[name.index("k_"):].split("(")[0]
    agg[short][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[(short, r["Counter_Name"])] += 1
with open("gpurun_out/%s_pmc.csv" % tag, "w") as out:
    ctr = sorted({c for k in agg for c in agg[k]})
    out.write("kernel,dispatches," + ",".join(ctr) + "\n")
    for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
        out.write(k + "," + str(max(calls[(k, c)] for c in ctr)) + "," + ",".join("%.6g" % agg[k].get(c, 0) for c in ctr) + "\n")
print(open("gpurun_out/%s_pmc.csv" % tag).read()[:3500])
PY
