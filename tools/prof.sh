#!/bin/bash
# usage: tools_prof.sh <tag> <bench args...>   (runs on the GPU box; writes gpurun_out/<tag>_kernel_stats.csv)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $root/gpurun_out /tmp/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $root/bench.py "$@" > $root/gpurun_out/${tag}_bench.json 2> $root/gpurun_out/${tag}_err.log
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
# keep only this library's kernels (the synthetic-text generator's torch kernels are not the product)
head -1 $f > $root/gpurun_out/${tag}_kernel_stats.csv
grep -E '"[^"]*\(anonymous namespace\)::k_' $f >> $root/gpurun_out/${tag}_kernel_stats.csv
cd $root
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${tag}_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("kernel,calls,total_ms,avg_us,pct_of_lib")
for r in sorted(rows,key=lambda r:-int(r["TotalDurationNs"])):
    name=r["Name"].split("::")[-1].split("(")[0]
    if "<" in r["Name"]: name=r["Name"][r["Name"].index("k_"):].split("(")[0]
    print("%s,%s,%.3f,%.1f,%.1f"%(name,r["Calls"],int(r["TotalDurationNs"])/1e6,float(r["AverageNs"])/1e3,100*int(r["TotalDurationNs"])/tot))
PY
