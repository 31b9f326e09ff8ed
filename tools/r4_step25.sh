#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/r4_s25_bench.json 2> gpurun_out/r4_s25_bench.err || { tail -5 gpurun_out/r4_s25_bench.err; exit 1; }
python3 - <<'PY'
import json
j=json.loads(open("gpurun_out/r4_s25_bench.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["cpu_baseline"]["value"])
print(j["exact_order"]["ms_per_step"], j["fm_query"]["value"], j["fm_query"].get("fm_build"), j.get("sensitivity"))
PY
