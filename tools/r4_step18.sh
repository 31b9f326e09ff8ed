#!/bin/bash
# tests + default bench (all legs) on the library with the shuffle form of word_masks
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_s18_tests.out 2>&1
rc=$?; tail -3 gpurun_out/r4_s18_tests.out
[ $rc -ne 0 ] && exit $rc
timeout -k 10 420 python bench.py > gpurun_out/r4_s18_bench.json 2> gpurun_out/r4_s18_bench.err || exit 1
python3 - <<'PY'
import json
j=json.loads(open("gpurun_out/r4_s18_bench.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"])
print(j["config"].get("phase_ms"))
print({k:(v if not isinstance(v,dict) else {a:b for a,b in v.items() if not isinstance(b,(dict,list))}) for k,v in j.items() if k in("cpu_baseline","fm_query","exact_order","sensitivity")})
PY
